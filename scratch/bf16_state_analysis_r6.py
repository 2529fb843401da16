"""CPU analysis (round 6, follow-up of scratch/bf16_state_analysis.py; no GPU, no HIP kernel): on the saved CHAOTIC conditioned state, which of the bf16
rounding points of the HIP path costs the gradient's agreement with float64 -- and what did round 6's "virtual expanded tensors" (the expand outputs of
blocks 2-4 are recomputed in f32 and never rounded) change?  The rounding points are switched one family at a time on the float64 oracle:

   python scratch/bf16_state_analysis_r6.py gpurun_out/cond_state_2000.pt 0.25 > profiles/r6_bf16_state_analysis.txt
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(8, os.cpu_count() or 1))
import warnings; warnings.filterwarnings("ignore")
import torch.nn.functional as F
from oracle import krn_oracle as O
import tests.test_parity_conditioned_gpu as T

path = sys.argv[1]; noise = float(sys.argv[2]) if len(sys.argv) > 2 else T.CLUTTERED
st = torch.load(path)
state = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
x, y = T.structured_batch(T.B, 8, noise=noise)
ys = (y + T.TARGET_SHIFT).clamp(0, 1.2)
orig_conv = O._Net.conv
print("state %s, held-out batch seed 8, clutter %.2f, targets + %.2f" % (path, noise, T.TARGET_SHIFT), flush=True)


def grad(autocast=False):
    dtype = torch.float32 if autocast else torch.float64
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
    names = O._leafify(sd)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out, _ = O.krn_predict(sd, x.to(dtype), True, "")
        out = out.float()
    else:
        out, _ = O.krn_predict(sd, x.to(dtype), True, "")
    loss = O.krn_loss(out, ys.to(out.dtype))[0]
    loss.backward()
    return float(loss), torch.cat([sd[k].grad.double().flatten() for k in names])


def q(t, dt=torch.bfloat16):
    return t + (t.detach().to(dt).to(t.dtype) - t.detach())


def placement(store=lambda name, wout: True, operand=lambda name, win: True, dw_operand=lambda name, win: True, dt=torch.bfloat16):
    """the HIP path's rounding points with predicates: store(name, output width) -- the convolution output is stored rounded; operand(name, input width) --
    matrix-core operands (input and weight) rounded; dw_operand -- the depthwise layers of the maps >= 28 wide stage a rounded operand and rounded taps"""
    def conv(self, x_, name, stride=1, padding=0, groups=1):
        w = self.sd[self.p + name + ".weight"]; b = self.sd.get(self.p + name + ".bias")
        win = x_.shape[-1]
        if groups == 1:
            if operand(name, win): x_, w = q(x_, dt), q(w, dt)
        elif win >= self.DW_TILE_MIN and dw_operand(name, win): x_, w = q(x_, dt), q(w, dt)
        z = F.conv2d(x_, w, None, stride, padding, 1, groups)
        if b is not None: return z + b.view(1, -1, 1, 1)
        return q(z, dt) if store(name, win // stride) else z
    return conv


l0, g0 = grad()
print("float64: loss %.5f, |g| %.3f" % (l0, float(g0.norm())), flush=True)


def run(tag, conv=None, joins=False, **kw):
    if conv is not None: O._Net.conv = conv
    O._Net.quant = joins          # (with a replaced conv this switch only rounds the materialised residual sums and the concatenation)
    l, g = grad(**kw)
    O._Net.conv = orig_conv; O._Net.quant = False
    print("%-104s loss %.5f  cosine %.4f  norm ratio %.3f" % (tag, l, T._cos(g, g0), float(g.norm() / g0.norm())), flush=True)


VIRT = ("base.2.conv.0.0", "base.3.conv.0.0", "base.4.conv.0.0")     # expand convolutions of blocks 2-4: outputs never stored since round 6
is_expand = lambda n: n.startswith("base.") and n.endswith(".conv.0.0") and not n.startswith("base.1.")
print("\nA. rounds 1-5 against round 6 (every other rounding point as the HIP path has it; straight-through backward, gradients are not rounded here):")
run("round 5: every convolution output stored as bf16", placement(), joins=True)
run("round 6: the expand outputs of blocks 2-4 recomputed in f32, never rounded", placement(store=lambda n, w: n not in VIRT), joins=True)
run("hypothetical: NO expand output rounded (all 16 expand convolutions virtual)", placement(store=lambda n, w: not is_expand(n)), joins=True)
run("PyTorch CPU autocast(bfloat16) of the oracle, f32 master weights", autocast=True)
print("\nB. one family of rounding points at a time, everything else float64:")
run("stored outputs only, round-6 set", placement(store=lambda n, w: n not in VIRT, operand=lambda n, w: False, dw_operand=lambda n, w: False))
run("matrix-core operands only (1x1 convolutions, stem, head)", placement(store=lambda n, w: False, dw_operand=lambda n, w: False))
run("depthwise operand tiles + taps only (maps >= 28 wide)", placement(store=lambda n, w: False, operand=lambda n, w: False))
print("\nC. the 112x112 layers (alone responsible for 0.37 in round 5's analysis), one tensor at a time, everything else float64:")
for tag, kw in (("stem output stored (base.0.0, 32 ch)", dict(store=lambda n, w: n == "base.0.0", operand=lambda n, w: False, dw_operand=lambda n, w: False)),
                ("stem operands (image, weights)", dict(store=lambda n, w: False, operand=lambda n, w: n == "base.0.0", dw_operand=lambda n, w: False)),
                ("block 1 depthwise: operand tile + taps", dict(store=lambda n, w: False, operand=lambda n, w: False, dw_operand=lambda n, w: n == "base.1.conv.0.0")),
                ("block 1 depthwise output stored (32 ch)", dict(store=lambda n, w: n == "base.1.conv.0.0", operand=lambda n, w: False, dw_operand=lambda n, w: False)),
                ("block 1 project: operands", dict(store=lambda n, w: False, operand=lambda n, w: n == "base.1.conv.1", dw_operand=lambda n, w: False)),
                ("block 1 project output stored (16 ch)", dict(store=lambda n, w: n == "base.1.conv.1", operand=lambda n, w: False, dw_operand=lambda n, w: False)),
                ("block 2 expand: operands (the 16-channel operand x~ the kernels recompute from)", dict(store=lambda n, w: False, operand=lambda n, w: n == "base.2.conv.0.0", dw_operand=lambda n, w: False)),
                ("block 2 expand output stored (96 ch; round 5 only)", dict(store=lambda n, w: n == "base.2.conv.0.0", operand=lambda n, w: False, dw_operand=lambda n, w: False)),
                ("block 2 depthwise (stride 2): operand tile + taps", dict(store=lambda n, w: False, operand=lambda n, w: False, dw_operand=lambda n, w: n == "base.2.conv.1.0"))):
    run(tag, placement(**kw))
print("\nD. the same placement (round 6) in IEEE half:")
run("float16 at the round-6 points", placement(store=lambda n, w: n not in VIRT, dt=torch.float16))   # (joins left in float64: _Net.q rounds to bf16 only)
print("\nE. the stem's operands (follow-up of C: the largest single item), on top of the round-6 placement:")
r6 = lambda n, w: n not in VIRT
run("round 6 + stem operands exact (image and stem weights as hi + lo pairs)", placement(store=r6, operand=lambda n, w: n != "base.0.0"), joins=True)
run("round 6 + stem operands exact + block-1 depthwise operand tile exact", placement(store=r6, operand=lambda n, w: n != "base.0.0", dw_operand=lambda n, w: n != "base.1.conv.0.0"), joins=True)
run("round 6 + all operands of the 112x112 layers exact (stores still bf16)", placement(store=r6, operand=lambda n, w: w != 112 and n != "base.0.0", dw_operand=lambda n, w: w != 112), joins=True)
