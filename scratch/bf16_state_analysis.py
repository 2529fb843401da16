"""CPU analysis of a conditioned KRN state (not a test, no GPU): WHY the bf16 gradient of some states is unrelated to the float64 one.
Everything below runs on the float64 CPU oracle (oracle/krn_oracle.py) -- no HIP kernel is involved -- on a state file saved by
scratch/cond_explore.py (SAVE_AT=<step>; deterministic conditioning, so the file is reproducible from the recipe):

   python scratch/bf16_state_analysis.py gpurun_out/cond_state_2000.pt 0.25 > profiles/r5_bf16_state_analysis.txt

 1. BatchNorm survey on the held-out batch: max |mean| / sigma per layer (the round-4 hypothesis "|mean| >> sigma"), smallest sigma.
 2. The gradient (targets + 0.05) of: the oracle with bf16 rounding at the HIP path's storage / operand points (_Net.quant), PyTorch's own
    CPU bf16 autocast of the oracle, bf16 rounding of stored conv outputs only / of matrix operands only, float16 rounding instead.
 3. Rounding confined to the layers of one feature-map size: where the deviation is generated.
 4. Per BatchNorm layer, float64 vs rounding oracle: relative deviation of the forward tensor, and norm ratio / cosine of dL/d(tensor).
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
import warnings; warnings.filterwarnings("ignore")
import torch.nn.functional as F
from oracle import krn_oracle as O
import tests.test_parity_conditioned_gpu as T

path = sys.argv[1]; noise = float(sys.argv[2]) if len(sys.argv) > 2 else T.CLUTTERED
st = torch.load(path)
state = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
x, y = T.structured_batch(T.B, 8, noise=noise)
ys = (y + T.TARGET_SHIFT).clamp(0, 1.2)
orig_conv, orig_bn = O._Net.conv, O._Net.bn
print("state %s, held-out batch seed 8, clutter %.2f, targets + %.2f" % (path, noise, T.TARGET_SHIFT))


def grad(dtype=torch.float64, autocast=False, hooks=None):
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
    return float(loss), torch.cat([sd[k].grad.double().flatten() for k in names]), out.detach().double()


def rounding_conv(dt, pred=lambda n, w: True, ops=True, store=True):
    def q(t): return t + (t.detach().to(dt).to(t.dtype) - t.detach())
    def conv(self, x_, name, stride=1, padding=0, groups=1):
        w = self.sd[self.p + name + ".weight"]; b = self.sd.get(self.p + name + ".bias")
        on = pred(name, x_.shape[-1] // stride)
        if on and ops and (groups == 1 or x_.shape[-1] >= self.DW_TILE_MIN): x_, w = q(x_), q(w)
        z = F.conv2d(x_, w, None, stride, padding, 1, groups)
        if b is not None: return z + b.view(1, -1, 1, 1)
        return q(z) if (on and store) else z
    return conv


def report(tag, l, g, out=None):
    print("%-58s loss %.5f  cosine %.4f  norm ratio %.3f  projection %.3f%s" % (tag, l, T._cos(g, g0), float(g.norm() / g0.norm()), float(torch.dot(g, g0) / g0.norm() ** 2),
          "" if out is None else "   outputs vs float64: RMS %.4f" % float((out - out0).pow(2).mean().sqrt())))


# ---- 1
rows = []
def bn_survey(self, xx, name):
    with torch.no_grad():
        m = xx.mean((0, 2, 3)); s = xx.var((0, 2, 3), unbiased=False).sqrt()
        r = m.abs() / (s + 1e-30)
        rows.append((name, float(r.max()), float(r.median()), xx.shape[1], float(s.min())))
    return orig_bn(self, xx, name)
O._Net.bn = bn_survey
l0, g0, out0 = grad()
O._Net.bn = orig_bn
print("\n1. float64: loss %.5f, |g| %.3f.  BatchNorm layers: largest max |mean|/sigma = %.2f (%s); layers with a zero-variance channel: %s"
      % (l0, float(g0.norm()), max(r[1] for r in rows), max(rows, key=lambda r: r[1])[0], [r[0] for r in rows if r[4] == 0.0]))
# ---- 2
print("\n2. the same gradient under 16-bit rounding (all on the CPU oracle):")
O._Net.quant = True
report("bf16 at the HIP path's storage / operand points (_Net.quant)", *grad()); O._Net.quant = False
l, g, o = grad(torch.float32, True); report("PyTorch CPU autocast(bfloat16) of the oracle, f32 master weights", l, g, o)
for tag, kw in (("bf16 rounding of stored convolution outputs only", dict(ops=False)), ("bf16 rounding of matrix operands only", dict(store=False))):
    O._Net.conv = rounding_conv(torch.bfloat16, **kw); report(tag, *grad()); O._Net.conv = orig_conv
O._Net.conv = rounding_conv(torch.float16); report("float16 rounding at the same points (3 more mantissa bits)", *grad()); O._Net.conv = orig_conv
# ---- 3
print("\n3. bf16 rounding confined to the layers producing one feature-map size:")
for tag, pred in (("112x112", lambda n, w: w == 112), ("56x56", lambda n, w: w == 56), ("28x28", lambda n, w: w == 28), ("14x14", lambda n, w: w == 14),
                  ("7x7 backbone", lambda n, w: w == 7 and n.startswith("base")), ("7x7 extras + head", lambda n, w: (n.startswith("extras") and w == 7) or n.startswith("head"))):
    O._Net.conv = rounding_conv(torch.bfloat16, pred); report(tag, *grad()); O._Net.conv = orig_conv
# ---- 4
print("\n4. per BatchNorm layer, rounding oracle vs float64: forward tensor deviation, and dL/d(tensor) norm ratio / cosine")
res = {}
for quant in (False, True):
    zs = []
    def bn_keep(self, xx, name):
        xx.retain_grad(); zs.append((name, xx))
        return orig_bn(self, xx, name)
    O._Net.bn = bn_keep; O._Net.quant = quant
    grad()
    res[quant] = [(n, z.grad.clone(), z.detach().clone()) for n, z in zs]
    O._Net.quant = False; O._Net.bn = orig_bn
for (n, ga, za), (_, gb, zb) in zip(res[False], res[True]):
    print("   %-24s forward rel. deviation %.2e   gradient: |g| float64 %.3e  ratio %.3f  cosine %.4f" % (n, float((zb - za).norm() / za.norm()), float(ga.norm()), float(gb.norm() / ga.norm()), T._cos(gb.flatten(), ga.flatten())))
