"""Gradient probe (not a test): a briefly trained KRN state, then the HIP gradient in fp32 and bf16 mode against the float64 oracle at bs=48,
per tensor -- localises a wrong backward kernel without depending on where a long conditioning run ends.
   python scratch/grad_probe.py [steps]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
import tests.test_parity_conditioned_gpu as T

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
STATE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_state.pt")
if os.path.exists(STATE) and steps >= 0:
    state = {k: (v.double() if v.is_floating_point() else v) for k, v in torch.load(STATE).items()}
    print("state from", STATE)
else:
    steps = abs(steps)
    eng = KrnEngine(T.K).attach(dev, "fp32")
    T.load_state(eng, O.init_state(T.K))
    ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    for it in range(steps):
        if it == 300: ts.lr = 3e-4
        x, y = T.structured_batch(T.B, 100 + it, dev)
        s = ts(x, y)
    print("after %d f32 steps" % steps)
    state = T.dump_state(eng)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    torch.save({k: (v.float() if v.is_floating_point() else v) for k, v in state.items()}, os.path.join(out_dir, "probe_state.pt"))
    state = {k: (v.float().double() if v.is_floating_point() else v) for k, v in state.items()}
x, y = T.structured_batch(T.B, 8)
y_shift = (y + T.TARGET_SHIFT).clamp(0, 1.2)
sd = {k: v.clone() for k, v in state.items()}
names = O._leafify(sd)
out, _ = O.krn_predict(sd, x.double(), True, "")
loss = O.krn_loss(out, y_shift.double())[0]
loss.backward()
g_ref = torch.cat([sd[k].grad.flatten() for k in names])
for prec in ("fp32", "bf16"):
    e2 = KrnEngine(T.K).attach(dev, prec)
    for rep in range(2):
        T.load_state(e2, state)
        e2.grads.zero_()
        _, scal, _ = e2.forward(x.to(dev), y_shift.to(dev), training=True)
        e2.backward(T.B)
        torch.cuda.synchronize()
    g = torch.cat([e2.param_view(i, e2.grads).double().cpu().flatten() for i in e2.param_infos])
    print("%s: loss %.6f (float64 %.6f)  gradient cosine %.4f  norm ratio %.4f" % (prec, float(scal[0]), float(loss), T._cos(g, g_ref), float(g.norm() / g_ref.norm())))
    gn = float(g_ref.norm())
    rows = []
    for i in e2.param_infos:
        r = sd[i[0]].grad
        if float(r.norm()) > 1e-3 * gn:
            h = e2.param_view(i, e2.grads).double().cpu()
            rows.append((T._cos(h.flatten(), r.flatten()), float(h.norm() / r.norm()), i[0]))
    bad = [t for t in rows if t[0] < 0.8 or not (0.8 < t[1] < 1.25)]
    print("  %d of %d tensors off (cosine < 0.8 or ratio outside 0.8..1.25): " % (len(bad), len(rows)) + "; ".join("%.2f %.2f %s" % t for t in bad[:24]))
