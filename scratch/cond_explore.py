"""Conditioning-recipe explorer (not a test): deterministic f32 training, then at checkpoints the bf16 gradient against float64 on the
held-out batch the parity test uses.   python scratch/cond_explore.py "1000,1600,2000,2400,3000" """
import os, sys, time, hashlib, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
import tests.test_parity_conditioned_gpu as T

dev = torch.device("cuda:0")
cps = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000,1600,2000,2400").split(",")]
sched = eval(sys.argv[2]) if len(sys.argv) > 2 else {0: 1e-3, 300: 3e-4, 600: 1e-4, 1000: 3e-5, 1600: 1e-5}
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 100
print("checkpoints", cps, "schedule", sched, "seed0", seed0, flush=True)


def digest(eng):
    h = hashlib.sha256()
    for t in (eng.params, eng.buffers):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


eng = KrnEngine(T.K, deterministic=True).attach(dev, "fp32")
T.load_state(eng, O.init_state(T.K))
ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
x8, y8 = T.structured_batch(T.B, 8)
ys8 = (y8 + T.TARGET_SHIFT).clamp(0, 1.2)


def evaluate(step):
    state = T.dump_state(eng)
    sd = {k: v.clone() for k, v in state.items()}
    names = O._leafify(sd)
    t0 = time.time()
    out, _ = O.krn_predict(sd, x8.double(), True, "")
    loss = O.krn_loss(out, y8.double())[0]
    loss_s = O.krn_loss(out, ys8.double())[0]
    loss_s.backward()
    g_ref = torch.cat([sd[k].grad.flatten() for k in names])
    t_or = time.time() - t0
    res = []
    for det in (False, True):
        e2 = KrnEngine(T.K, deterministic=det).attach(dev, "bf16")
        gs = []
        for rep in range(1 if det else 8):
            T.load_state(e2, state)
            e2.grads.zero_()
            _, scal, _ = e2.forward(x8.to(dev), ys8.to(dev), training=True)
            e2.backward(T.B)
            torch.cuda.synchronize()
            gs.append(torch.cat([e2.param_view(i, e2.grads).double().cpu().flatten() for i in e2.param_infos]))
        g = torch.stack(gs).mean(0)
        single = [T._cos(v, g_ref) for v in gs]
        res.append((T._cos(g, g_ref), float(g.norm() / g_ref.norm()), float(scal[0]), min(single), max(single)))
        del e2
    print("step %5d digest %s: f64 loss %.5f shifted %.5f |g| %.3f (oracle %.0f s) | bf16 atomic mean-of-8: cos %.4f ratio %.3f loss %.5f (singles %.3f..%.3f) | bf16 exact: cos %.4f ratio %.3f loss %.5f"
          % (step, digest(eng), float(loss), float(loss_s), float(g_ref.norm()), t_or, res[0][0], res[0][1], res[0][2], res[0][3], res[0][4], res[1][0], res[1][1], res[1][2]), flush=True)
    return state


hist = []
t0 = time.time()
for it in range(max(cps)):
    if it in sched:
        ts.lr = sched[it]
    x, y = T.structured_batch(T.B, seed0 + it, dev)
    s = ts(x, y)
    hist.append(s[0:1].clone())
    if it + 1 in cps:
        torch.cuda.synchronize()
        h = torch.cat(hist[-50:]).cpu()
        print("  trained to %d in %.0f s; last-50 loss median %.5f max %.5f" % (it + 1, time.time() - t0, float(h.median()), float(h.max())), flush=True)
        st = evaluate(it + 1)
        if str(it + 1) in os.environ.get("SAVE_AT", "").split(","):
            os.makedirs("gpurun_out", exist_ok=True)
            torch.save({k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}, "gpurun_out/cond_state_%d.pt" % (it + 1))
print("misses", eng.det_misses())
