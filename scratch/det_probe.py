"""Reproducible-mode probe (not a test): N fused f32 training steps from the oracle's initial state on the deterministic twin
library, twice from scratch -> the two runs must agree BIT FOR BIT (parameters, BatchNorm buffers, losses); misses must be 0;
one step of the float-atomic library beside it for scale.     python scratch/det_probe.py [steps] [precision]"""
import os, sys, time, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
import tests.test_parity_conditioned_gpu as T

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"


def digest(eng):
    h = hashlib.sha256()
    for t in (eng.params, eng.buffers):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def run(det, n):
    eng = KrnEngine(T.K, deterministic=det).attach(dev, prec)
    T.load_state(eng, O.init_state(T.K))
    ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    losses = []
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(n):
        x, y = T.structured_batch(T.B, 100 + it, dev)
        losses.append(ts(x, y)[0:1].clone())
    torch.cuda.synchronize(); dt = time.time() - t0
    miss = eng.det_misses() if det else -1
    return digest(eng), torch.cat(losses).cpu(), dt / n, miss, eng


a = run(True, steps); b = run(True, steps)
print("deterministic %s run 1: digest %s  %.2f ms/step  misses %d  last loss %.6f" % (prec, a[0], a[2] * 1e3, a[3], float(a[1][-1])))
print("deterministic %s run 2: digest %s  %.2f ms/step  misses %d  last loss %.6f" % (prec, b[0], b[2] * 1e3, b[3], float(b[1][-1])))
print("bit-identical: params+buffers %s, losses %s" % (a[0] == b[0], bool(torch.equal(a[1], b[1]))))
c = run(False, steps); d = run(False, steps)
print("float-atomic  %s run 1: digest %s  %.2f ms/step  last loss %.6f" % (prec, c[0], c[2] * 1e3, float(c[1][-1])))
print("float-atomic  %s run 2: digest %s  (identical: %s)" % (prec, d[0], c[0] == d[0]))
print("first-step loss det %.7f  atomic %.7f;  after %d steps: det %.6f atomic %.6f / %.6f" % (float(a[1][0]), float(c[1][0]), steps, float(a[1][-1]), float(c[1][-1]), float(d[1][-1])))
rel = float((a[4].params - c[4].params).norm() / c[4].params.norm())
print("parameter distance det vs atomic after %d steps: %.3e (atomic vs atomic: %.3e)" % (steps, rel, float((d[4].params - c[4].params).norm() / c[4].params.norm())))
