"""Soak run (not a test): 20 000 bf16 fused KRN steps on fresh structured batches (the clean conditioning distribution, lr 1e-4 after a short warm
phase), float-atomic product library: no non-finite loss, no hang, the loss stays trained.   python scratch/soak.py [steps] [precision]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
import tests.test_parity_conditioned_gpu as T
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
eng = KrnEngine(T.K).attach(dev, prec)
T.load_state(eng, O.init_state(T.K))
ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
keep = []
t0 = time.time()
for it in range(steps):
    if it in T.SCHEDULE: ts.lr = T.SCHEDULE[it]
    if it == 2000: ts.lr = 1e-4
    x, y = T.structured_batch(T.B, 100 + it, dev, T.CLEAN)
    s = ts(x, y)
    if it % 500 == 0 or it == steps - 1: keep.append(s[0:1].clone())
torch.cuda.synchronize()
dt = time.time() - t0
l = torch.cat(keep).cpu()
print("%s: %d steps in %.1f s (%.2f ms/step incl. data generation); loss every 500 steps: %s" % (prec, steps, dt, dt / steps * 1e3, [round(float(v), 4) for v in l]))
print("all finite:", bool(torch.isfinite(l).all()), " parameters finite:", bool(torch.isfinite(eng.params).all()), " max |param| %.3f" % float(eng.params.abs().max()))
if prec == "fp16":
    from speedplusbaseline_amd import _lib as L
    a = eng.amp.cpu(); print("GradScaler: %d steps taken of %d, loss scale %g" % (int(a[L.AMP_STEPS]), steps, float(a[L.AMP_SCALE])))
