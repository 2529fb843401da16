"""Does the benchmarked dtype TRAIN?  The conditioning recipe of tests/test_parity_conditioned_gpu.py (AdamW, fresh structured batch per step,
clutter 0.05) run in float32, bfloat16 and float16 on the float-atomic product library, 2000 steps each from the same initial state; then the
float64 oracle evaluates every final state on one held-out batch (eval mode).  Prints the loss trajectory and the held-out keypoint MSE.
    python scratch/train_curves.py > profiles/r5_train_curves.txt"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
import warnings; warnings.filterwarnings("ignore")
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
from speedplusbaseline_amd import _lib as L
import tests.test_parity_conditioned_gpu as T

dev = torch.device("cuda:0")
xh, yh = T.structured_batch(T.B, 7, noise=T.CLEAN)
print("conditioning recipe (AdamW wd 0.01, clip 1.0, lr %s), %d steps, bs=%d, clutter %.2f; float-atomic product library" % (T.SCHEDULE, T.STEPS, T.B, T.CLEAN))
for prec in ("fp32", "bf16", "fp16"):
    eng = KrnEngine(T.K).attach(dev, prec)
    T.load_state(eng, O.init_state(T.K))
    ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    losses = []
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(T.STEPS):
        if it in T.SCHEDULE:
            ts.lr = T.SCHEDULE[it]
        x, y = T.structured_batch(T.B, 100 + it, dev, T.CLEAN)
        losses.append(ts(x, y)[0:1].clone())
    torch.cuda.synchronize(); dt = time.time() - t0
    losses = torch.cat(losses).cpu()
    sd = T.dump_state(eng)
    with torch.no_grad():
        xc, yc = O.krn_forward({k: v.clone() for k, v in sd.items()}, xh.double(), None, training=False)
    mse = float(((torch.stack([xc, yc], 1) - yh.double()) ** 2).mean())
    extra = ""
    if prec == "fp16":
        amp = eng.amp.cpu()
        extra = "; GradScaler: %d of %d steps taken, loss scale now %g" % (int(amp[L.AMP_STEPS]), T.STEPS, float(amp[L.AMP_SCALE]))
    print("%s: %.2f ms/step incl. data; loss every 200 steps %s; median of the last 50: %.5f; held-out keypoint MSE (float64 oracle, eval mode) %.3e%s"
          % (prec, dt / T.STEPS * 1e3, [round(float(v), 4) for v in losses[::200]], float(losses[-50:].median()), mse, extra), flush=True)
