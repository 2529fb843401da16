"""Per-launch table of one KRN train step from the plan's own profiler (HIP events around every launch, everything on ONE
stream: no overlap, so each launch's time is its own) with the ALGORITHMIC bytes of the launch and a floor model
`bytes / 4 TB/s + 4 us`.   usage: python scratch/launch_table.py [--batch 48] [--reps 10]  > profiles/rN_krn_launches.txt
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=48)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
B = args.batch
eng = KrnEngine(11).attach(dev, "bf16")
sd = O.init_state(11)
for info in eng.param_infos:
    eng.param_view(info).copy_(sd[info[0]].to(dev))
for name, shape, off, numel in eng.buffer_infos:
    eng.buffers[off: off + numel].copy_(sd[name].flatten().to(dev))
gen = torch.Generator(device="cpu"); gen.manual_seed(2021)
x = torch.rand(B, 3, 224, 224, generator=gen).to(dev)
y = torch.rand(B, 2, 11, generator=gen).to(dev)
step = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
for _ in range(5):
    step(x, y)
torch.cuda.synchronize()
eng.prof_enable(B, 0, True)
acc = None
for r in range(args.reps + 2):
    step.t += 1; step._refresh_hyper()
    eng.forward(x, y, training=True, slot=0)
    eng.grads.zero_()
    eng.backward(B, slot=0)
    step._update()
    torch.cuda.synchronize()
    rec = eng.prof_launches(B, 0)
    eng.prof_read(B, 0)
    if r < 2:
        continue
    if acc is None:
        acc = [[c, 0.0, b] for c, _, b in rec]
    for i, (c, ms, b) in enumerate(rec):
        acc[i][1] += ms
eng.prof_enable(B, 0, False)
print("# KRN train step bs=%d bf16, plan profiler (one stream), mean of %d steps; floor = MB / 4 TB/s + 4 us" % (B, args.reps))
print("%4s %-18s %8s %8s %8s %8s %8s" % ("i", "family", "us", "MB", "GB/s", "floor_us", "excess"))
tot = totf = 0.0
fam = {}
for i, (c, ms, b) in enumerate(acc):
    us = ms / args.reps * 1e3
    fl = b / 4e12 * 1e6 + 4.0
    tot += us; totf += fl
    f = fam.setdefault(c, [0, 0.0, 0.0, 0.0]); f[0] += 1; f[1] += us; f[2] += b; f[3] += fl
    print("%4d %-18s %8.1f %8.2f %8.0f %8.1f %8.1f" % (i, c, us, b / 1e6, b / (us * 1e-6) / 1e9 if us > 0 else 0, fl, us - fl))
print("# total %.1f us, floor model %.1f us" % (tot, totf))
print("# %-18s %4s %8s %8s %8s %8s" % ("family", "n", "us", "MB", "floor", "excess"))
for c, f in sorted(fam.items(), key=lambda kv: -(kv[1][1] - kv[1][3])):
    print("# %-18s %4d %8.1f %8.1f %8.1f %8.1f" % (c, f[0], f[1], f[2] / 1e6, f[3], f[1] - f[3]))
