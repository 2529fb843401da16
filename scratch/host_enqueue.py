"""host-side cost of enqueuing one KRN train step (no device sync inside the loop)"""
import sys, time, torch
sys.path.insert(0, '.')
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
dev = torch.device('cuda', 0)
B = 48
eng = KrnEngine(11).attach(dev, "bf16")
sd = O.init_state(11)
for info in eng.param_infos:
    eng.param_view(info).copy_(sd[info[0]].to(dev))
x = torch.rand(B, 3, 224, 224, device=dev); y = torch.rand(B, 2, 11, device=dev)
ts = FusedTrainStep(eng, B)
for _ in range(10): ts(x, y)
torch.cuda.synchronize()
# GPU saturated: queue 200 steps, measure host time to enqueue them all vs GPU time
t0 = time.perf_counter()
for _ in range(200): ts(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step (may include back-pressure), total %.3f ms/step" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
# host-only cost: enqueue a few steps into an idle queue
import statistics
hs = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts(x, y); hs.append(time.perf_counter() - t0)
print("host enqueue into an idle queue: median %.3f ms" % (statistics.median(hs) * 1e3))
