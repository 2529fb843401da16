"""time the HIP style decoder at the training shape (B=48, 224x224)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ghiasi_oracle as G
from speedplusbaseline_amd.styleaug import Ghiasi
dev = torch.device("cuda:0")
import os
if os.environ.get("GCONV_WLDS_PXG"):
    from speedplusbaseline_amd import _lib as L
    L.lib().spb_debug_set_gconv_wlds_pxg(int(os.environ["GCONV_WLDS_PXG"]))
if os.environ.get("GCONV_SLAB_PF"):
    from speedplusbaseline_amd import _lib as L
    L.lib().spb_debug_set_gconv_slab_pf(int(os.environ["GCONV_SLAB_PF"]))
if os.environ.get("GCONV_SLAB"):
    from speedplusbaseline_amd import _lib as L
    L.lib().spb_debug_set_gconv_slab(int(os.environ["GCONV_SLAB"]))
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # bf16 | fp16 (the IEEE-half build of the same kernels)
net = Ghiasi(precision=PREC); net.load_state_dict(G.init_state()); net.to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
x = torch.rand(B, 3, 224, 224, device=dev); s = torch.randn(B, 100, device=dev)
for _ in range(3): net(x, s)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): net(x, s)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("Ghiasi forward (%s) B=%d: %.3f ms  (%.1f TFLOP/s at 15.43 GFLOP/img)" % (PREC, B, dt * 1e3, 15.43e9 * B / dt / 1e12))

for _ in range(2):   # the first pass creates the events (one of them stalls ~40 ms in the runtime); the second is reported
    net.profile = []
    net(x, s); torch.cuda.synchronize()
    marks = net.profile; net.profile = None
agg = {}
for (l0, e0), (l1, e1) in zip(marks[:-1], marks[1:]):
    agg.setdefault(l1, [0, 0.0]); agg[l1][0] += 1; agg[l1][1] += e0.elapsed_time(e1)
for k, (n_, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s x%2d  %8.3f ms" % (k, n_, ms))
