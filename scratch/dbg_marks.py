import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ghiasi_oracle as G
from speedplusbaseline_amd.styleaug import Ghiasi
from speedplusbaseline_amd import _lib as L
dev = torch.device("cuda:0")
for kv in os.environ.get("SPB_DBG", "").split(";"):
    if kv:
        k, v = kv.split("="); getattr(L.lib(), k)(*[int(x) for x in v.split(",")])
net = Ghiasi(); net.load_state_dict(G.init_state()); net.to(dev)
x = torch.rand(48, 3, 224, 224, device=dev); s = torch.randn(48, 100, device=dev)
for _ in range(3): net(x, s)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): net(x, s)
torch.cuda.synchronize(); print("forward %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
for rep in range(3):
    net.profile = []
    net(x, s); torch.cuda.synchronize()
    marks = net.profile; net.profile = None
    if rep: print("rep", rep, " ".join("%s=%.3f" % (l1.split()[2] if l1.startswith("gconv") else l1, e0.elapsed_time(e1)) for (l0, e0), (l1, e1) in zip(marks[:-1], marks[1:])))
