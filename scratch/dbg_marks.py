import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ghiasi_oracle as G
from speedplusbaseline_amd.styleaug import Ghiasi
dev = torch.device("cuda:0")
net = Ghiasi(); net.load_state_dict(G.init_state()); net.to(dev)
x = torch.rand(48, 3, 224, 224, device=dev); s = torch.randn(48, 100, device=dev)
for _ in range(3): net(x, s)
torch.cuda.synchronize()
for rep in range(3):
    net.profile = []
    net(x, s); torch.cuda.synchronize()
    marks = net.profile; net.profile = None
    print("rep", rep, " ".join("%s=%.3f" % (l1.split()[2] if l1.startswith("gconv") else l1, e0.elapsed_time(e1)) for (l0, e0), (l1, e1) in zip(marks[:-1], marks[1:])))
