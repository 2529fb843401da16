"""sanity: the bf16 HIP train step drives the loss down on one fixed synthetic batch (and f32 alike)"""
import sys, torch
sys.path.insert(0, '.')
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
dev = torch.device('cuda', 0)
for prec in ("fp32", "bf16"):
    B = 48
    eng = KrnEngine(11).attach(dev, prec)
    sd = O.init_state(11)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(dev))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(dev))
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, 224, 224, generator=g).to(dev); y = torch.rand(B, 2, 11, generator=g).to(dev)
    ts = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    losses = []
    for i in range(120):
        s = ts(x, y)
        if i % 10 == 0 or i == 119: losses.append(round(float(s[0]), 4))
    print(prec, losses)
