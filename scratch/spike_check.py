"""Are the loss spikes of the f32 conditioning run real?  On a spike, the same batch through the float64 oracle on the same weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_parity_conditioned_gpu as T
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
dev = torch.device("cuda:0")
eng = KrnEngine(T.K).attach(dev, "fp32")
T.load_state(eng, O.init_state(T.K))
ts = FusedTrainStep(eng, T.B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
found = 0
for it in range(4000):
    if it in T.LR_AT: ts.lr = T.LR_AT[it]
    if it == 1000: ts.lr = 3e-5
    before = T.dump_state(eng) if it >= 1000 and it % 1 == 0 and False else None
    x, y = T.structured_batch(T.B, 100 + it, dev)
    if it >= 1000:
        snap_p, snap_b, snap_n = eng.params.clone(), eng.buffers.clone(), eng.nbt.clone()
    s = ts(x, y)
    if it >= 1000 and float(s[0]) > 0.5:
        keep_p, keep_b, keep_n = eng.params.clone(), eng.buffers.clone(), eng.nbt.clone()
        eng.params.copy_(snap_p); eng.buffers.copy_(snap_b); eng.nbt.copy_(snap_n)
        sd = T.dump_state(eng)
        _, sc2, _ = eng.forward(x, y, training=True)          # the same forward again on the HIP f32 path
        eng.params.copy_(keep_p); eng.buffers.copy_(keep_b); eng.nbt.copy_(keep_n)
        with torch.no_grad():
            l64 = O.krn_forward({k: v.clone() for k, v in sd.items()}, x.double().cpu(), y.double().cpu(), training=True)[0]
        print("step %d: HIP f32 loss %.4f, repeated %.4f, float64 oracle on the same weights and batch %.4f" % (it, float(s[0]), float(sc2[0]), float(l64)))
        found += 1
        if found >= 3: break
print("spikes examined:", found)
