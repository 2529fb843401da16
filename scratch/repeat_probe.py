"""the same fused train step (lr = 0) repeated N times on one state and one batch: loss and gradient must agree run to run up to the
f32-atomics noise; an intermittent race shows as an outlier (not a test)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
import tests.test_parity_conditioned_gpu as T
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
state = {k: (v.double() if v.is_floating_point() else v) for k, v in torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_state.pt")).items()}
x, y = T.structured_batch(T.B, 8, dev)
for prec in ("fp32", "bf16"):
    eng = KrnEngine(T.K).attach(dev, prec)
    T.load_state(eng, state)
    ts = FusedTrainStep(eng, T.B, kind="sgd", lr=0.0, momentum=0.0, weight_decay=0.0, max_norm=1e9)
    ref = None; worst = (0.0, -1); losses = []
    for it in range(N):
        s = ts(x, y)
        g = eng.grads.clone()
        losses.append(float(s[0]))
        if ref is None:
            ref = g
        else:
            d = float((g - ref).norm() / ref.norm())
            if d > worst[0]: worst = (d, it)
    e2 = KrnEngine(T.K).attach(dev, prec)
    T.load_state(e2, state)
    e2.grads.zero_()
    e2.forward(x, y, training=True)
    e2.backward(T.B)
    torch.cuda.synchronize()
    print("%s: fused-step gradient vs forward()/backward() gradient of a fresh engine: relative difference %.3e (norms %.4e / %.4e)"
          % (prec, float((ref - e2.grads).norm() / e2.grads.norm()), float(ref.norm()), float(e2.grads.norm())))
    print("%s: %d repeats: loss min %.6f max %.6f; largest relative gradient deviation from the first run %.3e (step %d)" % (prec, N, min(losses), max(losses), worst[0], worst[1]))
