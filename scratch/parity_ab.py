"""One conditioned state: the bf16 gradient against float64 for several target shifts and kernel selections (numerics A/B, not speed)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_parity_conditioned_gpu as T
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd import _lib as L
dev = torch.device("cuda:0")
sdc = T._condition(dev)
B = T.B
x, y = T.structured_batch(B, 8)
shifts = (0.0, 0.05, 0.1, 0.2)
sd = {k: v.clone() for k, v in sdc.items()}
names = O._leafify(sd)
out, _ = O.krn_predict(sd, x.double(), True, "")
refs = []
for sh in shifts:
    l = O.krn_loss(out, (y + sh).double())[0]
    g = torch.autograd.grad(l, [sd[k] for k in names], retain_graph=True)
    refs.append((float(l), [t.flatten() for t in g]))
lib = L.lib()
def run(tag):
    eng = KrnEngine(T.K).attach(dev, "bf16")
    row = []
    for sh, (l, gr) in zip(shifts, refs):
        T.load_state(eng, sdc); eng.grads.zero_()
        _, scal, _ = eng.forward(x.to(dev), (y + sh).to(dev), training=True); eng.backward(B); torch.cuda.synchronize()
        gh = [eng.param_view(i, eng.grads).double().cpu().flatten() for i in eng.param_infos]
        G, R = torch.cat(gh), torch.cat(gr)
        worst = min((T._cos(a, b), i[0]) for a, b, i in zip(gh, gr, eng.param_infos) if i[0].endswith(".weight") and a.numel() > 64)
        row.append("shift %.2f: cos %.4f ratio %.3f |g| %.2e worst %.2f %s" % (sh, T._cos(G, R), float(G.norm() / R.norm()), float(R.norm()), worst[0], worst[1]))
    print(tag + "\n   " + "\n   ".join(row))
run("default")
lib.spb_debug_set_pwb(32, 0, 4); run("pwb old config (z read, 32-row chunks)")
lib.spb_debug_set_pwb(16, 1, 8); lib.spb_debug_set_dw_tile(0, 0); run("dw tile off")
