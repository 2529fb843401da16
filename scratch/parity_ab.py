"""One conditioned state, the bf16 gradient against float64 under different kernel selections (A/B of numerics, not of speed)."""
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
ys = (y + T.TARGET_SHIFT)
sd = {k: v.clone() for k, v in sdc.items()}
names = O._leafify(sd)
out, _ = O.krn_predict(sd, x.double(), True, "")
loss = O.krn_loss(out, y.double())[0]; loss_s = O.krn_loss(out, ys.double())[0]
gmin = torch.cat([g.flatten() for g in torch.autograd.grad(loss, [sd[k] for k in names], retain_graph=True)])
loss_s.backward()
gref = torch.cat([sd[k].grad.flatten() for k in names])
print("float64: loss %.6f shifted %.6f |g| %.3e |g_shift| %.3e" % (float(loss), float(loss_s), float(gmin.norm()), float(gref.norm())))
lib = L.lib()
def run(tag, prec="bf16"):
    eng = KrnEngine(T.K).attach(dev, prec)
    res = []
    for tgt, gr in ((y, gmin), (ys, gref)):
        T.load_state(eng, sdc); eng.grads.zero_()
        _, scal, _ = eng.forward(x.to(dev), tgt.to(dev), training=True); eng.backward(B); torch.cuda.synchronize()
        g = torch.cat([eng.param_view(i, eng.grads).double().cpu().flatten() for i in eng.param_infos])
        res.append((float(scal[0]), T._cos(g, gr), float(g.norm() / gr.norm())))
    print("%-28s loss %.6f cos %.4f ratio %.3f | shifted: loss %.6f cos %.4f ratio %.3f" % ((tag,) + res[0] + res[1]))
run("fp32", "fp32")
run("default"); run("default (again)")
lib.spb_debug_set_dw_tile(0, 0); run("dw tile off")
lib.spb_debug_set_pwb(32, 0, 4); run("+ pwb old config")
lib.spb_debug_set_fused_pw_bwd(0); run("+ fused pw bwd off")
lib.spb_debug_set_fused_pw_bwd(1); lib.spb_debug_set_pwb(16, 1, 8); lib.spb_debug_set_dw_tile(28, 0); run("default (end)")
