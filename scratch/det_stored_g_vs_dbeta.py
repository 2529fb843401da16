import sys, ctypes as C, torch
sys.argv = [sys.argv[0]]
from oracle import krn_oracle as O
from speedplusbaseline_amd import _lib as L
from speedplusbaseline_amd.engine import KrnEngine
dev = "cuda"; B = 48; K = 11
x, y = O.synth_batch(B)
sd0 = O.init_state(K)
def load_state(eng, state):
    for info in eng.param_infos: eng.param_view(info).copy_(state[info[0]].to(dev))
    for name, shape, off, numel in eng.buffer_infos: eng.buffers[off:off+numel].view(shape).copy_(state[name].to(dev))
def g_of(eng, bn_name):
    h, ws = eng._ctx[(B, 0)]
    ai = L.ActInfo()
    for a in range(eng.lib.spb_krn_num_acts(eng.h)):
        eng.lib.spb_krn_ctx_act_info(h, a, C.byref(ai))
        if eng.bn_names[ai.bn_index].startswith(bn_name + ".num"):
            n = B * ai.H * ai.W * ai.C
            return ws[ai.g_off: ai.g_off + n * 2].view(torch.bfloat16).view(-1, ai.C).double()
for det in (False, True):
    eng = KrnEngine(K, deterministic=det).attach(dev, "bf16")
    load_state(eng, sd0); eng.grads.zero_()
    pred, scal, _ = eng.forward(x.to(dev), y.to(dev), training=True)
    eng.backward(B); torch.cuda.synchronize()
    gr = {i[0]: eng.param_view(i, eng.grads).double().flatten().clone() for i in eng.param_infos}
    for bn in ("base.0.1", "base.1.conv.0.1", "base.2.conv.0.1", "base.3.conv.0.1", "base.4.conv.0.1", "base.5.conv.0.1"):
        g = g_of(eng, bn)
        sg = g.sum(0)
        db = gr[bn + ".bias"]
        print("det" if det else "atm", bn, "|sum g from the stored g| %.4e  |dbeta| %.4e  rel diff %.3e   |dgamma| %.4e" %
              (float(sg.norm()), float(db.norm()), float((sg - db).norm() / (sg.norm() + 1e-30)), float(gr[bn + ".weight"].norm())))
    if det: print("misses", eng.det_misses())
