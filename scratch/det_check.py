"""run-to-run reproducibility of one KRN backward (same inputs, same weights): relative L2 difference of the gradient arena"""
import os, sys, torch
sys.path.insert(0, '.')
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd import _lib as L
device = torch.device('cuda', 0)
def grads(B, prec, side, fused, dwmode=None):
    L.lib().spb_debug_set_side_wgrad(side); L.lib().spb_debug_set_fused_pw_bwd(fused)
    if dwmode is not None: L.lib().spb_debug_set_dw_mode(dwmode)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, 224, 224, generator=g).to(device); y = torch.rand(B, 2, 11, generator=g).to(device)
    eng = KrnEngine(11).attach(device, prec)
    sd = O.init_state(11)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(device))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(device))
    out = []
    for _ in range(3):
        _, sc, _ = eng.forward(x, y, training=True, slot=0)
        eng.grads.zero_()
        eng.backward(B, slot=0)
        torch.cuda.synchronize()
        out.append(eng.grads.clone())
    return out, eng
for B in (8, 48):
    for prec in ("fp32", "bf16"):
        for side, fused in ((1, 1), (0, 1), (0, 0)):
            gs, eng = grads(B, prec, side, fused)
            d01 = float((gs[0]-gs[1]).norm()/gs[0].norm()); d12 = float((gs[1]-gs[2]).norm()/gs[1].norm())
            # which tensors differ most
            worst = []
            for info in eng.param_infos:
                v0 = eng.param_view(info, gs[0]); v1 = eng.param_view(info, gs[1])
                n = float(v0.norm())
                if n > 0:
                    worst.append((float((v0 - v1).norm()) / n, info[0]))
            worst.sort(reverse=True)
            print("B=%d %s side=%d fused=%d: run0-run1 %.2e run1-run2 %.2e | worst %s" % (B, prec, side, fused, d01, d12,
                  ", ".join("%s %.1e" % (n, e) for e, n in worst[:4])), flush=True)
