import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from tests.test_krn_gpu import load_state, relerr
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
x, y = O.synth_batch(4)
sd = O.init_state(11); names = O._leafify(sd)
loss, lx, ly = O.krn_forward(sd, x, y, training=True); loss.backward()
eng = KrnEngine(11).attach(dev, prec); load_state(eng, O.init_state(11))
eng.grads.zero_()
pred, scal, _ = eng.forward(x.to(dev), y.to(dev), training=True)
eng.backward(4); torch.cuda.synchronize()
print('loss', scal.cpu().numpy(), float(loss))
for info in eng.param_infos:
    g = eng.param_view(info, eng.grads); ref = sd[info[0]].grad
    print('%-32s %-18s rel %.3e  |g| %.3e |ref| %.3e' % (info[0], info[1], relerr(g, ref), float(g.norm()), float(ref.norm())))
