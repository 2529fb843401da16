import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from tests.test_krn_gpu import load_state, relerr
dev = torch.device('cuda:0')
x, y = O.synth_batch(4)
eng = KrnEngine(11).attach(dev, 'bf16'); load_state(eng, O.init_state(11))
pred, scal, _ = eng.forward(x.to(dev), y.to(dev), training=True); torch.cuda.synchronize()
init = O.init_state(11)
O._Net.quant = True; O._Net.momentum = 1.0
sd = O.init_state(11, dtype=torch.float64)
with torch.no_grad():
    loss, lx, ly = O.krn_forward(sd, x.double(), y.double(), training=True)
print('loss hip', scal.cpu().numpy(), 'oracle-q', float(loss))
for (name, shape, off, numel) in eng.buffer_infos:
    a = (eng.buffers[off:off+numel].cpu().double() - 0.9*init[name].double()) / 0.1
    b = sd[name]
    print('%-34s rel %.3e  |b| %.3e' % (name, float((a-b).norm()/(b.norm()+1e-30)), float(b.norm())))
