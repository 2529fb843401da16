import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import spn_oracle as S
from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet
GOLD = np.load("tests/golden/spn_golden.npz")
NC, B = 5000, 32
dev = torch.device("cuda:0")
sd = S.init_state(NC); x, yc, yw = S.synth_batch(B, NC, seed=23)
net = SpacecraftPoseNet(NC, keep_prob=0.0, pretrain=False, precision="bf16"); net.load_state_dict(sd, strict=True)
net = net.to(dev).train()
out = net.loss_and_grads(x.to(dev), yc.to(dev), yw.to(dev)); torch.cuda.synchronize()
sv = net._saved
for head, g_key in (("fc8", "dc"), ("fc11", "dr")):
    gW = getattr(net, head).weight.grad.float()
    dl = net._ws[g_key].float()            # [B, NC] dlogits (bf16)
    h = sv["in" + head].float()            # [B, 4096]
    ref = dl.t() @ h
    print(head, "rel L2 vs dl^T h:", float((gW - ref).norm() / ref.norm()), "sum", float(gW.sum()), float(ref.sum()),
          "digest", S.checksum(gW.cpu()), S.checksum(ref.cpu()), GOLD["full_grad_sum/%s.weight" % head])
    print("   column sums max", float(gW.sum(0).abs().max()), "row sums of dl max", float(dl.sum(1).abs().max()))
crop = net.fc11.weight.grad[:6, :6].float().cpu().numpy()
print(crop); print(GOLD["full_grad_fc11_crop"])
