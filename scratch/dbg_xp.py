import torch
from speedplusbaseline_amd import _lib as L, ops
dev = "cuda"; DT = torch.bfloat16
torch.manual_seed(0)
def run(B, H, Ce, C, stride, We, bn=False):
    X = torch.randn(B, H, H, Ce, device=dev).to(DT)
    Wd = (torch.randn(C, 1, 3, 3, device=dev) * 0.3).contiguous()
    Z = (X.float().view(-1, Ce) @ We.float().t()).view(B, H, H, C)
    if bn:
        z2 = Z.double().view(-1, C)
        sums = torch.stack([z2.sum(0), (z2 * z2).sum(0)]).float().unsqueeze(0).contiguous()
        g = (torch.rand(C, device=dev) + 0.5); b = torch.randn(C, device=dev) * 0.2
        pro = ops.bnref(C, sums=sums, gamma=g, beta=b, n=z2.shape[0], R=1, act=L.ACT_RELU6)
    else:
        pro = ops.bnref(C, act=L.ACT_RELU6)
    OH = (H - 1) // stride + 1
    Y0 = torch.empty(B, OH, OH, C, dtype=DT, device=dev); Y1 = torch.empty_like(Y0)
    ops.dwconv_fwd(Z.to(DT), Wd, Y0, pro, stride)
    ops.dwconv_fwd(None, Wd, Y1, pro, stride, expand=(X, We, None))
    torch.cuda.synchronize()
    d = (Y0.float() - Y1.float())
    print(B, H, Ce, C, stride, "bn" if bn else "", "rel", (d.norm() / Y0.float().norm()).item())
    e = d.abs().amax((0, 1, 2))
    print("   per-channel err", [round(v, 3) for v in e.tolist()][:72])
run(1, 28, 32, 64, 1, torch.cat([torch.eye(32), torch.eye(32)]).to(dev).to(DT).contiguous())
run(1, 28, 24, 32, 1, torch.eye(32, 24).to(dev).to(DT).contiguous())
run(1, 28, 32, 32, 1, (torch.randn(32, 32) * 0.3).to(dev).to(DT).contiguous())
run(1, 28, 32, 32, 1, (torch.randn(32, 32) * 0.3).to(dev).to(DT).contiguous(), bn=True)
run(1, 28, 32, 64, 1, (torch.randn(64, 32) * 0.3).to(dev).to(DT).contiguous(), bn=True)
run(2, 56, 24, 144, 2, (torch.randn(144, 24) * 0.3).to(dev).to(DT).contiguous(), bn=True)
