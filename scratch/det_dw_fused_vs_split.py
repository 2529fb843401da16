"""the fused depthwise backward instance (input gradient + weight gradient in one pass: what the reproducible library runs) vs the split pair, both libraries"""
import sys, torch
from speedplusbaseline_amd import _lib as L, ops
dev = "cuda"; DT = torch.bfloat16
def run(lib_fn, B, H, C, stride, R):
    L.lib = lib_fn
    torch.manual_seed(1)
    OH = (H - 1) // stride + 1
    G = torch.randn(B, OH, OH, C, device=dev).to(DT); Z = torch.randn(B, OH, OH, C, device=dev).to(DT)
    Zin = (torch.randn(B, H, H, C, device=dev) + 0.3).to(DT)
    Wd = (torch.randn(C, 1, 3, 3, device=dev) * 0.3).contiguous()
    def sums(t, R):
        t2 = t.double().view(-1, C)
        s = torch.stack([t2.sum(0), (t2 * t2).sum(0)]).float()
        out = torch.zeros(R, 2, C, device=dev); out[0] = s; return out
    z2 = Z.double().view(-1, C); g2 = G.double().view(-1, C)
    mean = z2.mean(0); var = z2.var(0, unbiased=False); xh = (z2 - mean) / torch.sqrt(var + 1e-5)
    bs = torch.zeros(R, 2, C, device=dev); bs[0] = torch.stack([g2.sum(0), (g2 * xh).sum(0)]).float()
    ones, zeros = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    pro = ops.bnref(C, sums=sums(Z, R), gamma=ones * 1.3, beta=zeros, bsums=bs, n=z2.shape[0], R=R, act=L.ACT_RELU6)
    pro_in = ops.bnref(C, sums=sums(Zin, R), gamma=ones * 0.9, beta=zeros + 0.1, n=B * H * H, R=R, act=L.ACT_RELU6)
    res = {}
    for fused in (False, True):
        Y = torch.empty(B, H, H, C, dtype=DT, device=dev); osums = torch.zeros(R, 2, C, device=dev); dW = torch.zeros(C, 1, 3, 3, device=dev)
        if fused:
            ops.dwconv_dgrad(G, Z, Wd, Y, pro, stride, (H, H), epi=pro_in, Zout=Zin, osums=osums, oR=R, dW=dW)
        else:
            ops.dwconv_dgrad(G, Z, Wd, Y, pro, stride, (H, H), epi=pro_in, Zout=Zin, osums=osums, oR=R)
            ops.dwconv_wgrad(G, Z, Zin, Wd, dW, pro, pro_in, stride)
        torch.cuda.synchronize()
        res[fused] = (Y.float(), osums.sum(0), dW)
    a, b = res[False], res[True]
    rel = lambda u, v: float((u - v).norm() / (v.norm() + 1e-30))
    print("%-10s B%d H%d C%d s%d R%d: Y rel %.2e  sum g rel %.2e  sum g*xhat rel %.2e (norm ratio %.3f)  dW rel %.2e" %
          (lib_fn.__name__, B, H, C, stride, R, rel(b[0], a[0]), rel(b[1][0], a[1][0]), rel(b[1][1], a[1][1]), float(b[1][1].norm() / a[1][1].norm()), rel(b[2], a[2])))
    return res
for lib_fn in (L.lib, L.lib_det):
    orig = L.lib
    for shp in ((4, 112, 32, 1, 8), (4, 112, 96, 2, 8), (4, 56, 144, 1, 8), (8, 28, 192, 1, 1)):
        run(lib_fn, *shp)
    L.lib = orig
