"""the statistics-only expand pass (spb_pwconv_gemm with Y = NULL) alone: time per launch by shape / operand / Ymat"""
import torch
from speedplusbaseline_amd import _lib as L, ops
dev = "cuda"; DT = torch.bfloat16
def bench(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, K, N in ((602112, 16, 96), (150528, 24, 144)):
    X = torch.randn(M, K, device=dev).to(DT); W = (torch.randn(N, K, device=dev) * 0.3).to(DT)
    Ym = torch.empty(M, K, dtype=DT, device=dev); Y = torch.empty(M, N, dtype=DT, device=dev)
    for R in (1, 8):
        sums = torch.zeros(R, 2, K, device=dev); x2 = X.double(); sums[0, 0] = x2.sum(0).float(); sums[0, 1] = (x2 * x2).sum(0).float()
        pro = ops.bnref(K, sums=sums, gamma=torch.ones(K, device=dev), beta=torch.zeros(K, device=dev), n=M, R=R)
        osums = torch.zeros(8, 2, N, device=dev)
        t_full = bench(lambda: ops.pwconv_gemm(X, W, Y, pro, 1, 1, osums=osums, oR=8))
        t_so = bench(lambda: ops.pwconv_gemm(X, W, None, pro, 1, 1, osums=osums, oR=8))
        t_som = bench(lambda: ops.pwconv_gemm(X, W, None, pro, 1, 1, osums=osums, oR=8, Ymat=Ym))
        print("M=%d K=%d N=%d R=%d: storing GEMM %.1f us | statistics only %.1f us | + operand written %.1f us" % (M, K, N, R, t_full, t_so, t_som))
    ident = ops.bnref(K)
    print("   identity operand: statistics only %.1f us" % bench(lambda: ops.pwconv_gemm(X, W, None, ident, 1, 1, osums=osums, oR=8)))
    t_copy = bench(lambda: Ym.copy_(X))
    print("   torch copy of the operand: %.1f us; empty-ish kernel (64 KB fill): %.1f us" % (t_copy, bench(lambda: osums.zero_())))
