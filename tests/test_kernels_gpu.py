"""Per-kernel parity: each HIP entry point of libspb_hip.so against a float64 PyTorch (CPU) reference of the same op.

Backward kernels are checked against torch.autograd through the composite
    z_in -> act1(bn1(z_in)) -> conv -> bn2 -> act2 -> loss
so the (g, sum g, sum g*xhat) contract between kernels is pinned to autograd's definition of BatchNorm backward.
Tolerances: f32 mode 2e-5 relative (exact f32 MFMA, different summation order); bf16 mode 2e-2 relative (operands
are rounded to bf16 before the matrix cores, accumulation is f32).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speedplusbaseline_amd import _lib as L  # noqa: E402
from speedplusbaseline_amd import ops  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
EPS = 1e-5


def relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rt(x, dt):
    """round-trip through the storage dtype (what the kernels read)"""
    return x.to(dt).double()


def act_fn(u, act, slope=0.2):
    if act == L.ACT_RELU: return F.relu(u)
    if act == L.ACT_RELU6: return F.relu6(u)
    if act == L.ACT_LEAKY: return F.leaky_relu(u, slope)
    return u


def bn_train(z2d, gamma, beta):
    mean = z2d.mean(0); var = z2d.var(0, unbiased=False)
    xhat = (z2d - mean) / torch.sqrt(var + EPS)
    return xhat * gamma + beta, xhat


def sums_of(z2d, R, dev):
    """[R][2][C] replicas that add up to (sum z, sum z^2)"""
    C = z2d.shape[1]
    s = torch.stack([z2d.sum(0), (z2d * z2d).sum(0)])  # [2][C]
    w = torch.rand(R, 1, 1, dtype=torch.float64) + 0.1
    w = w / w.sum()
    return (w * s.unsqueeze(0)).float().contiguous().to(dev)


def test_trread_semantics(device):
    """pins ds_read_b64_tr_b16: lane i of a 16-lane group gets column i of the 4x16 block whose row r is supplied
    by lanes 4r..4r+3 (4 consecutive elements each)."""
    inp = torch.arange(4096, dtype=torch.int16, device=device)
    out = torch.zeros(256, dtype=torch.int16, device=device)
    ops.debug_trread(inp, out)
    torch.cuda.synchronize()
    out = out.cpu().view(64, 4).long()
    exp = torch.zeros(64, 4, dtype=torch.long)
    for l in range(64):
        q, i = l // 16, l % 16
        for j in range(4):
            # block row j is held by lanes 4j..4j+3 of the group; column i lives in lane 4j + i//4, element i%4
            src_lane = q * 16 + 4 * j + i // 4
            exp[l, j] = src_lane * 4 + i % 4
    assert torch.equal(out, exp), (out[:20], exp[:20])


@pytest.fixture(params=["default", "lds_dma", "tiled"])
def gemm_variant(request):
    """the pointwise-GEMM kernels: the default dispatch (one-shot kernel for medium reductions with a narrow output on the 28x28 /
    14x14 maps, split-K-over-waves kernel for small M with a long reduction, the 128 x 128 tile kernel for a wide output behind a long
    reduction on the 7x7 maps, the register-prefetch tiled kernel otherwise), the LDS-DMA ring (small M, bf16, K >= 64) and the tiled kernel alone.
    "default" runs on the product library (which has no knobs); the forced variants on the tuning build (include/spb_hip_tuning.h)"""
    if request.param == "default":
        yield request.param
        return
    with L.tuning():
        yield from _gemm_variant_tuned(request)


def _gemm_variant_tuned(request):
    L.lib().spb_debug_set_gemm_dma(1 if request.param == "lds_dma" else 0)
    L.lib().spb_debug_set_gemm_sk(0 if request.param != "default" else 1, 0, 0)
    L.lib().spb_debug_set_gemm_os(0 if request.param != "default" else 1, 0, 0, 0)
    L.lib().spb_debug_set_gemm_big(0 if request.param != "default" else 1, 0, 0)
    L.lib().spb_debug_set_gemm_rs(0 if request.param != "default" else 1, 0)
    L.lib().spb_debug_set_wgrad_tile(1 if request.param == "tiled" else 0, 0)   # the weight gradient's 128-wide tiles (measured slower in the step; tuning build only)
    yield request.param
    L.lib().spb_debug_set_wgrad_tile(0, 0)
    L.lib().spb_debug_set_gemm_dma(0)
    L.lib().spb_debug_set_gemm_sk(1, 0, 0)
    L.lib().spb_debug_set_gemm_os(1, 0, 0, 0)
    L.lib().spb_debug_set_gemm_big(1, 0, 0)
    L.lib().spb_debug_set_gemm_rs(1, 0)


@pytest.fixture(params=["auto", "rows"])
def dw_variant(request):
    """both depthwise kernel families: the default choice (plane kernels on maps up to 28 columns wide; product library) and the row-unit
    kernels everywhere (tuning build)"""
    if request.param == "auto":
        yield request.param
        return
    with L.tuning():
        L.lib().spb_debug_set_dw_mode(0)
        yield request.param
        L.lib().spb_debug_set_dw_mode(1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,K,N,act,R", [(300, 24, 144, L.ACT_RELU6, 1), (1000, 144, 32, L.ACT_RELU6, 3),
                                          (257, 320, 1024, L.ACT_NONE, 1), (2352, 96, 64, L.ACT_RELU, 2),
                                          (129, 16, 96, L.ACT_NONE, 8),
                                          # project convolutions of the 14x14 / 7x7 maps at bs=48 and the ConvDw pointwise layers
                                          (9408, 384, 64, L.ACT_RELU6, 1), (9408, 576, 96, L.ACT_RELU6, 1), (2352, 960, 160, L.ACT_RELU6, 1),
                                          (2352, 960, 320, L.ACT_RELU6, 1), (2352, 1280, 1024, L.ACT_RELU, 1), (2349, 200, 72, L.ACT_LEAKY, 2),
                                          # one-shot kernel (gemm_os.hip): 28x28 project, ragged rows / reduction / columns
                                          (37632, 192, 32, L.ACT_RELU6, 8), (4100, 200, 72, L.ACT_LEAKY, 2), (4099, 168, 40, L.ACT_RELU, 1)])
def test_pw_gemm_fwd(device, gemm_variant, dt, M, K, N, act, R):
    torch.manual_seed(M + K + N)
    zin = rt(torch.randn(M, K, dtype=torch.float64) * 1.5 + 0.3, dt)
    gamma = torch.rand(K, dtype=torch.float64) + 0.5; beta = torch.randn(K, dtype=torch.float64) * 0.3
    W = rt(torch.randn(N, K, dtype=torch.float64) / math.sqrt(K), dt)
    u, _ = bn_train(zin, gamma, beta)
    a = act_fn(u, act)
    y = a @ W.t()
    pro = ops.bnref(K, sums=sums_of(zin, R, device), gamma=gamma.float().to(device), beta=beta.float().to(device), n=M,
                    R=R, act=act, slope=0.2)
    Y = torch.empty(M, N, dtype=dt, device=device)
    oR = 4
    osums = torch.zeros(oR, 2, N, dtype=torch.float32, device=device)
    ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y, pro, 1, 1, osums=osums, oR=oR)
    torch.cuda.synchronize()
    assert relerr(Y, y) < TOL[dt]
    ys = Y.double().cpu()
    s = osums.double().cpu().sum(0)
    assert relerr(s[0], ys.sum(0)) < 1e-4 and relerr(s[1], (ys * ys).sum(0)) < 1e-4
    # plain epilogue with bias + relu
    bias = torch.randn(N, dtype=torch.float32, device=device)
    Y2 = torch.empty(M, N, dtype=dt, device=device)
    ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y2, pro, 1, 0, bias=bias, out_act=L.ACT_RELU, out_scale=1.0)
    torch.cuda.synchronize()
    assert relerr(Y2, F.relu(y + bias.double().cpu())) < TOL[dt]


def _composite(M, K, N, act1, act2, dt, seed):
    """z_in -> a=act1(bn1) -> z=a W^T -> o=act2(bn2(z)); returns tensors + autograd grads (float64, CPU)"""
    torch.manual_seed(seed)
    zin = rt(torch.randn(M, K, dtype=torch.float64) + 0.2, dt).requires_grad_(True)
    g1 = (torch.rand(K, dtype=torch.float64) + 0.5); b1 = torch.randn(K, dtype=torch.float64) * 0.3
    g2 = (torch.rand(N, dtype=torch.float64) + 0.5).requires_grad_(True); b2 = (torch.randn(N, dtype=torch.float64) * 0.3).requires_grad_(True)
    W = rt(torch.randn(N, K, dtype=torch.float64) / math.sqrt(K), dt).requires_grad_(True)
    u1, xh1 = bn_train(zin, g1, b1); u1.retain_grad()
    a = act_fn(u1, act1)
    z = a @ W.t()
    zq = rt(z.detach(), dt)  # the stored z the kernels see
    z = z + (zq - z).detach()
    u2, xh2 = bn_train(z, g2, b2); u2.retain_grad()
    o = act_fn(u2, act2)
    Rm = torch.randn(M, N, dtype=torch.float64)
    (o * Rm).sum().backward()
    return dict(zin=zin, g1=g1, b1=b1, g2=g2, b2=b2, W=W, u1=u1, xh1=xh1, z=zq, u2=u2, xh2=xh2, a=a.detach())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,K,N,act1,act2", [(300, 24, 144, L.ACT_RELU6, L.ACT_RELU6), (513, 144, 32, L.ACT_RELU6, L.ACT_NONE),
                                             (260, 96, 64, L.ACT_NONE, L.ACT_LEAKY), (200, 1024, 1024, L.ACT_RELU, L.ACT_RELU),
                                             # expand convolutions of the 14x14 / 7x7 maps at bs=48 (the input gradient reduces over N)
                                             (9408, 64, 384, L.ACT_NONE, L.ACT_RELU6), (9408, 96, 576, L.ACT_NONE, L.ACT_RELU6),
                                             (2352, 160, 960, L.ACT_NONE, L.ACT_RELU6), (2352, 320, 1024, L.ACT_NONE, L.ACT_RELU),
                                             (2352, 1024, 1024, L.ACT_RELU, L.ACT_RELU), (2349, 72, 200, L.ACT_RELU6, L.ACT_LEAKY),
                                             # one-shot kernel (gemm_os.hip): 28x28 expand, ragged shapes
                                             (37632, 32, 192, L.ACT_NONE, L.ACT_RELU6), (4100, 72, 200, L.ACT_RELU6, L.ACT_LEAKY), (4099, 40, 168, L.ACT_NONE, L.ACT_RELU),
                                             # project convolutions, one ragged in every axis
                                             (9408, 384, 64, L.ACT_RELU6, L.ACT_NONE), (2352, 960, 160, L.ACT_RELU6, L.ACT_NONE), (1003, 200, 40, L.ACT_RELU6, L.ACT_NONE)])
def test_pw_gemm_bwd(device, gemm_variant, dt, M, K, N, act1, act2):
    c = _composite(M, K, N, act1, act2, dt, seed=M * 7 + N)
    dev = device
    g2 = rt(c["u2"].grad, dt)  # g = dL/d(bn2 output) with act2' applied by autograd
    # NB: bsums must be computed from the *stored* g for the chain to be self-consistent
    bs = torch.stack([g2.sum(0), (g2 * c["xh2"].detach()).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(N, sums=sums_of(c["z"], 1, dev), gamma=c["g2"].detach().float().to(dev), beta=c["b2"].detach().float().to(dev),
                    bsums=bs, n=M, act=act2, slope=0.2)
    epi = ops.bnref(K, sums=sums_of(c["zin"].detach(), 2, dev), gamma=c["g1"].float().to(dev), beta=c["b1"].float().to(dev),
                    n=M, R=2, act=act1, slope=0.2)
    Wt = c["W"].detach().t().contiguous().to(dt).to(dev)
    G1 = torch.empty(M, K, dtype=dt, device=dev)
    osums = torch.zeros(2, 2, K, dtype=torch.float32, device=dev)
    res = rt(torch.randn(M, K, dtype=torch.float64) * 0.1, dt)
    ops.pwconv_gemm(g2.to(dt).to(dev), Wt, G1, pro, 2, 2, A2=c["z"].to(dt).to(dev), Zout=c["zin"].detach().to(dt).to(dev),
                    res=res.to(dt).to(dev), osums=osums, epi=epi, oR=2)
    torch.cuda.synchronize()
    # expected: autograd's dL/du1 plus the injected residual gradient routed through act1'
    u1 = c["u1"].detach()
    if act1 == L.ACT_RELU6: m1 = ((u1 > 0) & (u1 < 6)).double()
    elif act1 == L.ACT_RELU: m1 = (u1 > 0).double()
    elif act1 == L.ACT_LEAKY: m1 = torch.where(u1 > 0, 1.0, 0.2).double()
    else: m1 = torch.ones_like(u1)
    exp = c["u1"].grad + res * m1
    tol = TOL[dt] * (3 if dt == torch.bfloat16 else 1)
    assert relerr(G1, exp) < tol
    gs = G1.double().cpu()
    s = osums.double().cpu().sum(0)
    assert relerr(s[0], gs.sum(0)) < 1e-3 and relerr(s[1], (gs * c["xh1"].detach()).sum(0)) < 1e-3
    # weight gradient
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ops.pwconv_wgrad(g2.to(dt).to(dev), c["zin"].detach().to(dt).to(dev), dW, pro, epi, Zn=c["z"].to(dt).to(dev))
    torch.cuda.synchronize()
    assert relerr(dW, c["W"].grad) < tol
    # plain dgrad (no output-side BN): dA = dz W, scaled
    P = torch.empty(M, K, dtype=dt, device=dev)
    ops.pwconv_gemm(g2.to(dt).to(dev), Wt, P, pro, 2, 0, A2=c["z"].to(dt).to(dev), out_scale=-0.5)
    torch.cuda.synchronize()
    # reconstruct dz in float64 from the same definition
    xh2 = c["xh2"].detach(); n = float(M)
    var2 = c["z"].var(0, unbiased=False)
    dz = c["g2"].detach() / torch.sqrt(var2 + EPS) * (g2 - g2.sum(0) / n - xh2 * (g2 * xh2).sum(0) / n)
    assert relerr(P, -0.5 * (dz @ c["W"].detach())) < tol


@pytest.mark.parametrize("M,K,N,act1,act2,with_res,mat", [
    (1000, 32, 16, L.ACT_RELU6, L.ACT_NONE, False, False),   # 32 -> 16 project @112
    (777, 16, 96, L.ACT_NONE, L.ACT_RELU6, True, True),      # 16 -> 96 expand, materialised input, skip gradient
    (500, 96, 24, L.ACT_RELU6, L.ACT_NONE, False, False),    # 96 -> 24
    (1234, 24, 144, L.ACT_NONE, L.ACT_RELU6, True, False),   # 24 -> 144
    (901, 144, 24, L.ACT_RELU6, L.ACT_NONE, False, False),   # 144 -> 24: two K splits
    (4133, 144, 32, L.ACT_RELU6, L.ACT_NONE, False, False),
    # the production sizes at bs=48 (BASELINE configs[1]): 112x112 and 56x56 maps, M = 602 112 and 150 528 rows
    (602112, 16, 96, L.ACT_NONE, L.ACT_RELU6, False, False),  # block 2 expand 16 -> 96 @112
    (602112, 32, 16, L.ACT_RELU6, L.ACT_NONE, False, False),  # block 1 project 32 -> 16 @112
    (150528, 144, 24, L.ACT_RELU6, L.ACT_NONE, False, False), # block 3 project 144 -> 24 @56
    (150528, 24, 144, L.ACT_NONE, L.ACT_RELU6, True, True)])  # block 3 expand 24 -> 144 @56, skip gradient, materialised input
def test_pw_bwd_fused(device, M, K, N, act1, act2, with_res, mat):
    """fused dgrad+wgrad (bf16 only) against the same float64 composite as the two-kernel path"""
    dt = torch.bfloat16
    c = _composite(M, K, N, act1, act2, dt, seed=M + K + N)
    dev = device
    g2 = rt(c["u2"].grad, dt)
    bs = torch.stack([g2.sum(0), (g2 * c["xh2"].detach()).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(N, sums=sums_of(c["z"], 1, dev), gamma=c["g2"].detach().float().to(dev), beta=c["b2"].detach().float().to(dev),
                    bsums=bs, n=M, act=act2, slope=0.2)
    epi = ops.bnref(K, sums=sums_of(c["zin"].detach(), 2, dev), gamma=c["g1"].float().to(dev), beta=c["b1"].float().to(dev),
                    n=M, R=2, act=act1, slope=0.2)
    Wt = c["W"].detach().t().contiguous().to(dt).to(dev)
    zin = c["zin"].detach().to(dt).to(dev)
    if mat:   # the conv input is a materialised tensor (bn + act already applied, as after a residual block)
        X = rt(c["a"].detach(), dt).to(dt).to(dev); pro_a = ops.bnref(K)
    else:
        X = zin; pro_a = epi
    res = rt(torch.randn(M, K, dtype=torch.float64) * 0.1, dt) if with_res else None
    G1 = torch.empty(M, K, dtype=dt, device=dev)
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    osums = torch.zeros(2, 2, K, dtype=torch.float32, device=dev)
    ran = ops.pwconv_bwd_fused(g2.to(dt).to(dev), c["z"].to(dt).to(dev), Wt, X, zin, G1, dW, osums, pro, pro_a, epi,
                               res=None if res is None else res.to(dt).to(dev), oR=2)
    assert ran, "no fused instance for this shape"
    torch.cuda.synchronize()
    u1 = c["u1"].detach()
    if act1 == L.ACT_RELU6: m1 = ((u1 > 0) & (u1 < 6)).double()
    else: m1 = torch.ones_like(u1)
    exp = c["u1"].grad + (res * m1 if with_res else 0)
    tol = TOL[dt] * 3
    assert relerr(G1, exp) < tol
    gs = G1.double().cpu()
    s = osums.double().cpu().sum(0)
    assert relerr(s[0], gs.sum(0)) < 1e-3 and relerr(s[1], (gs * c["xh1"].detach()).sum(0)) < 2e-3
    assert relerr(dW, c["W"].grad) < tol * (4 if mat else 1)   # materialised a is itself rounded to bf16


@pytest.mark.parametrize("rs", [1, 0])
@pytest.mark.parametrize("M,K,N", [(9408, 64, 384), (9408, 96, 576), (37632, 32, 192), (4100, 96, 384), (4099, 64, 192), (300, 24, 144)])
@L.tuned
def test_pw_gemm_row_slab_forward_and_residual_join(device, rs, M, K, N):
    """expand convolutions of the 28x28 / 14x14 maps (short reduction, wide output): the row-slab kernel (gemm_rs.hip) and the tiled kernel,
    plain prologue and the residual join of pro_mode 3 (a = bn(A) + bn2(A2); the launch also writes the joined block output)"""
    dt = torch.bfloat16
    L.lib().spb_debug_set_gemm_rs(rs, 0)
    try:
        torch.manual_seed(M + N)
        zin = rt(torch.randn(M, K, dtype=torch.float64) * 1.5 + 0.3, dt)
        res = rt(torch.randn(M, K, dtype=torch.float64), dt)
        gamma = torch.rand(K, dtype=torch.float64) + 0.5; beta = torch.randn(K, dtype=torch.float64) * 0.3
        W = rt(torch.randn(N, K, dtype=torch.float64) / math.sqrt(K), dt)
        u, _ = bn_train(zin, gamma, beta)
        for join in (False, True):
            act = L.ACT_NONE if join else L.ACT_RELU6
            a = rt(u + res, dt) if join else act_fn(u, act)        # the joined operand is what the launch materialises (rounded)
            y = a @ W.t()
            pro = ops.bnref(K, sums=sums_of(zin, 2, device), gamma=gamma.float().to(device), beta=beta.float().to(device), n=M, R=2, act=act)
            Y = torch.empty(M, N, dtype=dt, device=device)
            osums = torch.zeros(4, 2, N, dtype=torch.float32, device=device)
            if join:
                Ym = torch.zeros(M, K, dtype=dt, device=device)
                ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y, pro, 3, 1, A2=res.to(dt).to(device), osums=osums, oR=4, Ymat=Ym)
            else:
                ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y, pro, 1, 1, osums=osums, oR=4)
            torch.cuda.synchronize()
            assert relerr(Y, y) < TOL[dt], (join, relerr(Y, y))
            ys = Y.double().cpu(); s = osums.double().cpu().sum(0)
            assert relerr(s[0], ys.sum(0)) < 1e-4 and relerr(s[1], (ys * ys).sum(0)) < 1e-4
            if join:
                assert relerr(Ym, u + res) < TOL[dt]
    finally:
        L.lib().spb_debug_set_gemm_rs(1, 0)


@pytest.mark.parametrize("M,K,N,join", [(4099, 32, 16, False), (4100, 16, 96, False), (5000, 96, 24, False), (4099, 24, 144, False), (4097, 144, 24, False),
                                         (4100, 144, 32, False), (4099, 24, 144, True), (150528, 24, 144, True), (150528, 24, 144, False), (200704, 16, 96, False), (200704, 32, 16, False), (150528, 96, 24, False),
                                         (150528, 144, 24, False)])
def test_pw_gemm_streaming_forward_large_maps(device, M, K, N, join):
    """the 1x1 forward convolutions of the 112x112 / 56x56 maps on the streaming kernel (gemm_st.hip, round 5): every (K, N) of MobileNetV2
    blocks 1-4, plain prologue (BatchNorm + ReLU6 with 8 statistic replicas) and the residual join (a = bn(A) + bn2(A2), written out as Ymat),
    ragged row counts, against float64 and against the tiled kernel.  Small M runs on the tuning build (the dispatch threshold is a knob);
    M >= 100000 on the product library as the KRN plan launches it."""
    dt = torch.bfloat16
    small = M < 100000
    def run(st_on):
        torch.manual_seed(M + N)
        zin = rt(torch.randn(M, K, dtype=torch.float64) * 1.5 + 0.3, dt)
        res = rt(torch.randn(M, K, dtype=torch.float64), dt)
        gamma = torch.rand(K, dtype=torch.float64) + 0.5; beta = torch.randn(K, dtype=torch.float64) * 0.3
        W = rt(torch.randn(N, K, dtype=torch.float64) / math.sqrt(K), dt)
        u, _ = bn_train(zin, gamma, beta)
        act = L.ACT_NONE if join else L.ACT_RELU6
        a = rt(u + res, dt) if join else act_fn(u, act)
        y = a @ W.t()
        if small or not st_on:
            L.lib().spb_debug_set_gemm_st(3 if st_on else 0, 1000 if small else 100000, 0)      # 3: with the long-reduction instances
        pro = ops.bnref(K, sums=sums_of(zin, 8, device), gamma=gamma.float().to(device), beta=beta.float().to(device), n=M, R=8, act=act)
        Y = torch.empty(M, N, dtype=dt, device=device)
        osums = torch.zeros(8, 2, N, dtype=torch.float32, device=device)
        Ym = torch.zeros(M, K, dtype=dt, device=device) if join else None
        if join:
            ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y, pro, 3, 1, A2=res.to(dt).to(device), osums=osums, oR=8, Ymat=Ym)
        else:
            ops.pwconv_gemm(zin.to(dt).to(device), W.to(dt).to(device), Y, pro, 1, 1, osums=osums, oR=8)
        torch.cuda.synchronize()
        return Y, osums.double().cpu().sum(0), Ym, y, u + res
    if small:
        with L.tuning():
            Y, s, Ym, y, joined = run(True)
            Yt, st_, _, _, _ = run(False)
            L.lib().spb_debug_set_gemm_st(1, 100000, 0)
    else:
        Y, s, Ym, y, joined = run(True)
        with L.tuning():
            Yt, st_, _, _, _ = run(False)
            L.lib().spb_debug_set_gemm_st(1, 100000, 0)
    assert relerr(Y, y) < TOL[dt], relerr(Y, y)
    assert relerr(Y, Yt) < 2e-3                                           # vs the tiled kernel: reduction order only
    ys = Y.double().cpu()
    assert relerr(s[0], ys.sum(0)) < 1e-4 and relerr(s[1], (ys * ys).sum(0)) < 1e-4
    assert relerr(s[0], st_[0]) < 1e-3 and relerr(s[1], st_[1]) < 1e-3
    if join:
        assert relerr(Ym, joined) < TOL[dt]


@pytest.mark.parametrize("M,K,N", [(9408, 384, 64), (9408, 576, 96), (37632, 192, 32), (4100, 384, 96), (9408, 192, 64)])
@L.tuned
def test_pw_gemm_row_slab_input_gradient(device, M, K, N):
    """project convolutions of the 28x28 / 14x14 maps, input gradient without a residual (what the KRN plan launches): the row-slab
    kernel against float64 autograd and against the tiled kernel (same element arithmetic up to the order of the MFMA reduction)"""
    dt = torch.bfloat16
    c = _composite(M, K, N, L.ACT_RELU6, L.ACT_NONE, dt, seed=M * 3 + N)
    dev = device
    g2 = rt(c["u2"].grad, dt)
    bs = torch.stack([g2.sum(0), (g2 * c["xh2"].detach()).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(N, sums=sums_of(c["z"], 1, dev), gamma=c["g2"].detach().float().to(dev), beta=c["b2"].detach().float().to(dev),
                    bsums=bs, n=M, act=L.ACT_NONE)
    epi = ops.bnref(K, sums=sums_of(c["zin"].detach(), 2, dev), gamma=c["g1"].float().to(dev), beta=c["b1"].float().to(dev), n=M, R=2, act=L.ACT_RELU6)
    Wt = c["W"].detach().t().contiguous().to(dt).to(dev)
    outs = []
    for rs in (1, 0):
        L.lib().spb_debug_set_gemm_rs(rs, 0)
        G1 = torch.empty(M, K, dtype=dt, device=dev)
        osums = torch.zeros(2, 2, K, dtype=torch.float32, device=dev)
        ops.pwconv_gemm(g2.to(dt).to(dev), Wt, G1, pro, 2, 2, A2=c["z"].to(dt).to(dev), Zout=c["zin"].detach().to(dt).to(dev), osums=osums, epi=epi, oR=2)
        torch.cuda.synchronize()
        outs.append((G1, osums.sum(0)))
    L.lib().spb_debug_set_gemm_rs(1, 0)
    G1, sm = outs[0]
    assert relerr(G1, c["u1"].grad) < TOL[dt] * 3
    assert relerr(G1, outs[1][0]) < 2e-3                                   # vs the tiled kernel: reduction order only
    gs = G1.double().cpu(); sm = sm.double().cpu()
    assert relerr(sm[0], gs.sum(0)) < 1e-3 and relerr(sm[1], (gs * c["xh1"].detach()).sum(0)) < 1e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,C,stride,act", [(2, 14, 96, 1, L.ACT_RELU6), (3, 15, 32, 2, L.ACT_RELU6), (2, 7, 1280, 1, L.ACT_NONE),
                                              (2, 28, 144, 2, L.ACT_RELU),
                                              # the plane-kernel layers of KRN at bs=48 (dwconv_plane.hip), plus ragged image groups / segments
                                              (48, 14, 384, 1, L.ACT_RELU6), (48, 7, 960, 1, L.ACT_RELU6), (48, 14, 576, 2, L.ACT_RELU6),
                                              (12, 28, 192, 1, L.ACT_RELU6), (7, 28, 64, 2, L.ACT_RELU6), (5, 7, 320, 1, L.ACT_RELU),
                                              (3, 9, 40, 1, L.ACT_RELU), (2, 27, 24, 2, L.ACT_RELU6),
                                              # the large maps (bf16: LDS-tile kernels of dwconv_tile.hip): full and ragged 16 x 8 tiles, half channel chunks
                                              (2, 56, 144, 1, L.ACT_RELU6), (1, 112, 96, 2, L.ACT_RELU6),
                                              (3, 56, 40, 2, L.ACT_RELU6), (1, 60, 24, 1, L.ACT_RELU), (2, 35, 16, 2, L.ACT_NONE)])
def test_dwconv(device, dw_variant, dt, B, H, C, stride, act):
    torch.manual_seed(B * H + C)
    dev = device
    zin = rt(torch.randn(B, C, H, H, dtype=torch.float64) + 0.1, dt).requires_grad_(True)
    g1 = torch.rand(C, dtype=torch.float64) + 0.5; b1 = torch.randn(C, dtype=torch.float64) * 0.2
    g2 = torch.rand(C, dtype=torch.float64) + 0.5; b2 = torch.randn(C, dtype=torch.float64) * 0.2
    Wd = (torch.randn(C, 1, 3, 3, dtype=torch.float64) * 0.3).float().double().requires_grad_(True)

    def bn4(z, g, b):
        mean = z.mean((0, 2, 3), keepdim=True); var = z.var((0, 2, 3), unbiased=False, keepdim=True)
        xh = (z - mean) / torch.sqrt(var + EPS)
        return xh * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), xh

    u1, xh1 = bn4(zin, g1, b1); u1.retain_grad()
    a = act_fn(u1, act)
    z = F.conv2d(a, Wd, stride=stride, padding=1, groups=C)
    zq = rt(z.detach(), dt); z = z + (zq - z).detach()
    u2, xh2 = bn4(z, g2, b2); u2.retain_grad()
    o = F.relu6(u2)
    (o * torch.randn_like(o)).sum().backward()
    OH = z.shape[2]
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    n_in, n_out = B * H * H, B * OH * OH
    zin2d = nhwc(zin).view(-1, C); z2d = nhwc(zq).view(-1, C)
    pro_in = ops.bnref(C, sums=sums_of(zin2d, 2, dev), gamma=g1.float().to(dev), beta=b1.float().to(dev), n=n_in, R=2, act=act)
    X = nhwc(zin).to(dt).to(dev)
    Y = torch.empty(B, OH, OH, C, dtype=dt, device=dev)
    osums = torch.zeros(3, 2, C, dtype=torch.float32, device=dev)
    Wdev = Wd.detach().float().to(dev).contiguous()
    ops.dwconv_fwd(X, Wdev, Y, pro_in, stride, osums=osums, oR=3)
    torch.cuda.synchronize()
    assert relerr(Y, nhwc(z)) < TOL[dt]
    ys = Y.double().cpu().view(-1, C); s = osums.double().cpu().sum(0)
    assert relerr(s[0], ys.sum(0)) < 1e-4 and relerr(s[1], (ys * ys).sum(0)) < 1e-4
    # backward
    g2s = rt(nhwc(u2.grad), dt)
    xh2n = nhwc(xh2).view(-1, C)
    bs = torch.stack([g2s.view(-1, C).sum(0), (g2s.view(-1, C) * xh2n).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(C, sums=sums_of(z2d, 1, dev), gamma=g2.float().to(dev), beta=b2.float().to(dev), bsums=bs, n=n_out,
                    act=L.ACT_RELU6)
    G1 = torch.empty(B, H, H, C, dtype=dt, device=dev)
    os2 = torch.zeros(2, 2, C, dtype=torch.float32, device=dev)
    ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, G1, pro, stride, (H, H), epi=pro_in, Zout=X,
                     osums=os2, oR=2)
    torch.cuda.synchronize()
    tol = TOL[dt] * (3 if dt == torch.bfloat16 else 1)
    assert relerr(G1, nhwc(u1.grad)) < tol
    gs = G1.double().cpu().view(-1, C); s = os2.double().cpu().sum(0)
    assert relerr(s[0], gs.sum(0)) < 1e-3 and relerr(s[1], (gs * nhwc(xh1).view(-1, C)).sum(0)) < 1e-3
    dW = torch.zeros(C, 1, 3, 3, dtype=torch.float32, device=dev)
    ops.dwconv_wgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), X, Wdev, dW, pro, pro_in, stride)
    torch.cuda.synchronize()
    assert relerr(dW, Wd.grad) < tol
    # plain dgrad
    P = torch.empty(B, H, H, C, dtype=dt, device=dev)
    ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, P, pro, stride, (H, H))
    torch.cuda.synchronize()
    da = nhwc(_dw_da(a, Wd, z, g2, g2s, xh2, C, stride))
    assert relerr(P, da) < tol
    # fused: input gradient and weight gradient in one pass, with and without the activation/BN-sum epilogue
    for with_epi in (True, False):
        Gf = torch.empty(B, H, H, C, dtype=dt, device=dev)
        dWf = torch.zeros(C, 1, 3, 3, dtype=torch.float32, device=dev)
        osf = torch.zeros(2, 2, C, dtype=torch.float32, device=dev)
        if with_epi:
            ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, Gf, pro, stride, (H, H), epi=pro_in, Zout=X,
                             osums=osf, oR=2, dW=dWf)
        else:
            ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, Gf, pro, stride, (H, H), dW=dWf, Xin=X,
                             pro_in=pro_in)
        torch.cuda.synchronize()
        assert relerr(dWf, Wd.grad) < tol
        assert relerr(Gf, nhwc(u1.grad) if with_epi else da) < tol
        if with_epi:
            # (the fused instance is the row-unit kernel with f32 tap weights, G1 may come from the tile kernel with bf16 taps: the
            # tensors agree to the bf16 tolerance; each kernel's sums are held to ITS OWN output)
            gf = Gf.double().cpu().view(-1, C); sf = osf.double().cpu().sum(0)
            assert relerr(Gf, G1) < tol
            assert relerr(sf[0], gf.sum(0)) < 1e-3 and relerr(sf[1], (gf * nhwc(xh1).view(-1, C)).sum(0)) < 1e-3


@pytest.mark.parametrize("B,H,C,stride", [(4, 14, 96, 1), (2, 28, 64, 2), (2, 56, 48, 1), (1, 112, 32, 2)])   # plane, row-unit and tile kernels
def test_dwconv_dgrad_publishes_entry_word(device, B, H, C, stride):
    """spb_dw_args_t::entry_flag (include/spb_hip.h): the input-gradient launch stores entry_val to the device word before anything
    else -- the KRN plan's stream fork without an event (csrc/krn_plan.hip, fork_gate_kernel) -- and computes what it computes without it."""
    dt, dev = torch.bfloat16, device
    torch.manual_seed(H + C)
    OH = (H + 2 - 3) // stride + 1
    G = torch.randn(B, OH, OH, C, device=dev).to(dt); Z = torch.randn(B, OH, OH, C, device=dev).to(dt)
    Wd = (torch.randn(C, 1, 3, 3, device=dev) * 0.3).contiguous()
    n = B * OH * OH
    z2 = Z.double().view(-1, C).cpu()
    g2 = G.double().view(-1, C).cpu()
    mean = z2.mean(0); var = z2.var(0, unbiased=False); xh = (z2 - mean) / torch.sqrt(var + EPS)
    bs = torch.stack([g2.sum(0), (g2 * xh).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(C, sums=sums_of(z2, 1, dev), gamma=torch.ones(C, device=dev), beta=torch.zeros(C, device=dev), bsums=bs, n=n,
                    act=L.ACT_RELU6)
    outs = []
    word = torch.zeros(4, dtype=torch.int32, device=dev)
    for flag in (None, word):
        P = torch.empty(B, H, H, C, dtype=dt, device=dev)
        ops.dwconv_dgrad(G, Z, Wd, P, pro, stride, (H, H), entry_flag=flag, entry_val=41 + stride)
        torch.cuda.synchronize()
        outs.append(P)
    assert word.tolist() == [41 + stride, 0, 0, 0]
    assert torch.equal(outs[0], outs[1])


@pytest.fixture(params=["rows", "tile"])
def dw_tile_dgrad(request):
    """stride-1 input gradient on the LDS-tile kernel (off by default: spb_debug_set_dw_tile bit 16) and on the default kernels"""
    if request.param == "rows":
        yield request.param
        return
    with L.tuning():
        L.lib().spb_debug_set_dw_tile(28, 1 << 16)
        yield request.param
        L.lib().spb_debug_set_dw_tile(28, 0)


@pytest.mark.parametrize("B,H,C,act", [(3, 28, 192, L.ACT_RELU6), (2, 56, 144, L.ACT_RELU6), (1, 60, 24, L.ACT_RELU), (2, 35, 40, L.ACT_NONE)])
def test_dwconv_input_gradient_tile_instance(device, dw_tile_dgrad, B, H, C, act):
    """dz rebuilt from (g, z), flipped taps, ReLU-type mask of the input side and its two BatchNorm-backward sums, against float64"""
    dt, dev, stride = torch.bfloat16, device, 1
    torch.manual_seed(B * H + C)
    zin = rt(torch.randn(B, C, H, H, dtype=torch.float64) + 0.1, dt).requires_grad_(True)
    g1 = torch.rand(C, dtype=torch.float64) + 0.5; b1 = torch.randn(C, dtype=torch.float64) * 0.2
    g2 = torch.rand(C, dtype=torch.float64) + 0.5; b2 = torch.randn(C, dtype=torch.float64) * 0.2
    Wd = rt(torch.randn(C, 1, 3, 3, dtype=torch.float64) * 0.3, dt).requires_grad_(True)      # bf16-representable taps: both kernels see the same weights

    def bn4(z, g, b):
        mean = z.mean((0, 2, 3), keepdim=True); var = z.var((0, 2, 3), unbiased=False, keepdim=True)
        xh = (z - mean) / torch.sqrt(var + EPS)
        return xh * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), xh

    u1, xh1 = bn4(zin, g1, b1); u1.retain_grad()
    a = act_fn(u1, act)
    z = F.conv2d(a, Wd, stride=stride, padding=1, groups=C)
    zq = rt(z.detach(), dt); z = z + (zq - z).detach()
    u2, xh2 = bn4(z, g2, b2); u2.retain_grad()
    (F.relu6(u2) * torch.randn_like(u2)).sum().backward()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    n = B * H * H
    pro_in = ops.bnref(C, sums=sums_of(nhwc(zin).view(-1, C), 2, dev), gamma=g1.float().to(dev), beta=b1.float().to(dev), n=n, R=2, act=act)
    g2s = rt(nhwc(u2.grad), dt)
    bs = torch.stack([g2s.view(-1, C).sum(0), (g2s.view(-1, C) * nhwc(xh2).view(-1, C)).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(C, sums=sums_of(nhwc(zq).view(-1, C), 1, dev), gamma=g2.float().to(dev), beta=b2.float().to(dev), bsums=bs, n=n, act=L.ACT_RELU6)
    X = nhwc(zin).to(dt).to(dev); Wdev = Wd.detach().float().to(dev).contiguous()
    G1 = torch.empty(B, H, H, C, dtype=dt, device=dev)
    os2 = torch.zeros(2, 2, C, dtype=torch.float32, device=dev)
    ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, G1, pro, stride, (H, H), epi=pro_in, Zout=X, osums=os2, oR=2)
    torch.cuda.synchronize()
    assert relerr(G1, nhwc(u1.grad)) < TOL[dt] * 3
    gs = G1.double().cpu().view(-1, C); s = os2.double().cpu().sum(0)
    assert relerr(s[0], gs.sum(0)) < 1e-3 and relerr(s[1], (gs * nhwc(xh1).view(-1, C)).sum(0)) < 1e-3
    P = torch.empty(B, H, H, C, dtype=dt, device=dev)
    ops.dwconv_dgrad(g2s.to(dt).to(dev), nhwc(zq).to(dt).to(dev), Wdev, P, pro, stride, (H, H))
    torch.cuda.synchronize()
    assert relerr(P, nhwc(_dw_da(a, Wd, z, g2, g2s, xh2, C, stride))) < TOL[dt] * 3


def _dw_da(a, Wd, z, g2, g2s_nhwc, xh2, C, stride):
    """d loss / d a for the depthwise conv, rebuilt in float64 from the BN-backward definition"""
    g = g2s_nhwc.permute(0, 3, 1, 2)
    n = g.numel() / C
    xh = xh2.detach()
    var = z.detach().var((0, 2, 3), unbiased=False, keepdim=True)
    dz = g2.view(1, -1, 1, 1) / torch.sqrt(var + EPS) * (g - g.sum((0, 2, 3), keepdim=True) / n -
                                                            xh * (g * xh).sum((0, 2, 3), keepdim=True) / n)
    ad = a.detach().requires_grad_(True)
    out = F.conv2d(ad, Wd.detach(), stride=stride, padding=1, groups=C)
    out.backward(dz)
    return ad.grad


@pytest.fixture(params=["tile", "tile16", "mfma", "scalar"])
def stem_variant(request):
    """bf16 stem: implicit GEMM on the matrix cores from an LDS tile (default), the same with its taps gathered from global memory, and the
    scalar kernels (the f32 mode always uses those)"""
    if request.param == "tile":         # the default: product library
        yield request.param
        return
    with L.tuning():
        L.lib().spb_debug_set_stem_mfma(0 if request.param == "scalar" else 1)
        L.lib().spb_debug_set_stem_tile((2 if request.param == "tile16" else 1) if request.param.startswith("tile") else 0)   # 2: 4-row forward bands
        L.lib().spb_debug_set_stem_wgrad_tile(16 if request.param == "tile16" else 8)   # 16 rows per workgroup: a ragged last band at 48 x 48
        yield request.param
        L.lib().spb_debug_set_stem_mfma(1)
        L.lib().spb_debug_set_stem_tile(1)
        L.lib().spb_debug_set_stem_wgrad_tile(8)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H", [(3, 32), (2, 44), (2, 48), (1, 224), (1, 448)])   # 44: ragged 16-pixel groups and a ragged last band of rows; 448: the tile of an 8-row band no longer fits, 4-row bands do
def test_stem(device, stem_variant, dt, B, H):
    torch.manual_seed(5)
    dev = device
    O = (H - 1) // 2 + 1
    x = torch.rand(B, 3, H, H, dtype=torch.float32).double()
    W = (torch.randn(32, 3, 3, 3) * 0.2).double().requires_grad_(True)
    g2 = torch.rand(32, dtype=torch.float64) + 0.5; b2 = torch.randn(32, dtype=torch.float64) * 0.2
    z = F.conv2d(x, W, stride=2, padding=1)
    zq = rt(z.detach(), dt); z = z + (zq - z).detach()
    mean = z.mean((0, 2, 3), keepdim=True); var = z.var((0, 2, 3), unbiased=False, keepdim=True)
    xh = (z - mean) / torch.sqrt(var + EPS)
    u = xh * g2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1); u.retain_grad()
    (F.relu6(u) * torch.randn_like(u)).sum().backward()
    Y = torch.empty(B, O, O, 32, dtype=dt, device=dev)
    osums = torch.zeros(2, 2, 32, dtype=torch.float32, device=dev)
    ops.stem_fwd(x.float().to(dev), W.detach().float().to(dev), Y, osums=osums, oR=2)
    torch.cuda.synchronize()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    assert relerr(Y, nhwc(z)) < TOL[dt]
    if dt == torch.bfloat16 and stem_variant != "scalar":
        # round 6: image and weights enter the matrix cores as hi + lo pairs -- the stored output is the ROUNDING of the exact convolution, not the
        # convolution of a rounded image (measured 1e-4: a few values on the other side of a rounding boundary; with rounded operands 3e-3)
        assert relerr(Y, nhwc(zq)) < 6e-4, relerr(Y, nhwc(zq))
    ys = Y.double().cpu().view(-1, 32)
    assert relerr(osums.double().cpu().sum(0)[0], ys.sum(0)) < 1e-4
    gs = rt(nhwc(u.grad), dt).view(-1, 32)
    bs = torch.stack([gs.sum(0), (gs * nhwc(xh).view(-1, 32)).sum(0)]).float().unsqueeze(0).contiguous().to(dev)
    pro = ops.bnref(32, sums=sums_of(nhwc(zq).view(-1, 32), 1, dev), gamma=g2.float().to(dev), beta=b2.float().to(dev), bsums=bs,
                    n=B * O * O, act=L.ACT_RELU6)
    dW = torch.zeros(32, 3, 3, 3, dtype=torch.float32, device=dev)
    ops.stem_wgrad(x.float().to(dev), gs.view(B, O, O, 32).to(dt).to(dev), nhwc(zq).to(dt).to(dev), pro, dW)
    torch.cuda.synchronize()
    assert relerr(dW, W.grad) < TOL[dt] * 3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B", [5, 48])
def test_head(device, dt, B):
    torch.manual_seed(B)
    dev = device
    J, HW, C = 22, 49, 64  # narrower C than KRN's 1024 keeps the CPU reference cheap; the kernel is generic in C
    z = rt(torch.randn(B, C, 7, 7, dtype=torch.float64), dt).requires_grad_(True)
    g1 = torch.rand(C, dtype=torch.float64) + 0.5; b1 = torch.randn(C, dtype=torch.float64) * 0.2
    W = rt(torch.randn(J, C, 7, 7, dtype=torch.float64) * 0.05, dt).requires_grad_(True)
    bias = torch.randn(J, dtype=torch.float64).float().double().requires_grad_(True)
    tgt = torch.rand(B, 2, J // 2, dtype=torch.float64).float().double()
    mean = z.mean((0, 2, 3), keepdim=True); var = z.var((0, 2, 3), unbiased=False, keepdim=True)
    xh = (z - mean) / torch.sqrt(var + EPS)
    u = xh * g1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1); u.retain_grad()
    out = F.conv2d(F.relu(u), W, bias).view(B, J)
    xc, yc = out[:, 0::2], out[:, 1::2]
    lx = sum(F.mse_loss(xc[:, i], tgt[:, 0, i]) for i in range(J // 2))
    ly = sum(F.mse_loss(yc[:, i], tgt[:, 1, i]) for i in range(J // 2))
    (lx + ly).backward()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    Z = nhwc(z).view(B, HW * C).to(dt).to(dev)
    Wp = torch.zeros(32, HW * C, dtype=dt, device=dev)
    Wp[:J] = W.detach().permute(0, 2, 3, 1).reshape(J, HW * C).to(dt).to(dev)
    pro = ops.bnref(C, sums=sums_of(nhwc(z).view(-1, C), 1, dev), gamma=g1.float().to(dev), beta=b1.float().to(dev), n=B * HW,
                    act=L.ACT_RELU)
    pred, scal, dout = ops.head_fwd(Z, Wp, bias.detach().float().to(dev), pro, J, HW, C, target=tgt.float().to(dev))
    torch.cuda.synchronize()
    assert relerr(pred, out) < TOL[dt]
    assert abs(float(scal[1]) - float(lx)) < TOL[dt] * max(1.0, float(lx)) * 3
    assert abs(float(scal[0]) - float(lx + ly)) < TOL[dt] * max(1.0, float(lx + ly)) * 3
    G = torch.empty(B, HW * C, dtype=dt, device=dev)
    osums = torch.zeros(1, 2, C, dtype=torch.float32, device=dev)
    dW = torch.zeros(J, C, 7, 7, dtype=torch.float32, device=dev)
    db = torch.zeros(J, dtype=torch.float32, device=dev)
    ops.head_bwd(Z, Wp, dout, G, osums, dW, db, pro, J, HW, C)
    torch.cuda.synchronize()
    tol = TOL[dt] * 3
    assert relerr(G.view(B, 7, 7, C), nhwc(u.grad)) < tol
    assert relerr(dW, W.grad) < tol
    assert relerr(db, bias.grad) < tol
    gs = G.double().cpu().view(-1, C)
    s = osums.double().cpu()[0]
    assert relerr(s[0], gs.sum(0)) < 1e-3 and relerr(s[1], (gs * nhwc(xh).view(-1, C)).sum(0)) < 1e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("poison", ["nan", "inf", "huge"])
def test_head_reports_a_diverged_network_as_nan(device, dt, poison):
    """the head's split-K sum is 64-bit fixed point (order-independent); a partial sum it cannot hold -- NaN, Inf, |v| >= 2^26 -- must not
    come out as a saturated FINITE prediction: the reference would show nan / inf there (park2019.py:139-162).  The next call on clean
    data is clean again (the poison word is reset by the last arriver)."""
    torch.manual_seed(3)
    B, J, HW, C = 6, 22, 49, 64
    Z = torch.randn(B, HW * C, dtype=torch.float32).to(dt).to(device)
    Wp = torch.zeros(32, HW * C, dtype=dt, device=device)
    Wp[:J] = (torch.randn(J, HW * C) * 0.05).to(dt).to(device)
    bias = torch.zeros(J, dtype=torch.float32, device=device)
    tgt = torch.rand(B, 2, J // 2, dtype=torch.float32, device=device)
    pro_sums = torch.zeros(1, 2, C, dtype=torch.float32, device=device); pro_sums[0, 1] = 1.0 - EPS    # mean 0, variance 1 - eps
    pro = ops.bnref(C, sums=pro_sums, gamma=torch.ones(C, device=device), beta=torch.zeros(C, device=device), n=1, act=L.ACT_NONE, moments=1)
    Zbad, Wbad = Z.clone(), Wp.clone()
    if poison == "nan":      # through a weight: the activation helpers map a NaN INPUT to 0 (v_med3 / v_min drop NaN operands), a NaN product survives
        Wbad[3, 100] = float("nan")
    else:
        Zbad[2, 777] = {"inf": float("inf"), "huge": 3.0e38}[poison]
    ws = torch.zeros(256 * B * 32, dtype=torch.float32, device=device)       # ONE reduction workspace for both calls
    pred, scal, _ = ops.head_fwd(Zbad, Wbad, bias, pro, J, HW, C, target=tgt, partial=ws)
    torch.cuda.synchronize()
    assert not torch.isfinite(pred).any() and not torch.isfinite(scal[0])
    assert not ws.any()                                                       # accumulator, ticket and poison word are zero again
    pred, scal, _ = ops.head_fwd(Z, Wp, bias, pro, J, HW, C, target=tgt, partial=ws)
    torch.cuda.synchronize()
    ref = Z.double().cpu() @ Wp[:J].double().cpu().t()
    assert torch.isfinite(pred).all() and relerr(pred, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_bn_apply_and_bwd_prep(device, dt):
    torch.manual_seed(11)
    dev = device
    B, H, C = 2, 14, 64
    z = rt(torch.randn(B, H, H, C, dtype=torch.float64), dt)
    r = rt(torch.randn(B, H, H, C, dtype=torch.float64), dt)
    g1 = torch.rand(C, dtype=torch.float64) + 0.5; b1 = torch.randn(C, dtype=torch.float64) * 0.2
    u, xh = bn_train(z.view(-1, C), g1, b1)
    bn = ops.bnref(C, sums=sums_of(z.view(-1, C), 2, dev), gamma=g1.float().to(dev), beta=b1.float().to(dev), n=B * H * H, R=2,
                   act=L.ACT_LEAKY, slope=0.2)
    # residual add
    Y = torch.empty(B, H, H, C, dtype=dt, device=dev)
    ops.bn_apply(z.to(dt).to(dev), Y, bn, res=r.to(dt).to(dev))
    torch.cuda.synchronize()
    exp = F.leaky_relu(u, 0.2).view(B, H, H, C) + r
    assert relerr(Y, exp) < TOL[dt]
    # reorg (RouterV2, park2019.py:74-80) into a wider concat buffer: router channels first
    cat = torch.zeros(B, H // 2, H // 2, 4 * C + 32, dtype=dt, device=dev)
    ops.bn_apply(z.to(dt).to(dev), cat, bn, ldc=4 * C + 32, coff=0, reorg=2)
    torch.cuda.synchronize()
    x2 = F.leaky_relu(u, 0.2).view(B, H, H, C).permute(0, 3, 1, 2)  # NCHW
    s = 2
    t = x2.reshape(B, C, H // s, s, H // s, s).transpose(3, 4).contiguous()
    t = t.view(B, C, H // s * H // s, s * s).transpose(2, 3).contiguous()
    t = t.view(B, C, s * s, H // s, H // s).transpose(1, 2).contiguous()
    t = t.view(B, s * s * C, H // s, H // s)
    assert relerr(cat[..., : 4 * C], t.permute(0, 2, 3, 1)) < TOL[dt]
    assert float(cat[..., 4 * C:].abs().max()) == 0.0
    # backward of the same mapping
    dcat = rt(torch.randn(B, H // 2, H // 2, 4 * C + 32, dtype=torch.float64), dt)
    G = torch.empty(B, H, H, C, dtype=dt, device=dev)
    osums = torch.zeros(2, 2, C, dtype=torch.float32, device=dev)
    ops.bn_bwd_prep(dcat.to(dt).to(dev), z.to(dt).to(dev), G, osums, bn, ldc=4 * C + 32, coff=0, reorg=2, oR=2)
    torch.cuda.synchronize()
    d_nchw = dcat[..., : 4 * C].permute(0, 3, 1, 2)  # [B, 4C, h, w]
    dx = torch.zeros(B, C, H, H, dtype=torch.float64)
    for i in range(2):
        for j in range(2):
            dx[:, :, i::2, j::2] = d_nchw[:, (i * 2 + j) * C:(i * 2 + j + 1) * C]
    mask = torch.where(u > 0, 1.0, 0.2).view(B, H, H, C)
    expg = dx.permute(0, 2, 3, 1) * mask
    assert relerr(G, expg) < TOL[dt]
    gs = G.double().cpu().view(-1, C); sm = osums.double().cpu().sum(0)
    assert relerr(sm[0], gs.sum(0)) < 1e-3 and relerr(sm[1], (gs * xh).sum(0)) < 1e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,C,coff,reorg", [(3, 7, 96, 32, 0), (48, 7, 1024, 256, 0), (5, 14, 64, 0, 2)])
@L.tuned
def test_bn_bwd_prep_row_parallel_kernel(device, dt, B, H, C, coff, reorg):
    """spb_bn_bwd_prep's row-parallel kernel (the default) at the two KRN call sites (the 1024-channel slice of the concat gradient,
    the un-reorg of the router slice) and a ragged channel count: against float64 and against the walking kernel it replaced"""
    torch.manual_seed(B * 100 + C)
    dev = device
    z = rt(torch.randn(B, H, H, C, dtype=torch.float64), dt)
    g1 = torch.rand(C, dtype=torch.float64) + 0.5; b1 = torch.randn(C, dtype=torch.float64) * 0.2
    u, xh = bn_train(z.view(-1, C), g1, b1)
    bn = ops.bnref(C, sums=sums_of(z.view(-1, C), 1, dev), gamma=g1.float().to(dev), beta=b1.float().to(dev), n=B * H * H, R=1, act=L.ACT_RELU)
    s = max(reorg, 1)
    ldc = coff + s * s * C + 16
    dcat = rt(torch.randn(B, H // s, H // s, ldc, dtype=torch.float64), dt)
    outs = []
    for rows in (1, 0):
        L.lib().spb_debug_set_bn_bwd_prep_rows(rows)
        G = torch.empty(B, H, H, C, dtype=dt, device=dev)
        osums = torch.zeros(2, 2, C, dtype=torch.float32, device=dev)
        ops.bn_bwd_prep(dcat.to(dt).to(dev), z.to(dt).to(dev), G, osums, bn, ldc=ldc, coff=coff, reorg=reorg, oR=2)
        torch.cuda.synchronize()
        outs.append((G, osums.sum(0)))
    L.lib().spb_debug_set_bn_bwd_prep_rows(1)
    if reorg:
        d_nchw = dcat[..., coff: coff + s * s * C].permute(0, 3, 1, 2)
        dx = torch.zeros(B, C, H, H, dtype=torch.float64)
        for i in range(s):
            for j in range(s):
                dx[:, :, i::s, j::s] = d_nchw[:, (i * s + j) * C:(i * s + j + 1) * C]
        dx = dx.permute(0, 2, 3, 1)
    else:
        dx = dcat[..., coff: coff + C]
    expg = dx * (u > 0).double().view(B, H, H, C)
    G, sm = outs[0]
    assert relerr(G, expg) < TOL[dt]
    assert torch.equal(G, outs[1][0])                                     # same element arithmetic as the walking kernel
    gs = G.double().cpu().view(-1, C); sm = sm.double().cpu()
    assert relerr(sm[0], gs.sum(0)) < 1e-3 and relerr(sm[1], (gs * xh).sum(0)) < 1e-3
    assert relerr(sm, outs[1][1].double().cpu()) < 1e-4


@pytest.mark.parametrize("kind", ["adamw", "adam", "rmsprop", "sgd"])
def test_optim_step_matches_torch(device, kind):
    """fused clip + update vs clip_grad_norm_ + torch.optim (reference build.py:60-78, trainer.py:97-98)"""
    torch.manual_seed(3)
    n = 10007
    p0 = torch.randn(n); steps = 3
    grads = [torch.randn(n) * (0.05 if i else 3.0) for i in range(steps)]
    p_ref = torch.nn.Parameter(p0.clone())
    mk = dict(adamw=lambda: torch.optim.AdamW([p_ref], lr=1e-2, betas=(0.9, 0.999), weight_decay=0.01),
              adam=lambda: torch.optim.Adam([p_ref], lr=1e-2, betas=(0.9, 0.999), weight_decay=5e-5),
              rmsprop=lambda: torch.optim.RMSprop([p_ref], lr=1e-2, alpha=0.9, weight_decay=5e-5),
              sgd=lambda: torch.optim.SGD([p_ref], lr=1e-2, momentum=0.9, weight_decay=5e-5))[kind]
    opt = mk()
    wd = 0.01 if kind == "adamw" else 5e-5
    p = p0.clone().to(device); m = torch.zeros(n, device=device); v = torch.zeros(n, device=device)
    sq = torch.zeros(1, device=device)
    for i, g in enumerate(grads):
        p_ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gd = g.clone().to(device)
        ops.grad_sqnorm(gd, sq)
        ops.optim_step(kind, p, gd, m=m, v=v, sqnorm=sq, lr=1e-2, beta1=0.9, beta2=0.9 if kind == "rmsprop" else 0.999,
                       weight_decay=wd, max_norm=1.0, step=i + 1, first_step=(i == 0))
    torch.cuda.synchronize()
    assert relerr(p, p_ref.detach()) < 1e-5


def test_stream_fork_orders_the_second_stream(device):
    """ops.StreamFork / spb_fork_streams (include/spb_hip.h): what `b.wait_stream(a)` does, without an event record on `a` -- a one-wave
    kernel on `a` stores a serial number, a one-wave gate kernel on `b` spins on it.  Stream `b` must see everything `a` held at the fork,
    complete: forty rounds of four 64 MB read-modify-write passes on `a`, a copy on `b` right behind the fork."""
    fork = ops.StreamFork()
    a, b = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    x = torch.zeros(16 << 20, dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    seen = []
    for i in range(40):
        with torch.cuda.stream(a):
            for _ in range(4):
                x.add_(1.0)
            fork(b)                     # source: the current stream (a)
        with torch.cuda.stream(b):
            y = x.clone()
            seen.append(torch.stack([y.min(), y.max()]))
        a.wait_stream(b)                # the next round's writes wait for this round's read
    torch.cuda.synchronize()
    got = torch.stack(seen).cpu()
    want = torch.arange(1, 41, dtype=torch.float32).mul(4).unsqueeze(1).expand(40, 2)
    assert torch.equal(got, want), got[:6]
    fork(a, a)                          # same stream on both sides: nothing to order
    import copy
    assert isinstance(copy.deepcopy(fork), ops.StreamFork)


_FORK_CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from speedplusbaseline_amd import _lib as L, ops
dev = torch.device("cuda:0")
fork = ops.StreamFork()
a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
x = torch.zeros(4 << 20, device=dev)
mode = sys.argv[1]
if mode == "fallback":
    for i in range(10):
        with torch.cuda.stream(a):
            x.add_(1.0); x.add_(1.0)
            fork(b)
        with torch.cuda.stream(b):
            y = x.clone()
        a.wait_stream(b)
    torch.cuda.synchronize()
    assert float(y.min()) == 20.0 and float(y.max()) == 20.0, (float(y.min()), float(y.max()))
    print("SELFTEST", L.lib().spb_fork_selftest())
else:                                   # time-out: the storing kernel sits behind ~0.4 s of work, the gate gives up after 0.1 s
    big = torch.zeros(256 << 20, device=dev)
    with torch.cuda.stream(a):
        fork(b)                         # creates the native fork object (its creation synchronises the device) before the backlog exists
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        for _ in range(800):
            big.add_(1.0)
        fork(b)
    torch.cuda.synchronize()
    print("SELFTEST", L.lib().spb_fork_selftest())
    try:
        with torch.cuda.stream(a):
            fork(b)
        print("NO ERROR")
    except L.SpbError as e:
        print("ERROR", e)
    torch.cuda.synchronize()
    print("ALIVE", float(x.sum()))      # the device context survived (the old gate trapped: every later call failed)
"""


def _fork_child(mode, env):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    return subprocess.run([sys.executable, "-c", _FORK_CHILD % root, mode], env=e, capture_output=True, text=True, timeout=600)


def test_stream_fork_selftest_passes_on_this_runtime(device):
    """the property the device-word forks rest on (a kernel starts only after its stream's earlier kernels completed and released their
    results device-wide) is tested at start-up, csrc/elemwise.hip spb_fork_selftest: on this runtime it holds"""
    ops.StreamFork()(torch.cuda.Stream(device=device))          # creates a fork object -> runs the self-test
    assert L.lib().spb_fork_selftest() in (1, -1)               # (-1: events forced by SPB_EVENT_FORKS / a counter-collecting profiler)
    assert L.lib().spb_hip_runtime_version() > 0


def test_stream_fork_selftest_failure_falls_back_to_events(device):
    """SPB_FORK_SELFTEST_FAIL=1 (test rig): the library says so once on stderr, orders streams by events and still orders them"""
    r = _fork_child("fallback", {"SPB_FORK_SELFTEST_FAIL": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SELFTEST 0" in r.stdout and "self-test FAILED" in r.stderr and r.stderr.count("self-test FAILED") == 1


def test_stream_fork_gate_timeout_poisons_instead_of_trapping(device):
    """a gate whose storing launch does not run within SPB_FORK_TIMEOUT_S raises the fork's poison word: the next call on that fork
    fails with SPB_E_TIMEOUT (a Python exception) and the device context stays usable"""
    r = _fork_child("timeout", {"SPB_FORK_TIMEOUT_S": "0.1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SELFTEST 1" in r.stdout
    assert "ERROR" in r.stdout and "SPB_E_TIMEOUT" in r.stdout and "NO ERROR" not in r.stdout
    assert "ALIVE 0.0" in r.stdout
