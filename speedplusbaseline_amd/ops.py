"""Thin torch-tensor front end of the per-kernel C-ABI entry points (used by the parity tests and by callers that
want a single fused op).  Tensors are device memory only; all arithmetic happens in libspb_hip.so.
"""
import ctypes as C

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.BF16}   # float16: the same 16-bit code, served by the IEEE-half twin library


def lib_of(t):
    """the library that understands tensor t's 16-bit format: libspb_hip_f16.so for float16, libspb_hip.so otherwise"""
    return L.lib_f16() if (t is not None and t.dtype == torch.float16) else L.lib()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("speedplusbaseline_amd ops run on the GPU only (got a %s tensor)" % t.device)
        if t is not None and not t.is_contiguous():
            raise RuntimeError("speedplusbaseline_amd ops need contiguous tensors")


def _need_rows(*ts):
    """2-D operands that may be column slabs of a wider row-major matrix (unit column stride, any row stride)"""
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("speedplusbaseline_amd ops run on the GPU only (got a %s tensor)" % t.device)
        if t.dim() != 2 or t.stride(1) != 1 or (t.stride(0) & 7) or (t.storage_offset() & 7):
            raise RuntimeError("speedplusbaseline_amd GEMM operands are row-major 2-D tensors or 8-aligned column slabs of one")


def dtype_code(t, twin=False):
    """16-bit / 32-bit code of tensor t for the C-ABI.  float16 shares bfloat16's code and is only meaningful in the IEEE-half
    twin library (lib_of): entry points that exist in libspb_hip.so alone (the KRN kernels) pass twin=False and refuse it
    instead of reading half bits as bfloat16."""
    if t.dtype == torch.float16 and not twin:
        raise RuntimeError("this kernel has no float16 instance (bfloat16 / float32 only): float16 is built for the SPN path "
                           "(libspb_hip_f16.so); KRN / RevGrad use --precision bf16")
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError("activation dtype must be float32, bfloat16 or float16, got %s" % t.dtype)


def bnref(C_, sums=None, gamma=None, beta=None, bsums=None, n=1, R=1, act=L.ACT_NONE, slope=0.0, eps=1e-5, moments=0):
    """spb_bnref_t from f32 tensors. gamma=None means identity (already normalised tensor)."""
    _need_cuda(sums, gamma, beta, bsums)
    r = L.BNRef()
    r.sums = _ptr(sums); r.gamma = _ptr(gamma); r.beta = _ptr(beta); r.bsums = _ptr(bsums)
    r.inv_n = 1.0 / float(n); r.eps = eps; r.slope = slope; r.C = C_; r.R = R; r.act = act; r.moments = moments
    r._keep = (sums, gamma, beta, bsums)
    return r


def pwconv_gemm(A, Bw, Y, pro, pro_mode, epi_mode, A2=None, res=None, Zout=None, bias=None, osums=None, epi=None,
                out_act=L.ACT_NONE, oR=1, out_scale=1.0, pro2=None, Ymat=None):
    if Y is None:               # statistics only (epi_mode 1): the batch sums of a product that is never stored
        _need_rows(A)
    else:
        _need_rows(A, Y)
    _need_cuda(Bw, A2, res, Zout, bias, osums)
    g = L.GemmArgs()
    g.A = _ptr(A); g.A2 = _ptr(A2); g.Bw = _ptr(Bw); g.Y = _ptr(Y); g.res = _ptr(res); g.Zout = _ptr(Zout)
    g.bias = _ptr(bias); g.osums = _ptr(osums); g.pro = pro
    g.epi = epi if epi is not None else bnref(Bw.shape[0])
    g.M, g.K = A.shape[0], A.shape[1]; g.N = Bw.shape[0]
    g.lda = A.stride(0) if A.stride(0) != A.shape[1] else 0     # column slabs of wider matrices (grouped convolutions)
    g.ldc = 0 if Y is None else (Y.stride(0) if Y.stride(0) != Y.shape[1] else 0)
    g.pro_mode = pro_mode; g.epi_mode = epi_mode; g.out_act = out_act; g.oR = oR; g.out_scale = out_scale
    if pro_mode == 3:       # residual join: a = bn(A) + bn2(A2), written to Ymat by the launch
        _need_cuda(Ymat)
        g.pro2 = pro2 if pro2 is not None else bnref(A.shape[1]); g.Ymat = _ptr(Ymat)
    elif Ymat is not None:  # statistics-only pass (Y is None): also write the operand round16(act(bn(A)))
        _need_cuda(Ymat)
        g.Ymat = _ptr(Ymat)
    L.check(lib_of(A).spb_pwconv_gemm(dtype_code(A, True), C.byref(g), _stream()), "spb_pwconv_gemm")


def pwconv_wgrad(G, X, dW, pro_dz, pro_a, Zn=None):
    _need_rows(G, X)
    _need_cuda(dW, Zn)
    w = L.WgradArgs()
    w.G = _ptr(G); w.Zn = _ptr(Zn); w.X = _ptr(X); w.dW = _ptr(dW); w.pro_dz = pro_dz; w.pro_a = pro_a
    w.M = G.shape[0]; w.N = G.shape[1]; w.K = X.shape[1]
    w.ldg = G.stride(0) if G.stride(0) != G.shape[1] else 0
    w.ldx = X.stride(0) if X.stride(0) != X.shape[1] else 0
    L.check(lib_of(G).spb_pwconv_wgrad(dtype_code(G, True), C.byref(w), _stream()), "spb_pwconv_wgrad")


def pwconv_bwd_fused(G, Zn, Wt, X, Zout, Y, dW, osums, pro_dz, pro_a, epi, res=None, oR=1):
    """Fused input-gradient + weight-gradient of a pointwise conv.  Returns False when the library has no fused instance
    for this (dtype, N, K) -- the caller then uses pwconv_gemm + pwconv_wgrad."""
    _need_cuda(G, Zn, Wt, X, Zout, Y, dW, osums, res)
    a = L.PwBwdArgs()
    a.G = _ptr(G); a.Zn = _ptr(Zn); a.Wt = _ptr(Wt); a.X = _ptr(X); a.Zout = _ptr(Zout); a.res = _ptr(res); a.Y = _ptr(Y)
    a.dW = _ptr(dW); a.osums = _ptr(osums); a.pro_dz = pro_dz; a.pro_a = pro_a; a.epi = epi
    a.M = G.shape[0]; a.N = G.shape[1]; a.K = X.shape[1]; a.oR = oR
    code = L.lib().spb_pwconv_bwd_fused(dtype_code(G), C.byref(a), _stream())
    if code == -4:  # SPB_E_UNSUPPORTED
        return False
    L.check(code, "spb_pwconv_bwd_fused")
    return True


def _dwargs(X, Wd, B, H, W_, C_, stride, **kw):
    d = L.DwArgs()
    d.X = _ptr(X); d.Wd = _ptr(Wd); d.B = B; d.H = H; d.W = W_; d.C = C_; d.stride = stride
    for k, v in kw.items():
        if isinstance(v, torch.Tensor) or v is None:
            setattr(d, k, _ptr(v))
        else:
            setattr(d, k, v)
    ident = bnref(C_)
    for k in ("pro", "pro_in", "epi"):
        if k not in kw or kw[k] is None:
            setattr(d, k, ident)
    if "oR" not in kw:
        d.oR = 1
    d.xe = bnref(8)
    return d


def _set_expand(d, expand):
    """spb_dw_args_t::Xe -- the depthwise layer's input is the (never stored) output of a 1x1 expand convolution:
    expand = (Xe [B,H,W,Ce], We [C,Ce] compute dtype, xe = BN/activation turning Xe into that convolution's operand or None)"""
    Xe, We, xe = expand
    _need_cuda(Xe, We)
    d.Xe = _ptr(Xe); d.We = _ptr(We); d.Ce = Xe.shape[-1]
    d.xe = xe if xe is not None else bnref(Xe.shape[-1])
    d._keep_expand = (Xe, We, xe)


def dwconv_fwd(X, Wd, Y, pro, stride, osums=None, oR=1, expand=None):
    """expand: see _set_expand; X is then None and `pro` names the BatchNorm / activation on the recomputed tensor"""
    src = X if expand is None else expand[0]
    B, H, W_ = src.shape[:3]
    C_ = Y.shape[3]
    _need_cuda(X, Wd, Y, osums)
    d = _dwargs(X, Wd, B, H, W_, C_, stride, Y=Y, pro=pro, osums=osums, oR=oR, epi_mode=1 if osums is not None else 0)
    if expand is not None:
        _set_expand(d, expand)
    L.check(L.lib().spb_dwconv_fwd(dtype_code(Y), C.byref(d), _stream()), "spb_dwconv_fwd")


def dwconv_dgrad(G, Z, Wd, Y, pro, stride, in_hw, epi=None, Zout=None, res=None, osums=None, oR=1, dW=None, Xin=None,
                 pro_in=None, entry_flag=None, entry_val=0, expand=None):
    """Input gradient of the depthwise conv.  With `dW` the weight gradient is accumulated in the same pass; `Xin` and
    `pro_in` (the conv's input tensor and its BN/activation) are then required unless `epi`/`Zout` already name them.
    `entry_flag` (a device int32 word): the launch's first thread stores `entry_val` there before anything else
    (spb_dw_args_t::entry_flag: a stream fork without an event)."""
    B, C_ = G.shape[0], G.shape[3]
    _need_cuda(G, Z, Wd, Y, Zout, res, osums, dW, Xin)
    kw = dict(X2=Z, Y=Y, pro=pro, epi_mode=2 if epi is not None else 0, oR=oR)
    if epi is not None:
        kw.update(epi=epi, Zout=Zout, res=res, osums=osums)
    if dW is not None:
        kw.update(dW=dW)
        if epi is None:
            kw.update(epi=pro_in, Zout=Xin)
    d = _dwargs(G, Wd, B, in_hw[0], in_hw[1], C_, stride, **kw)
    if entry_flag is not None:
        _need_cuda(entry_flag)
        d.entry_flag = _ptr(entry_flag); d.entry_val = int(entry_val)
    if expand is not None:      # the conv input (Zout / Xin) is recomputed: `epi` / `pro_in` name the BatchNorm on the recomputed tensor
        _set_expand(d, expand)
    L.check(L.lib().spb_dwconv_dgrad(dtype_code(G), C.byref(d), _stream()), "spb_dwconv_dgrad")


def dwconv_wgrad(G, Z, Xin, Wd, dW, pro, pro_in, stride, expand=None):
    src = Xin if expand is None else expand[0]
    B, H, W_ = src.shape[:3]
    C_ = G.shape[3]
    _need_cuda(G, Z, Xin, Wd, dW)
    d = _dwargs(G, Wd, B, H, W_, C_, stride, X2=Z, Xin=Xin, dW=dW, pro=pro, pro_in=pro_in)
    if expand is not None:
        _set_expand(d, expand)
    L.check(L.lib().spb_dwconv_wgrad(dtype_code(G), C.byref(d), _stream()), "spb_dwconv_wgrad")


def stem_fwd(x_nchw, w, y_nhwc, osums=None, oR=1):
    _need_cuda(x_nchw, w, y_nhwc, osums)
    B, _, H, W_ = x_nchw.shape
    L.check(L.lib().spb_stem_fwd(dtype_code(y_nhwc), _ptr(x_nchw), _ptr(w), _ptr(y_nhwc), _ptr(osums), oR, B, H, W_,
                                 _stream()), "spb_stem_fwd")


def stem_wgrad(x_nchw, G, Z, pro_dz, dW):
    _need_cuda(x_nchw, G, Z, dW)
    B, _, H, W_ = x_nchw.shape
    L.check(L.lib().spb_stem_wgrad(dtype_code(G), _ptr(x_nchw), _ptr(G), _ptr(Z), C.byref(pro_dz), _ptr(dW), B, H, W_,
                                   _stream()), "spb_stem_wgrad")


def bn_apply(Z, Y, bn, res=None, bn_res=None, ldc=None, coff=0, reorg=0):
    _need_cuda(Z, Y, res)
    B, H, W_, C_ = Z.shape
    a = L.BnApplyArgs()
    a.Z = _ptr(Z); a.res = _ptr(res); a.Y = _ptr(Y); a.bn = bn
    a.bn_res = bn_res if bn_res is not None else bnref(C_)
    a.B, a.H, a.W, a.C = B, H, W_, C_
    a.ldc = ldc if ldc is not None else C_
    a.coff = coff; a.reorg = reorg
    L.check(L.lib().spb_bn_apply(dtype_code(Z), C.byref(a), _stream()), "spb_bn_apply")


def bn_bwd_prep(dY, Z, G, osums, bn, ldc=None, coff=0, reorg=0, oR=1):
    _need_cuda(dY, Z, G, osums)
    B, H, W_, C_ = Z.shape
    a = L.BnBwdArgs()
    a.dY = _ptr(dY); a.Z = _ptr(Z); a.G = _ptr(G); a.osums = _ptr(osums); a.bn = bn
    a.B, a.H, a.W, a.C = B, H, W_, C_
    a.ldc = ldc if ldc is not None else C_
    a.coff = coff; a.reorg = reorg; a.oR = oR
    L.check(L.lib().spb_bn_bwd_prep(dtype_code(Z), C.byref(a), _stream()), "spb_bn_bwd_prep")


def head_fwd(Z, Wp, bias, pro, J, HW, C_, target=None, S=256, partial=None):
    """Z [B, HW*C]; Wp [Jp, HW*C] (compute dtype). Returns pred [B,J], scalars [3], dout [B,J].  partial: the reduction workspace
    (S*B*Jp floats, zero before its first use; every call leaves it zero again) -- a fresh one is allocated when None."""
    _need_cuda(Z, Wp, bias, target)
    B = Z.shape[0]
    Jp = Wp.shape[0]
    dev = Z.device
    if partial is None:
        partial = torch.zeros(S * B * Jp, dtype=torch.float32, device=dev)   # ends in the reduction's ticket word: zero before the first call
    pred = torch.empty(B, J, dtype=torch.float32, device=dev)
    dout = torch.zeros(B, J, dtype=torch.float32, device=dev)
    scalars = torch.zeros(3, dtype=torch.float32, device=dev)
    h = L.HeadArgs()
    h.Z = _ptr(Z); h.Wp = _ptr(Wp); h.bias = _ptr(bias); h.target = _ptr(target); h.partial = _ptr(partial)
    h.pred = _ptr(pred); h.dout = _ptr(dout); h.scalars = _ptr(scalars); h.pro = pro
    h.B, h.J, h.Jp, h.HW, h.C, h.S = B, J, Jp, HW, C_, S
    L.check(L.lib().spb_head_fwd(dtype_code(Z), C.byref(h), _stream()), "spb_head_fwd")
    return pred, scalars, dout


def head_bwd(Z, Wp, dout, G, osums, dW, dbias, pro, J, HW, C_, gscale=1.0, oR=1):
    _need_cuda(Z, Wp, dout, G, osums, dW, dbias)
    h = L.HeadBwdArgs()
    h.Z = _ptr(Z); h.Wp = _ptr(Wp); h.dout = _ptr(dout); h.G = _ptr(G); h.osums = _ptr(osums); h.dW = _ptr(dW)
    h.dbias = _ptr(dbias); h.pro = pro; h.gscale = gscale
    h.B, h.J, h.Jp, h.HW, h.C, h.oR = Z.shape[0], J, Wp.shape[0], HW, C_, oR
    L.check(L.lib().spb_head_bwd(dtype_code(Z), C.byref(h), _stream()), "spb_head_bwd")


def grad_sqnorm(grads, out):
    _need_cuda(grads, out)
    L.check(L.lib().spb_grad_sqnorm(_ptr(grads), grads.numel(), _ptr(out), _stream()), "spb_grad_sqnorm")


SQ_PARTS = 256


def grad_sqnorm_partials(grads, out):
    """sum of squares of the arena as SQ_PARTS partial sums (no arrival counter); optim_step(sq_partials=out) adds them up"""
    _need_cuda(grads, out)
    assert out.numel() >= SQ_PARTS and out.dtype == torch.float32
    L.check(L.lib().spb_grad_sqnorm_partials(_ptr(grads), grads.numel(), _ptr(out), _stream()), "spb_grad_sqnorm_partials")


def arena_zero(arena):
    """optimizer.zero_grad() on a flat f32 arena (no ATen kernel on the hot path)"""
    _need_cuda(arena)
    L.check(L.lib().spb_arena_zero(_ptr(arena), arena.numel(), _stream()), "spb_arena_zero")


def arena_add(dst, src):
    _need_cuda(dst, src)
    L.check(L.lib().spb_arena_add(_ptr(dst), _ptr(src), dst.numel(), _stream()), "spb_arena_add")


OPT_KIND = {"sgd": 0, "rmsprop": 1, "adam": 2, "adamw": 3}


def optim_step(kind, params, grads, m=None, v=None, sqnorm=None, gmul=None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
               weight_decay=0.0, max_norm=0.0, clip_value=0.0, step=1, first_step=False, hyper=None, shadow=None, max_blocks=0,
               sq_partials=None, skip=None):
    a = _optim_args(kind, params, grads, m, v, sqnorm, gmul, lr, beta1, beta2, eps, weight_decay, max_norm, clip_value, step, first_step,
                    hyper, shadow, max_blocks)
    if sq_partials is not None:
        _need_cuda(sq_partials)
        a.sq_partials = _ptr(sq_partials); a.n_sq_partials = SQ_PARTS
    if skip is not None:
        _need_cuda(skip)
        a.skip = _ptr(skip)
    L.check(lib_of(shadow).spb_optim_step(C.byref(a), _stream()), "spb_optim_step")   # a float16 shadow: the half twin writes it


def _optim_args(kind, params, grads, m, v, sqnorm, gmul, lr, beta1, beta2, eps, weight_decay, max_norm, clip_value, step, first_step,
                hyper, shadow, max_blocks):
    _need_cuda(params, grads, m, v, sqnorm, gmul, hyper, shadow)
    a = L.OptimArgs()
    a.params = _ptr(params); a.grads = _ptr(grads); a.m = _ptr(m); a.v = _ptr(v); a.sqnorm = _ptr(sqnorm)
    a.gmul = _ptr(gmul); a.hyper = _ptr(hyper); a.n = params.numel(); a.kind = OPT_KIND[kind]
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay
    a.max_norm = max_norm; a.clip_value = clip_value; a.shadow_bf16 = _ptr(shadow); a.max_blocks = max_blocks
    a.bias_c1 = 1.0 - beta1 ** step; a.bias_c2 = 1.0 - beta2 ** step; a.first_step = 1 if first_step else 0
    return a


def fc_wgrad_update(gT, xT, M, kind, params, grads=None, m=None, v=None, gmul=None, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                    weight_decay=0.0, clip_value=0.0, step=1, first_step=False, shadow=None):
    """weight gradient of a fully connected layer (params: its [N][K] weight) fused with its optimizer step (spb_fc_wgrad_update)"""
    _need_cuda(gT, xT)
    N, K = params.shape
    a = _optim_args(kind, params, grads, m, v, None, gmul, lr, beta1, beta2, eps, weight_decay, 0.0, clip_value, step, first_step, None,
                    shadow, 0)
    L.check(lib_of(gT).spb_fc_wgrad_update(_ptr(gT), _ptr(xT), M, N, K, C.byref(a), _stream()), "spb_fc_wgrad_update")


def debug_trread(inp, out):
    _need_cuda(inp, out)
    L.check(L.lib().spb_debug_trread(_ptr(inp), _ptr(out), _stream()), "spb_debug_trread")


class StreamFork:
    """`fork(to)`: everything enqueued on stream `to` afterwards runs behind everything the current stream holds now -- what
    `to.wait_stream(torch.cuda.current_stream())` does, without the event record that costs the CURRENT stream 6-9 us on this device
    (spb_fork_streams in include/spb_hip.h: a one-wave store kernel here, a one-wave gate kernel there; events inside a stream capture, with
    SPB_EVENT_FORKS=1 and under rocprofv3 counter collection).  One native object per source stream (serial numbers must be stored in
    order); joins -- the current stream waiting for `to` -- stay `wait_stream`: the waiting stream has to stall there anyway."""

    def __init__(self):
        self._h = {}

    def __call__(self, to, src=None):
        src = torch.cuda.current_stream(to.device) if src is None else src
        if src.cuda_stream == to.cuda_stream:
            return
        h = self._h.get(src.cuda_stream)
        if h is None:
            h = C.c_void_p()
            with torch.cuda.device(to.device):
                L.check(L.lib().spb_fork_create(C.byref(h)), "spb_fork_create")
            self._h[src.cuda_stream] = h
        L.check(L.lib().spb_fork_streams(h, C.c_void_p(src.cuda_stream), C.c_void_p(to.cuda_stream)), "spb_fork_streams")

    def __del__(self):
        try:
            for h in self._h.values():
                L.lib().spb_fork_destroy(h)
        except Exception:
            pass

    def __deepcopy__(self, memo):     # native handles are not copied: a copied module forks through objects of its own
        return StreamFork()

    def __reduce__(self):
        return (StreamFork, ())
