"""Style augmentation surface of the reference (src/styleaug/styleAugmentor.py:12-68, ghiasi.py:6-136) on the MI355X.

`Ghiasi` keeps the reference's parameter layout (84 state-dict tensors, `layers.N.{conv,conv1,conv2,fc_*}.{weight,bias}`)
as a parameter container; its forward is HIP only (speedplusbaseline_amd/csrc/ghiasi.hip through the C-ABI in
include/spb_hip.h): implicit-GEMM convolutions on the matrix cores with reflection padding / stride / nearest upsampling
resolved while the input tile is staged, instance normalisation carried as per-(image, channel) sums and applied by the
consumer.  Inference only (the reference runs it under no_grad and detaches the result, styleAugmentor.py:56-68).
There is no CPU or eager-PyTorch fallback.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L

IN_EPS = 1e-5


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_UP2_BY_PHASE = os.environ.get("SPB_GCONV_UP2", "1") != "0"
_WIDE = os.environ.get("SPB_GCONV_WIDE", "1") != "0"     # residual-block convolutions through csrc/ghiasi_wide.hip


def _phase_weights(w):
    """[Cout, Cin, 3, 3] float32 -> [4 = py*2+px][Cout][4 = ty*2+tx][Cin] bf16: the 2x2 kernels that nearest x2 upsampling +
    ReflectionPad2d(1) + this 3x3 convolution amount to on the low-resolution input (csrc/ghiasi.hip gconv_up2_kernel): along an
    axis, output phase 0 sees taps (w0, w1 + w2) of rows (i-1, i), phase 1 sees (w0 + w1, w2) of rows (i, i+1)"""
    def split(t, dim, p):        # t indexed by a 3-tap axis `dim` -> the two summed taps of phase p
        a, b, c = t.unbind(dim)
        return (a, b + c) if p == 0 else (a + b, c)
    out = []
    for py in (0, 1):
        rows = split(w, 2, py)                         # each [Cout, Cin, 3 (kx)]
        for px in (0, 1):
            taps = []
            for r in rows:
                taps += list(split(r, 2, px))          # (ty, tx) order, each [Cout, Cin]
            out.append(torch.stack(taps, dim=1))       # [Cout, 4, Cin]
    return torch.stack(out).contiguous()           # (float32: the caller rounds to its 16-bit storage format)


class _Conv(nn.Module):
    """parameter container with nn.Conv2d's state-dict names"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        nn.init.uniform_(self.bias, -0.05, 0.05)


class _ConvInRelu(nn.Module):                      # ghiasi.py:6-23
    def __init__(self, cin, cout, k):
        super().__init__()
        self.n_params = 0
        self.conv = _Conv(cin, cout, k)


class _UpsampleConvInRelu(nn.Module):              # ghiasi.py:26-59
    def __init__(self, cin, cout, k):
        super().__init__()
        self.n_params = cout * 2
        self.conv = _Conv(cin, cout, k)
        self.fc_beta = nn.Linear(100, cout)
        self.fc_gamma = nn.Linear(100, cout)


class _ResidualBlock(nn.Module):                   # ghiasi.py:62-104
    def __init__(self, c):
        super().__init__()
        self.n_params = c * 4
        self.conv1 = _Conv(c, c, 3)
        self.fc_beta1 = nn.Linear(100, c)
        self.fc_gamma1 = nn.Linear(100, c)
        self.fc_beta2 = nn.Linear(100, c)
        self.fc_gamma2 = nn.Linear(100, c)
        self.conv2 = _Conv(c, c, 3)


class Ghiasi(nn.Module):
    """Ghiasi(): same constructor, attributes (`layers`, `n_params`) and state_dict as ghiasi.py:107-123; forward(x, styles)
    returns the sigmoid image like ghiasi.py:125-135, computed by the HIP kernels."""

    def __init__(self, precision="bf16"):
        """precision "bf16" (default): the matrix-core kernels; "fp16" (round 6): the same kernels in IEEE half (libspb_hip_f16.so: half
        storage, v_mfma_f32_16x16x32_f16, f32 accumulation and statistics) -- same speed, eight times finer rounding, the closest a
        matrix-core path gets to the reference, which runs the decoder outside autocast in float32 (trainer.py:68-69); "fp32": float32
        tensors and arithmetic through csrc/ghiasi_f32.hip, ~30x slower, for parity work"""
        super().__init__()
        if precision not in ("bf16", "fp16", "fp32"):
            raise ValueError("precision must be bf16, fp16 or fp32, got %r" % (precision,))
        self.precision = precision
        self._dt16 = torch.float16 if precision == "fp16" else torch.bfloat16
        self.layers = nn.ModuleList([
            _ConvInRelu(3, 32, 9), _ConvInRelu(32, 64, 3), _ConvInRelu(64, 128, 3),
            _ResidualBlock(128), _ResidualBlock(128), _ResidualBlock(128), _ResidualBlock(128), _ResidualBlock(128),
            _UpsampleConvInRelu(128, 64, 3), _UpsampleConvInRelu(64, 32, 3), _UpsampleConvInRelu(32, 3, 9)])
        self.n_params = sum(layer.n_params for layer in self.layers)
        self._packed = None
        self._ws = {}
        self.profile = None   # set to a list to collect (label, cuda event) marks of one forward (scratch/bench_ghiasi.py)

    def _lib(self):
        return L.lib_f16() if self.precision == "fp16" else L.lib()

    def _mark(self, label):
        if self.profile is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.profile.append((label, ev))

    # ---- compute copies: conv weights as bf16 [Cout][K*K][Cin], every fc stacked into one [N,100] matrix
    def _pack(self):
        dev = self.layers[0].conv.weight.device
        if dev.type != "cuda":
            raise RuntimeError("Ghiasi runs on the MI355X only; move the module to a cuda device")
        convs, fcw, fcb, off = {}, [], [], {}
        n = 0
        for i, layer in enumerate(self.layers):
            for name in ("conv", "conv1", "conv2"):
                if hasattr(layer, name):
                    c = getattr(layer, name)
                    if self.precision == "fp32":    # [Cout][K*K][Cin] float32 for every layer
                        convs[(i, name)] = (c.weight.detach().float().permute(0, 2, 3, 1).contiguous(), c.bias.detach().float().contiguous())
                    elif i == 0:
                        convs[(i, name)] = (c.weight.detach().float().contiguous(), c.bias.detach().float().contiguous())
                    else:
                        w = c.weight.detach().permute(0, 2, 3, 1).contiguous().to(self._dt16)
                        convs[(i, name)] = (w, c.bias.detach().float().contiguous())
                        if isinstance(layer, _UpsampleConvInRelu) and tuple(c.weight.shape[2:]) == (3, 3):
                            convs[(i, name, "up2")] = _phase_weights(c.weight.detach().float()).to(self._dt16)
                        if isinstance(layer, _ResidualBlock) and tuple(w.shape) == (128, 3, 3, 128):
                            wp = torch.empty_like(w)      # one 8 KB LDS image per reduction step (spb_gconv_wide_pack)
                            L.check(self._lib().spb_gconv_wide_pack(_p(w), _p(wp), _stream()), "spb_gconv_wide_pack")
                            convs[(i, name, "wide")] = wp
            for name in ("fc_beta", "fc_gamma", "fc_beta1", "fc_gamma1", "fc_beta2", "fc_gamma2"):
                if hasattr(layer, name):
                    fc = getattr(layer, name)
                    off[(i, name)] = n
                    n += fc.weight.shape[0]
                    fcw.append(fc.weight.detach().float()); fcb.append(fc.bias.detach().float())
        self._packed = dict(convs=convs, fcw=torch.cat(fcw).contiguous(), fcb=torch.cat(fcb).contiguous(), off=off, n=n)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        self._ws = {}
        return r

    def _buf(self, key, shape, dtype, dev):
        t = self._ws.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=dev)
            self._ws[key] = t
        return t

    @torch.no_grad()
    def forward(self, x, styles):
        lib = self._lib()
        if not (x.is_cuda and styles.is_cuda):
            raise RuntimeError("Ghiasi.forward needs cuda tensors (no CPU path)")
        if self._packed is None:
            self._pack()
        pk = self._packed
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError("image size must be a multiple of 32 (two stride-2 stages, 8x8 output tiles); got %dx%d" % (H, W))
        dev = x.device
        x = x.contiguous().float()
        styles = styles.contiguous().float()
        bf = self._dt16
        st = _stream()
        fc = self._buf("fc", (B, pk["n"]), torch.float32, dev)
        L.check(lib.spb_style_fc(_p(styles), _p(pk["fcw"]), _p(pk["fcb"]), _p(fc), B, pk["n"], st), "spb_style_fc")
        if self.precision == "fp32":
            return self._forward_f32(x, fc, pk, st)
        stats = self._buf("stats", (16, B, 128, 2), torch.float32, dev)
        stats.zero_()
        coef = self._buf("coef", (16, B, 128, 2), torch.float32, dev)
        si = [0]

        def norm_of(C_, hw, gamma_key=None, beta_key=None):
            """the instance norm (+ style affine) of the tensor whose sums were just accumulated in stats[si]: its consumer builds the
            per-(image, channel) scale / shift from the sums in its own prologue (spb_gconv_args_t.in_stats, spb_in_apply_stats,
            spb_final_sigmoid_stats); only spb_gconv_wide reads a coefficient table (coef_table below)"""
            k = si[0]
            si[0] += 1
            return dict(k=k, C=C_, hw=hw, gamma=fc[:, pk["off"][gamma_key]:] if gamma_key else None,
                        beta=fc[:, pk["off"][beta_key]:] if beta_key else None)

        def coef_table(n):
            L.check(lib.spb_in_coef(_p(stats[n["k"]]), _p(n["gamma"]), _p(n["beta"]), pk["n"], _p(coef[n["k"]]), B, n["C"], n["hw"],
                                    IN_EPS, st), "spb_in_coef")
            return coef[n["k"]]

        def gconv(key, X, Hin, Win, Cin, Cout, k, stride=1, up=1, cf=None, relu=0, ldc=None, name=None):
            w, bias = pk["convs"][key]
            Hout, Wout = Hin * up // stride, Win * up // stride
            ldc = ldc or Cout
            Y = self._buf(name or ("z%d%s" % key), (B, Hout, Wout, ldc), bf, dev)
            a = L.GconvArgs()
            a.X = _p(X); a.W = _p(w); a.bias = _p(bias); a.coef = None; a.Y = _p(Y); a.stats = _p(stats[si[0]])
            a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KH = k; a.stride = stride; a.upsample = up
            a.relu = relu; a.ldc = ldc
            if cf is not None:            # the producer's sums: coefficients built in the consumer's prologue
                a.in_stats = _p(stats[cf["k"]]); a.in_gamma = _p(cf["gamma"]); a.in_beta = _p(cf["beta"]); a.in_ld = pk["n"]
                a.in_inv_n = 1.0 / cf["hw"]; a.in_eps = IN_EPS
            wp = pk["convs"].get(key + ("up2",)) if (up == 2 and k == 3 and stride == 1 and _UP2_BY_PHASE) else None
            ww = pk["convs"].get(key + ("wide",)) if (_WIDE and up == 1 and stride == 1 and not (Hin % 8 or Win % 8)) else None
            if wp is not None:        # Upsample(2) + ReflectionPad(1) + 3x3 as four 2x2 phase convolutions on the low-res input
                a.W = _p(wp)
                L.check(lib.spb_gconv_up2(L.BF16, C.byref(a), st), "spb_gconv_up2")
            elif ww is not None:
                a.W = _p(ww)
                if cf is not None:
                    a.coef = _p(coef_table(cf))
                L.check(lib.spb_gconv_wide(L.BF16, C.byref(a), st), "spb_gconv_wide")
            else:
                L.check(lib.spb_gconv(L.BF16, C.byref(a), st), "spb_gconv")
            self._mark("gconv %dx%d %d->%d s%d u%d @%d" % (k, k, Cin, Cout, stride, up, Hout))
            return Y, Hout, Wout

        # ConvInRelu x3 (no style renormalisation, ghiasi.py:128-131)
        w0, b0 = pk["convs"][(0, "conv")]
        z0 = self._buf("z0", (B, H, W, 32), bf, dev)
        self._mark("start")
        L.check(lib.spb_conv9_rgb(_p(x), _p(w0), _p(b0), _p(z0), _p(stats[si[0]]), B, H, W, st), "spb_conv9_rgb")
        self._mark("conv9_rgb")
        c0 = norm_of(32, H * W)
        z1, H1, W1 = gconv((1, "conv"), z0, H, W, 32, 64, 3, stride=2, cf=c0, relu=1)
        c1 = norm_of(64, H1 * W1)
        z2, H2, W2 = gconv((2, "conv"), z1, H1, W1, 64, 128, 3, stride=2, cf=c1, relu=1)
        c2 = norm_of(128, H2 * W2)
        hw2 = H2 * W2
        r = self._buf("r0", (B, H2, W2, 128), bf, dev)
        L.check(lib.spb_in_apply_stats(_p(z2), _p(stats[c2["k"]]), _p(c2["gamma"]), _p(c2["beta"]), pk["n"], IN_EPS, None,
                                       _p(r), B, hw2, 128, 1, st), "spb_in_apply_stats")
        # ResidualBlock x5 (ghiasi.py:92-104)
        for i in range(3, 8):
            za, _, _ = gconv((i, "conv1"), r, H2, W2, 128, 128, 3, name="za")
            ca = norm_of(128, hw2, (i, "fc_gamma1"), (i, "fc_beta1"))
            zb, _, _ = gconv((i, "conv2"), za, H2, W2, 128, 128, 3, cf=ca, relu=1, name="zb")
            cb = norm_of(128, hw2, (i, "fc_gamma2"), (i, "fc_beta2"))
            rn = self._buf("r1" if r is self._ws.get("r0") else "r0", (B, H2, W2, 128), bf, dev)
            L.check(lib.spb_in_apply_stats(_p(zb), _p(stats[cb["k"]]), _p(cb["gamma"]), _p(cb["beta"]), pk["n"], IN_EPS, _p(r), _p(rn),
                                           B, hw2, 128, 0, st), "spb_in_apply_stats")
            r = rn
        # UpsampleConvInRelu x3 (ghiasi.py:46-59)
        z8, H8, W8 = gconv((8, "conv"), r, H2, W2, 128, 64, 3, up=2)
        c8 = norm_of(64, H8 * W8, (8, "fc_gamma"), (8, "fc_beta"))
        z9, H9, W9 = gconv((9, "conv"), z8, H8, W8, 64, 32, 3, up=2, cf=c8, relu=1)
        c9 = norm_of(32, H9 * W9, (9, "fc_gamma"), (9, "fc_beta"))
        z10, _, _ = gconv((10, "conv"), z9, H9, W9, 32, 3, 9, cf=c9, relu=1, ldc=4)
        c10 = norm_of(3, H9 * W9, (10, "fc_gamma"), (10, "fc_beta"))
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        L.check(lib.spb_final_sigmoid_stats(_p(z10), _p(stats[c10["k"]]), _p(c10["gamma"]), _p(c10["beta"]), pk["n"], IN_EPS, _p(out), B,
                                            H * W, 4, st), "spb_final_sigmoid_stats")
        self._mark("final")
        return out

    def _forward_f32(self, x, fc, pk, st):
        """ghiasi.py:125-135 in float32: the layer sequence of forward() through the direct-convolution kernel (spb_gconv with SPB_F32), the
        coefficient table of every instance norm from spb_in_coef, float32 residual stream"""
        lib = L.lib()
        B, _, H, W = x.shape
        dev = x.device
        f32 = torch.float32
        stats = self._buf("stats32", (16, B, 128, 2), f32, dev)
        stats.zero_()
        coef = self._buf("coef32", (16, B, 128, 2), f32, dev)
        k = [0]

        def conv(key, X, Hin, Win, Cin, Cout, ks, stride=1, up=1, cf=None, relu=0, ldc=None, name=None):
            w, bias = pk["convs"][key]
            Hout, Wout = Hin * up // stride, Win * up // stride
            ldc = ldc or Cout
            Y = self._buf(name or ("y32_%d%s" % key), (B, Hout, Wout, ldc), f32, dev)
            a = L.GconvArgs()
            a.X = _p(X); a.W = _p(w); a.bias = _p(bias); a.coef = _p(cf) if cf is not None else None; a.Y = _p(Y); a.stats = _p(stats[k[0]])
            a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KH = ks; a.stride = stride; a.upsample = up
            a.relu = relu; a.ldc = ldc
            L.check(lib.spb_gconv(L.F32, C.byref(a), st), "spb_gconv (fp32)")
            return Y, Hout, Wout

        def norm(C_, hw, gk=None, bk=None):
            """coefficient table of the tensor whose sums were just accumulated"""
            i = k[0]; k[0] += 1
            ga = fc[:, pk["off"][gk]:] if gk else None
            be = fc[:, pk["off"][bk]:] if bk else None
            L.check(lib.spb_in_coef(_p(stats[i]), _p(ga), _p(be), pk["n"], _p(coef[i]), B, C_, hw, IN_EPS, st), "spb_in_coef")
            return coef[i]

        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
        z0, _, _ = conv((0, "conv"), x_nhwc, H, W, 3, 32, 9)
        c0 = norm(32, H * W)
        z1, H1, W1 = conv((1, "conv"), z0, H, W, 32, 64, 3, stride=2, cf=c0, relu=1)
        c1 = norm(64, H1 * W1)
        z2, H2, W2 = conv((2, "conv"), z1, H1, W1, 64, 128, 3, stride=2, cf=c1, relu=1)
        c2 = norm(128, H2 * W2)
        hw2 = H2 * W2
        r = self._buf("r32_0", (B, H2, W2, 128), f32, dev)
        L.check(lib.spb_in_apply_f32(_p(z2), _p(c2), None, _p(r), B, hw2, 128, 1, st), "spb_in_apply_f32")
        for i in range(3, 8):
            za, _, _ = conv((i, "conv1"), r, H2, W2, 128, 128, 3, name="za32")
            ca = norm(128, hw2, (i, "fc_gamma1"), (i, "fc_beta1"))
            zb, _, _ = conv((i, "conv2"), za, H2, W2, 128, 128, 3, cf=ca, relu=1, name="zb32")
            cb = norm(128, hw2, (i, "fc_gamma2"), (i, "fc_beta2"))
            rn = self._buf("r32_1" if r is self._ws.get("r32_0") else "r32_0", (B, H2, W2, 128), f32, dev)
            L.check(lib.spb_in_apply_f32(_p(zb), _p(cb), _p(r), _p(rn), B, hw2, 128, 0, st), "spb_in_apply_f32")
            r = rn
        z8, H8, W8 = conv((8, "conv"), r, H2, W2, 128, 64, 3, up=2)
        c8 = norm(64, H8 * W8, (8, "fc_gamma"), (8, "fc_beta"))
        z9, H9, W9 = conv((9, "conv"), z8, H8, W8, 64, 32, 3, up=2, cf=c8, relu=1)
        c9 = norm(32, H9 * W9, (9, "fc_gamma"), (9, "fc_beta"))
        z10, _, _ = conv((10, "conv"), z9, H9, W9, 32, 3, 9, cf=c9, relu=1, ldc=4)
        c10 = norm(3, H9 * W9, (10, "fc_gamma"), (10, "fc_beta"))
        out = torch.empty(B, 3, H, W, dtype=f32, device=dev)
        L.check(lib.spb_final_sigmoid_f32(_p(z10), _p(c10), _p(out), B, H * W, 4, st), "spb_final_sigmoid_f32")
        return out


class StyleAugmentor(nn.Module):
    """StyleAugmentor(alpha, device) as styleAugmentor.py:12-68.  The reference's checkpoints (transformer weights,
    embedding mean/covariance, SPEED+ base embedding) are data files of the reference repository: pass their directory as
    `checkpoint_dir` (default: <this package>/styleaug/checkpoints, mirroring styleAugmentor.py:23-31).  For tests and
    synthetic benchmarks `StyleAugmentor.synthetic(alpha, device, seed)` builds one with random weights and a synthetic
    SPD covariance."""

    def __init__(self, alpha, device, checkpoint_dir=None, _parts=None, precision="bf16"):
        super().__init__()
        self.alpha = alpha
        self.device = device
        self.ghiasi = Ghiasi(precision)     # "fp32": the reference's own precision for this module (trainer.py:68-69), ~30x slower
        if _parts is None:
            d = checkpoint_dir or os.path.join(os.path.dirname(__file__), "checkpoints")
            need = [os.path.join(d, f) for f in ("checkpoint_transformer.pth", "checkpoint_embeddings.pth", "embedding_mean_speedplus.npy")]
            missing = [f for f in need if not os.path.exists(f)]
            if missing:
                raise FileNotFoundError("style augmentation checkpoints not found (they ship with the reference repository, "
                                        "src/styleaug/checkpoints): %s" % ", ".join(missing))
            ck = torch.load(need[0], map_location="cpu")
            emb = torch.load(need[1], map_location="cpu")
            self.ghiasi.load_state_dict(ck["state_dict_ghiasi"], strict=False)
            base = torch.from_numpy(np.load(need[2])).float()
            mean, cov = emb["pbn_embedding_mean"], emb["pbn_embedding_covariance"]
        else:
            sd, base, mean, cov = _parts
            self.ghiasi.load_state_dict(sd, strict=True)
        self.ghiasi.to(device)
        self.imagenet_embedding = base.float().to(device)       # SPEED+ embedding, despite the name (styleAugmentor.py:33)
        self.mean = mean.float().to(device)                       # 1 x 100
        self.cov = cov
        u, s, _ = np.linalg.svd(np.asarray(cov, dtype=np.float64))
        self.A = torch.tensor(np.matmul(u, np.diag(s ** 0.5))).float().to(device)   # 100 x 100

    @classmethod
    def synthetic(cls, alpha, device, state_dict, seed=0, precision="bf16"):
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(100, 100, generator=g, dtype=torch.float64)
        cov = (q @ q.t() / 100.0 + 0.05 * torch.eye(100, dtype=torch.float64)).numpy()
        mean = torch.randn(1, 100, generator=g) * 0.3
        base = torch.randn(100, generator=g) * 0.3
        return cls(alpha, device, _parts=(state_dict, base, mean, cov), precision=precision)

    def sample_embedding(self, n):
        embedding = torch.randn(n, 100).to(self.device)
        return torch.mm(embedding, self.A.transpose(1, 0)) + self.mean

    def forward(self, x):
        base = self.imagenet_embedding
        with torch.no_grad():
            embedding = self.sample_embedding(x.size(0))
            embedding = self.alpha * embedding + (1 - self.alpha) * base
            restyled = self.ghiasi(x, embedding)
        return restyled.detach()
