"""Spacecraft Pose Network on the MI355X (reference src/nets/spn.py:37-143; training step src/core/trainer.py:114-199).

Same constructor, attributes, state_dict keys (conv1..conv5, fc6..fc11 `.weight` / `.bias`) and return values as the
reference class; forward, loss and backward are HIP launches through the C-ABI (include/spb_hip.h):
every convolution / fully connected layer is a matrix-core GEMM (spb_pwconv_gemm with the bias + ReLU epilogue,
spb_pwconv_wgrad) on im2col'd NHWC operands, with the pooling / LRN / dropout / soft-target cross-entropy kernels of
csrc/spn.hip in between.  Grouped convolutions run as block-diagonal full convolutions.  First version of this row:
correct and matrix-core based, not tuned (weight re-layouts are torch copies per step).  No CPU / eager fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops

LRN_ALPHA, LRN_BETA, LRN_K = 2e-5, 0.75, 1.0           # nn.LocalResponseNorm(2, alpha=2e-5, beta=0.75, k=1.0), spn.py:62,67

# name, Cout, Cin (full), groups, K, stride, pad
_CONVS = (("conv1", 96, 3, 1, 11, 4, 0), ("conv2", 256, 96, 2, 5, 1, 2), ("conv3", 384, 256, 1, 3, 1, 1),
          ("conv4", 384, 384, 2, 3, 1, 1), ("conv5", 256, 384, 2, 3, 1, 1))


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def softmax_cross_entropy_with_logits(logits, target, reduction="mean"):
    """spn.py:37-48 on the GPU (HIP kernel): -sum(target * log_softmax(logits), 1) reduced over the batch"""
    if not logits.is_cuda:
        raise RuntimeError("softmax_cross_entropy_with_logits needs cuda tensors (no CPU path)")
    if reduction not in ("mean", "sum"):
        raise NotImplementedError("reduction='none' is not provided by the HIP kernel")
    B, Cn = logits.shape
    lg = logits.detach().contiguous()
    lg = lg if lg.dtype in (torch.float32, torch.bfloat16) else lg.float()
    out = torch.zeros(3, dtype=torch.float32, device=logits.device)
    L.check(L.lib().spb_softce(ops.dtype_code(lg), _p(lg), _p(target.detach().float().contiguous()), None, _p(out), 1, B, Cn, 1.0,
                               _st()), "spb_softce")
    return out[1] * (B if reduction == "sum" else 1)


class _Layer(nn.Module):
    """parameter container with nn.Conv2d / nn.Linear state-dict names"""

    def __init__(self, wshape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        self.bias = nn.Parameter(torch.empty(wshape[0]))
        fan_in = 1
        for d in wshape[1:]:
            fan_in *= d
        bound = 1.0 / fan_in ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class SpacecraftPoseNet(nn.Module):
    def __init__(self, num_classes, keep_prob=0.5, pretrain=True, precision="bf16"):
        super().__init__()
        self.num_classes = num_classes
        self.regress_size = num_classes
        self.keep_prob = keep_prob
        for name, cout, cin, g, k, _, _ in _CONVS:
            setattr(self, name, _Layer((cout, cin // g, k, k)))
        for name, o, i in (("fc6", 4096, 9216), ("fc7", 4096, 4096), ("fc8", num_classes, 4096), ("fc9", 4096, 9216),
                           ("fc10", 4096, 4096), ("fc11", num_classes, 4096)):
            setattr(self, name, _Layer((o, i)))
        self.precision = precision
        self._version = 0          # bumped whenever parameters change: compute copies are rebuilt lazily
        self._copies = None
        self._ws = {}
        self._saved = None
        self.dropout_seed = 2021
        self._step = 0
        if pretrain:
            self.load_weights('checkpoints/pretrained/bvlc_alexnet.npy')

    # ---- spn.py:104-123
    def load_weights(self, weight_path):
        import numpy as np
        weights_dict = np.load(weight_path, allow_pickle=True, encoding='bytes').item()
        with torch.no_grad():
            for name in weights_dict:
                if name in ['conv1', 'conv2', 'conv3', 'conv4', 'conv5']:
                    for data in weights_dict[name]:
                        if len(data.shape) == 4:
                            data = np.transpose(data, (3, 2, 0, 1))   # [H, W, Cin, Cout] -> [Cout, Cin, H, W]
                            getattr(self, name).weight.copy_(torch.from_numpy(data).float())
                        else:
                            getattr(self, name).bias.copy_(torch.from_numpy(data).float())
        self.invalidate()

    def invalidate(self):
        self._version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate(); self._ws = {}; self._copies = None
        return r

    # ---- compute-dtype copies: conv weights as [Cout][Kpad] in (ky,kx,c_full) order (block diagonal for groups),
    #      fc6/fc9 columns permuted from the reference's NCHW flatten to NHWC, plus the transposes the input gradients need
    def _dt(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    def _build_copies(self, need_t):
        if self._copies is not None and self._copies["v"] == self._version and (self._copies["t"] or not need_t):
            return self._copies
        dt = self._dt()
        cp = {"v": self._version, "t": need_t}
        for name, cout, cin, g, k, _, _ in _CONVS:
            w = getattr(self, name).weight.detach()
            kk = k * k * cin
            kpad = (kk + 7) // 8 * 8
            full = torch.zeros(cout, k, k, cin, dtype=torch.float32, device=w.device)
            cog, cig = cout // g, cin // g
            for gi in range(g):
                full[gi * cog:(gi + 1) * cog, :, :, gi * cig:(gi + 1) * cig] = w[gi * cog:(gi + 1) * cog].permute(0, 2, 3, 1)
            wf = torch.zeros(cout, kpad, dtype=torch.float32, device=w.device)
            wf[:, :kk] = full.reshape(cout, kk)
            cp[name] = wf.to(dt).contiguous()
            if need_t and name != "conv1":
                cp[name + "T"] = wf.t().contiguous().to(dt)
        for name in ("fc6", "fc7", "fc8", "fc9", "fc10", "fc11"):
            w = getattr(self, name).weight.detach()
            if name in ("fc6", "fc9"):
                w = w.view(4096, 256, 6, 6).permute(0, 2, 3, 1).reshape(4096, 9216)
            cp[name] = w.to(dt).contiguous()
            if need_t:
                cp[name + "T"] = w.t().contiguous().to(dt)
        self._copies = cp
        return cp

    def _buf(self, key, shape, dtype):
        dev = self.conv1.weight.device
        t = self._ws.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=dev)
            self._ws[key] = t
        return t

    # ---- pieces
    def _gemm(self, A, W, bias, Y, relu):
        ops.pwconv_gemm(A, W, Y, ops.bnref(A.shape[1]), 1, 0, bias=bias, out_act=L.ACT_RELU if relu else L.ACT_NONE, out_scale=1.0)

    def _forward_impl(self, x, training, masks=None):
        lib = L.lib()
        if not x.is_cuda:
            raise RuntimeError("SpacecraftPoseNet runs on the MI355X only (no CPU path)")
        B, _, H, W = x.shape
        dt, dc = self._dt(), (L.BF16 if self.precision == "bf16" else L.F32)
        cp = self._build_copies(need_t=training)
        x = x.contiguous().float()
        st = _st()
        sv = {"B": B}
        # conv1 .. conv5
        cur, Hc, Wc, Cc = None, H, W, 3
        for li, (name, cout, cin, g, k, stride, pad) in enumerate(_CONVS):
            OH, OW = (Hc + 2 * pad - k) // stride + 1, (Wc + 2 * pad - k) // stride + 1
            kpad = cp[name].shape[1]
            col = self._buf("col" + name, (B * OH * OW, kpad), dt)
            if li == 0:
                L.check(lib.spb_im2col_rgb(dc, _p(x), _p(col), B, Hc, Wc, k, k, stride, kpad, st), "spb_im2col_rgb")
            else:
                L.check(lib.spb_im2col(dc, _p(cur), _p(col), B, Hc, Wc, Cc, k, k, stride, pad, kpad, st), "spb_im2col")
            y = self._buf("y" + name, (B * OH * OW, cout), dt)
            self._gemm(col, cp[name], getattr(self, name).bias.detach().float(), y, relu=True)
            sv["col" + name], sv["y" + name], sv["in" + name] = col, y, (Hc, Wc, Cc)
            cur, Hc, Wc, Cc = y, OH, OW, cout
            if name in ("conv1", "conv2", "conv5"):
                PH, PW = (Hc - 3) // 2 + 1, (Wc - 3) // 2 + 1
                p = self._buf("p" + name, (B, PH, PW, Cc), dt)
                arg = self._buf("arg" + name, (B, PH, PW, Cc), torch.uint8)
                L.check(lib.spb_maxpool3s2_fwd(dc, _p(cur), _p(p), _p(arg), B, Hc, Wc, Cc, st), "spb_maxpool3s2_fwd")
                sv["pool" + name] = (p, arg, Hc, Wc)
                cur, Hc, Wc = p, PH, PW
                if name != "conv5":
                    n = self._buf("n" + name, (B, PH, PW, Cc), dt)
                    L.check(lib.spb_lrn2_fwd(dc, _p(cur), _p(n), B * PH * PW, Cc, LRN_ALPHA, LRN_BETA, LRN_K, st), "spb_lrn2_fwd")
                    sv["lrn" + name] = (cur, n)
                    cur = n
        f = cur.view(B, Hc * Wc * Cc)   # NHWC flatten; fc6 / fc9 columns are permuted accordingly
        if f.shape[1] != 9216:
            raise ValueError("SpacecraftPoseNet needs 227x227 inputs (pool5 must be 6x6x256); got %dx%d" % (H, W))
        sv["f"] = f
        outs = []
        for hi, (a, b_, c_) in enumerate((("fc6", "fc7", "fc8"), ("fc9", "fc10", "fc11"))):
            h = f
            for name in (a, b_):
                y = self._buf("h" + name, (B, 4096), dt)
                self._gemm(h, cp[name], getattr(self, name).bias.detach().float(), y, relu=True)
                if training:
                    m = self._buf("m" + name, (B, 4096), torch.uint8)
                    given = 0
                    if masks is not None:
                        m.copy_(masks[name].to(torch.uint8)); given = 1
                    seed = self.dropout_seed * 1000003 + self._step * 16 + hi * 4 + (0 if name == a else 1)
                    L.check(lib.spb_dropout(dc, _p(y), _p(m), B * 4096, float(self.keep_prob), seed, given, st), "spb_dropout")
                sv["in" + name] = h; sv["h" + name] = y
                h = y
            o = self._buf("o" + c_, (B, self.num_classes), dt)
            self._gemm(h, cp[c_], getattr(self, c_).bias.detach().float(), o, relu=False)
            sv["in" + c_] = h
            outs.append(o)
        self._saved = sv if training else None
        return outs[0], outs[1]

    def forward(self, x, y=None):
        """spn.py:125-143: (class logits c, regression logits r), each [B, num_classes] on the device (float32)"""
        c, r = self._forward_impl(x, self.training)
        return c.float(), r.float()

    # ---- one training step's loss + gradients (trainer.py:146-177): loss = softCE(c, yClasses) + 10 softCE(r, yWeights)
    def loss_and_grads(self, x, y_classes, y_weights, masks=None):
        """Runs forward (training mode), the loss and the backward pass; gradients land in p.grad of every parameter.
        Returns a device tensor (loss, loss_class, loss_regress)."""
        lib = L.lib()
        dt, dc = self._dt(), (L.BF16 if self.precision == "bf16" else L.F32)
        c, r = self._forward_impl(x, True, masks)
        self._step += 1
        sv, cp = self._saved, self._copies
        B = sv["B"]
        st = _st()
        NC = self.num_classes
        out = torch.zeros(3, dtype=torch.float32, device=c.device)
        dcg, drg = self._buf("dc", (B, NC), dt), self._buf("dr", (B, NC), dt)
        L.check(lib.spb_softce(dc, _p(c), _p(y_classes.float().contiguous()), _p(dcg), _p(out), 1, B, NC, 1.0, st), "spb_softce")
        L.check(lib.spb_softce(dc, _p(r), _p(y_weights.float().contiguous()), _p(drg), _p(out), 2, B, NC, 10.0, st), "spb_softce")
        ident = ops.bnref

        def grads_of(name, g, xin):
            """weight / bias gradient of layer `name` from g [M][N] and its GEMM input xin [M][K] -> p.grad"""
            lay = getattr(self, name)
            N, K = g.shape[1], xin.shape[1]
            dW = self._buf("dW" + name, (N, K), torch.float32); dW.zero_()
            ops.pwconv_wgrad(g, xin, dW, ident(N), ident(K))
            db = self._buf("db" + name, (N,), torch.float32); db.zero_()
            L.check(lib.spb_colsum(dc, _p(g), _p(db), g.shape[0], N, st), "spb_colsum")
            return dW, db

        def set_grad(p, val):
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(val)

        scale = 1.0 / (1.0 - self.keep_prob)
        df = None
        for (a, b_, c_), g in ((("fc6", "fc7", "fc8"), dcg), (("fc9", "fc10", "fc11"), drg)):
            for name, prev in ((c_, b_), (b_, a), (a, None)):
                xin = sv["in" + name]
                dW, db = grads_of(name, g, xin)
                if name in ("fc6", "fc9"):   # back to the reference's NCHW-flatten column order
                    dW = dW.view(4096, 6, 6, 256).permute(0, 3, 1, 2).reshape(4096, 9216)
                set_grad(getattr(self, name).weight, dW); set_grad(getattr(self, name).bias, db)
                dx = self._buf("dx" + name, (B, xin.shape[1]), dt)
                ops.pwconv_gemm(g, cp[name + "T"], dx, ident(g.shape[1]), 1, 0, out_scale=1.0)
                if prev is not None:       # through inverted dropout and ReLU of the previous fc
                    gn = self._buf("g" + prev, (B, 4096), dt)
                    L.check(lib.spb_relu_bwd(dc, _p(dx), _p(sv["h" + prev]), None, _p(gn), B * 4096, scale, st), "spb_relu_bwd")
                    g = gn
                else:
                    if df is None:
                        df = dx
                    else:
                        df = df + dx       # the two heads meet at pool5's output (tiny: B x 9216)
        # trunk, last to first
        g_act = df.contiguous()            # gradient w.r.t. the current layer's (pooled / normalised) output, NHWC
        for name, cout, cin, grp, k, stride, pad in reversed(_CONVS):
            Hc, Wc, Cc = sv["in" + name]
            y = sv["y" + name]
            OH, OW = (Hc + 2 * pad - k) // stride + 1, (Wc + 2 * pad - k) // stride + 1
            if "lrn" + name in sv:
                xin, _ = sv["lrn" + name]
                t = self._buf("dl" + name, tuple(xin.shape), dt)
                L.check(lib.spb_lrn2_bwd(dc, _p(xin), _p(g_act), _p(t), xin.shape[0] * xin.shape[1] * xin.shape[2], xin.shape[3],
                                         LRN_ALPHA, LRN_BETA, LRN_K, st), "spb_lrn2_bwd")
                g_act = t
            if "pool" + name in sv:
                _, arg, PHin, PWin = sv["pool" + name]
                t = self._buf("dp" + name, (B, PHin, PWin, cout), dt)
                L.check(lib.spb_maxpool3s2_bwd(dc, _p(g_act), _p(arg), _p(t), B, PHin, PWin, cout, st), "spb_maxpool3s2_bwd")
                g_act = t
            g = self._buf("gy" + name, (B * OH * OW, cout), dt)
            L.check(lib.spb_relu_bwd(dc, _p(g_act), _p(y), None, _p(g), g.numel(), 1.0, st), "spb_relu_bwd")
            col = sv["col" + name]
            dW, db = grads_of(name, g, col)
            kk = k * k * cin
            full = dW[:, :kk].reshape(cout, k, k, cin)
            cog, cig = cout // grp, cin // grp
            wg = torch.cat([full[gi * cog:(gi + 1) * cog, :, :, gi * cig:(gi + 1) * cig] for gi in range(grp)], 0).permute(0, 3, 1, 2)
            set_grad(getattr(self, name).weight, wg); set_grad(getattr(self, name).bias, db)
            if name != "conv1":
                dcol = self._buf("dcol" + name, tuple(col.shape), dt)
                ops.pwconv_gemm(g, cp[name + "T"], dcol, ident(cout), 1, 0, out_scale=1.0)
                dx = self._buf("dxin" + name, (B, Hc, Wc, Cc), dt)
                L.check(lib.spb_col2im(dc, _p(dcol), _p(dx), B, Hc, Wc, Cc, k, k, pad, col.shape[1], st), "spb_col2im")
                g_act = dx
        return out
