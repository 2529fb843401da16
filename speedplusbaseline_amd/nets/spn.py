"""Spacecraft Pose Network on the MI355X (reference src/nets/spn.py:37-143; training step src/core/trainer.py:114-199).

Same constructor, attributes, state_dict keys (conv1..conv5, fc6..fc11 `.weight` / `.bias`) and return values as the
reference class; forward, loss and backward are HIP launches through the C-ABI (include/spb_hip.h):
every convolution is a matrix-core GEMM (spb_pwconv_gemm with the bias + ReLU epilogue, spb_pwconv_wgrad) on im2col'd NHWC
operands, with the pooling / LRN / soft-target cross-entropy kernels of csrc/spn.hip in between; grouped convolutions
(conv2/4/5, groups=2) run as one dense GEMM per group on that group's own column slab.  The fully connected layers (150 M of the 152 M parameters) are weight streams at
training batch sizes: in bf16 with B <= 64 they use the skinny kernels of csrc/spn_fc.hip (each weight element read once
per pass straight from a bf16 shadow arena the optimizer maintains; gradients written once, in place, into the flat
gradient arena).  f32 (parity mode) and larger batches go through the general GEMM kernels.  No CPU / eager fallback.

Memory: parameters and gradients are two flat f32 arenas in state-dict order (nn.Parameter.data / .grad are views), so the
clip + update is one launch and the data-parallel exchange one all-reduce.
"""
import ctypes as C
import os

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
    if reduction not in ("mean", "sum", "none"):
        raise ValueError("reduction must be 'mean', 'sum' or 'none' (spn.py:45-48)")
    B, Cn = logits.shape
    lg = logits.detach().contiguous()
    lg = lg if lg.dtype in (torch.float32, torch.bfloat16, torch.float16) else lg.float()
    tg = target.detach().float().contiguous()
    if reduction == "none":
        rows = torch.empty(B, dtype=torch.float32, device=logits.device)
        L.check(ops.lib_of(lg).spb_softce_rows(ops.dtype_code(lg), _p(lg), _p(tg), _p(rows), B, Cn, _st()), "spb_softce_rows")
        return rows
    out = torch.zeros(3, dtype=torch.float32, device=logits.device)
    L.check(ops.lib_of(lg).spb_softce(ops.dtype_code(lg), _p(lg), _p(tg), None, _p(out), 1, B, Cn, 1.0, _st()), "spb_softce")
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


_BACKGROUND_BLOCKS = int(os.environ.get("SPB_SPN_UPDATE_BLOCKS", "0"))   # > 0: cap on the workgroups of the heads' update beside backward
# conv1 straight from the image through an LDS band (spb_spn_stem): 48 us alone against 67 + 41 us for column matrix + GEMM, but
# 250-500 us beside the previous step's HBM-bound parameter update in every variant tried (a kernel with one workgroup per CU
# has too few requests in flight to get its share of a saturated memory system), so: 1 = evaluation only, 2 = always, 0 = never
_STEM = int(os.environ.get("SPB_SPN_STEM", "1"))
_FUSED_FC_UPDATE = os.environ.get("SPB_SPN_FUSED_FC_UPDATE", "0") == "1"


def _low_priority_stream(device):
    """lowest dispatch priority (spb_stream_create): its kernels take the compute units the launch stream leaves free.
    SPB_SPN_UPDATE_PRIORITY = -1 / 0 / 1 overrides the level for experiments"""
    h = C.c_void_p()
    with torch.cuda.device(device):
        L.check(L.lib().spb_stream_create(int(os.environ.get("SPB_SPN_UPDATE_PRIORITY", "1")), C.byref(h)), "spb_stream_create")
    return torch.cuda.ExternalStream(h.value, device=device)


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
        precision = precision or "fp32"       # None: the reference's default (train.py without --use_fp16)
        if precision not in ("bf16", "fp16", "fp32"):
            raise ValueError("precision must be bf16, fp16 or fp32, got %r" % (precision,))
        self.precision = precision
        self._half = precision in ("bf16", "fp16")     # 16-bit activations / weight shadows, f32 accumulation and master weights
        self._flat = self._gflat = self._shadow = None
        self._version = 0          # bumped whenever parameters change: compute copies are rebuilt lazily
        self._copies = None
        self._ws = {}
        self._saved = None
        self.dropout_seed = 2021
        self.side_wgrad = True     # weight gradients on a side stream (False: everything on the launch stream)
        self.concurrent_heads = os.environ.get("SPB_SPN_CONCURRENT_HEADS", "1") != "0"
        self.implicit_conv = os.environ.get("SPB_SPN_IMPLICIT", "1") != "0"   # conv2..5 as implicit GEMMs (bf16)
        self._step = 0
        if pretrain:
            self.load_weights('checkpoints/pretrained/bvlc_alexnet.npy')

    def load_weights(self, weight_path):
        """AlexNet trunk from a caffe-tensorflow `bvlc_alexnet.npy` (spn.py:104-123): a pickled dict layer -> [filters HWIO,
        bias]; only conv1..conv5 are taken (the fully connected layers start from their random init), filters go to
        nn.Conv2d's [Cout, Cin/groups, KH, KW] order."""
        import numpy as np
        blob = np.load(weight_path, allow_pickle=True, encoding='bytes').item()
        blob = {(k.decode() if isinstance(k, bytes) else k): v for k, v in blob.items()}
        with torch.no_grad():
            for layer in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5'):
                for arr in blob.get(layer, ()):
                    arr = np.asarray(arr)
                    is_filter = arr.ndim == 4
                    dst = getattr(self, layer).weight if is_filter else getattr(self, layer).bias
                    src = torch.from_numpy(np.ascontiguousarray(arr.transpose(3, 2, 0, 1) if is_filter else arr)).float()
                    if src.shape != dst.shape:
                        raise ValueError("%s: %s in %s does not fit %s" % (layer, tuple(src.shape), weight_path, tuple(dst.shape)))
                    dst.copy_(src)
        self.invalidate()

    def invalidate(self):
        """parameters were changed by someone other than SpnOptimizer: compute copies and the bf16 shadow are stale"""
        self.join_updates()
        self._version += 1

    # ---- flat arenas: parameter i lives at [off, off + numel) of self._flat (offsets multiples of 8 elements)
    def _ensure_arena(self):
        if self._flat is not None:
            return
        dev = self.conv1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("SpacecraftPoseNet runs on the MI355X only (no CPU path)")
        offs, off = {}, 0
        for n, p_ in self.named_parameters():
            offs[n] = (off, p_.numel())
            off += (p_.numel() + 7) // 8 * 8
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p_ in self.named_parameters():
                o, k = offs[n]
                flat[o:o + k].copy_(p_.detach().reshape(-1).float())
                p_.data = flat[o:o + k].view(p_.shape)
                p_.grad = gflat[o:o + k].view(p_.shape)
        self._flat, self._gflat, self._offs = flat, gflat, offs
        self._shadow = torch.empty(off, dtype=self._dt(), device=dev) if self._half else None
        self._shadow_version = -1
        self._conv_end = offs["fc6.weight"][0]

    # ---- fp16: dynamic loss scaling on the device (GradScaler's arithmetic: init 65536, x2 after 2000 clean steps, x0.5 on overflow)
    def amp_state(self, init_scale=65536.0):
        """float32 [SPB_AMP_STATE] device tensor (include/spb_hip.h): loss scale, unscale factor of the step in flight, growth
        tracker, found_inf, the optimizer's own step count, lr / Adam bias corrections, skip flag.  Created on first use."""
        if getattr(self, "_amp", None) is None:
            self._ensure_arena()
            st = torch.zeros(L.AMP_STATE, dtype=torch.float32, device=self._flat.device)
            st[L.AMP_SCALE] = float(init_scale); st[L.AMP_INV_SCALE] = 1.0 / float(init_scale)
            self._amp = st
        pending = getattr(self, "_amp_pending", None)
        if pending is not None:           # SpnOptimizer.load_state_dict: the checkpointed scaler state (the model may have been on the CPU then)
            self._amp.copy_(pending.to(self._amp.device))
            self._amp_pending = None
        return self._amp

    def restore_amp_state(self, state):
        """float32 [SPB_AMP_STATE] host tensor from a checkpoint (SpnOptimizer.state_dict()['spn_fused']['amp']): written to the
        device state on its next use, i.e. before the next forward scales a loss gradient"""
        if state.numel() != L.AMP_STATE:
            raise ValueError("AMP state has %d floats, expected %d" % (state.numel(), L.AMP_STATE))
        self._amp_pending = state

    def loss_scale(self):
        """current loss scale (host float; synchronises) -- 1.0 outside fp16 mode"""
        return float(self.amp_state()[L.AMP_SCALE]) if self.precision == "fp16" else 1.0

    def _lib(self):
        """libspb_hip.so, or its IEEE-half twin libspb_hip_f16.so in fp16 mode (same entry points, csrc/common.h SPB_F16)"""
        return L.lib_for(self.precision)

    def flat_parameters(self):
        self._ensure_arena()
        self.join_updates()
        self.sync_sharded_params()       # collective when sharded steps left stale master slices
        return self._flat

    def flat_grads(self):
        self._ensure_arena()
        return self._gflat

    def _sh(self, name):
        """bf16 shadow view of parameter `name` (valid after _refresh)"""
        o, k = self._offs[name]
        return self._shadow[o:o + k]

    def optimizer_updated(self):
        """SpnOptimizer wrote new parameters AND their bf16 shadow"""
        self._version += 1
        self._shadow_version = self._version

    def load_state_dict(self, *a, **k):
        self.join_updates()
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate(); self._ws = {}; self._copies = None; self._flat = self._gflat = self._shadow = None
        return r

    # ---- compute-dtype copies: conv weights as [Cout][Kpad] in (ky,kx,c_full) order (block diagonal for groups),
    #      fc6/fc9 columns permuted from the reference's NCHW flatten to NHWC, plus the transposes the input gradients need
    def _dt(self):
        return {"bf16": torch.bfloat16, "fp16": torch.float16}.get(self.precision, torch.float32)

    def _implicit(self):
        """bf16: conv2..conv5 run as implicit GEMMs (no column matrix); float32 keeps the im2col + GEMM path"""
        return self._half and self.implicit_conv

    def _conv(self, X, Wp, bias, mask, Y, B, H, W, Cx, k, stride, pad, groups, Cg, Ng, relu):
        a = L.SpnConvArgs()
        a.X, a.Wp, a.bias, a.mask, a.Y = _p(X).value, _p(Wp).value, _p(bias).value, _p(mask).value, _p(Y).value
        a.B, a.H, a.W, a.Cx, a.KH, a.KW, a.stride, a.pad = B, H, W, Cx, k, k, stride, pad
        a.groups, a.Cg, a.Ng, a.Kp, a.relu = groups, Cg, Ng, Wp.shape[1], 1 if relu else 0
        L.check(self._lib().spb_spn_conv(C.byref(a), _st()), "spb_spn_conv")

    def _fast(self, B):
        return self._half and B <= 64 and self.num_classes % 8 == 0

    def _build_copies(self, need_t, fast):
        cpo = self._copies
        if cpo is not None and cpo["v"] == self._version and (cpo["t"] or not need_t) and (cpo["fc"] or fast):
            return cpo
        lib, st = self._lib(), _st()
        dt, dc = self._dt(), (L.BF16 if self._half else L.F32)
        cp = {"v": self._version, "t": need_t, "fc": not fast}
        jobs = (L.SpnPackJob * 12)()
        nj = 0
        for name, cout, cin, g, k, _, _ in _CONVS:
            kg = (k * k * (cin // g) + 7) // 8 * 8             # one column slab / one [cout/g][kg] operand per group
            implicit = name != "conv1" and self._implicit()
            wp = self._buf("wp" + name, (cout, kg), dt)
            # [g][kg][cout/g]: operand of the explicit-GEMM input gradient (float32 parity mode / implicit_conv off)
            wt = self._buf("wpT" + name, (g, kg, cout // g), dt) if (need_t and name != "conv1" and not implicit) else None
            q = jobs[nj]; nj += 1
            q.W, q.out, q.outT = _p(getattr(self, name).weight.detach()).value, _p(wp).value, _p(wt).value
            q.Cout, q.Cin, q.groups, q.KH, q.KW, q.Kp, q.mode, q.chw = cout, cin, g, k, k, kg, 0, 1 if name == "conv1" else 0
            cp[name] = wp
            if wt is not None:
                cp[name + "T"] = wt
            if need_t and implicit:
                # mirrored-tap weights of the input-gradient pass (csrc/spn_conv.hip): [cin][(tap', cout/g)]
                kd = (k * k * (cout // g) + 7) // 8 * 8
                wd = self._buf("wpD" + name, (cin, kd), dt)
                q = jobs[nj]; nj += 1
                q.W, q.out, q.outT = _p(getattr(self, name).weight.detach()).value, _p(wd).value, None
                q.Cout, q.Cin, q.groups, q.KH, q.KW, q.Kp, q.mode, q.chw = cout, cin, g, k, k, kd, 1, 0
                cp[name + "D"] = wd
        if self._implicit() and (_STEM == 2 or (_STEM == 1 and not need_t)):      # conv1 straight from the image: k' = (ci*11 + ky)*16 + kx band layout (spb_spn_stem)
            name, cout, cin, g, k, _, _ = _CONVS[0]
            kb = (cin * k + 1) // 2 * 32
            wb = self._buf("wb" + name, (cout, kb), dt)
            q = jobs[nj]; nj += 1
            q.W, q.out, q.outT = _p(getattr(self, name).weight.detach()).value, _p(wb).value, None
            q.Cout, q.Cin, q.groups, q.KH, q.KW, q.Kp, q.mode, q.chw = cout, cin, 1, k, k, kb, 2, 0
            cp[name + "b"] = wb
        L.check(lib.spb_spn_pack_jobs(dc, jobs, nj, st), "spb_spn_pack_jobs")      # one launch for all of them
        if fast:
            if self._shadow_version != self._version:     # load_state_dict / load_weights / manual edits: rare
                self.join_updates()
                self._shadow.copy_(self._flat)
                self._shadow_version = self._version
        else:
            for name in ("fc6", "fc7", "fc8", "fc9", "fc10", "fc11"):
                w = getattr(self, name).weight.detach()
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
        ops.pwconv_gemm(A, W, Y, ops.bnref(A.shape[1]), 0, 0, bias=bias, out_act=L.ACT_RELU if relu else L.ACT_NONE, out_scale=1.0)

    def _epi(self, B, F, mode, accT=None, src=None, bias=None, H=None, Y=None, YT=None, mask=None, db=None, relu=0, p=0.0, scale=1.0,
             seed=0, given=0):
        a = L.FcEpiArgs()
        a.accT, a.src, a.bias, a.H, a.Y, a.YT, a.mask, a.db = (_p(accT).value, _p(src).value, _p(bias).value, _p(H).value, _p(Y).value,
                                                                _p(YT).value, _p(mask).value, _p(db).value)
        a.M, a.F, a.mode, a.relu, a.p, a.scale, a.seed, a.mask_given = B, F, mode, relu, p, scale, seed, given
        L.check(self._lib().spb_fc_epilogue(C.byref(a), _st()), "spb_fc_epilogue")

    def _acc(self, key, F, MP):
        """zero-initialised feature-major f32 accumulator; the epilogue kernels hand it back zeroed"""
        t = self._ws.get(key)
        if t is None or t.shape != (F, MP):
            t = torch.zeros(F, MP, dtype=torch.float32, device=self.conv1.weight.device)
            self._ws[key] = t
        return t

    def _trunk(self, x, cp, sv):
        lib, st = self._lib(), _st()
        B, _, H, W = x.shape
        dt, dc = self._dt(), (L.BF16 if self._half else L.F32)
        x = x.contiguous().float()
        cur, Hc, Wc, Cc = None, H, W, 3
        for li, (name, cout, cin, g, k, stride, pad) in enumerate(_CONVS):
            OH, OW = (Hc + 2 * pad - k) // stride + 1, (Wc + 2 * pad - k) // stride + 1
            kg = cp[name].shape[1]
            kpad, cog = kg * g, cout // g
            y = self._buf("y" + name, (B * OH * OW, cout), dt)
            bias = getattr(self, name).bias.detach()
            if li > 0 and self._implicit():
                self._conv(cur, cp[name], bias, None, y, B, Hc, Wc, Cc, k, stride, pad, g, cin // g, cog, relu=True)
                sv["x" + name] = cur
            elif li == 0 and (name + "b") in cp:
                # straight from the float32 NCHW image; backward builds the column matrix of its weight gradient on the side stream
                wb = cp[name + "b"]
                L.check(lib.spb_spn_stem(_p(x), _p(wb), _p(bias), _p(y), B, Hc, Wc, k, k, stride, cout, wb.shape[1], 1, st), "spb_spn_stem")
                sv["image"] = x
            else:
                col = self._buf("col" + name, (B * OH * OW, kpad), dt)
                if li == 0:
                    L.check(lib.spb_im2col_rgb(dc, _p(x), _p(col), B, Hc, Wc, k, k, stride, kpad, st), "spb_im2col_rgb")
                else:
                    L.check(lib.spb_im2col(dc, _p(cur), _p(col), B, Hc, Wc, Cc, k, k, stride, pad, kpad, g, st), "spb_im2col")
                for gi in range(g):      # one dense GEMM per convolution group on its column slab
                    self._gemm(col[:, gi * kg:(gi + 1) * kg], cp[name][gi * cog:(gi + 1) * cog], bias[gi * cog:(gi + 1) * cog],
                               y[:, gi * cog:(gi + 1) * cog], relu=True)
                sv["col" + name] = col
            sv["y" + name], sv["in" + name] = y, (Hc, Wc, Cc)
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
        if Hc * Wc * Cc != 9216:
            raise ValueError("SpacecraftPoseNet needs 227x227 inputs (pool5 must be 6x6x256); got %dx%d" % (H, W))
        return cur

    def _drop_seed(self, hi, second):
        return self.dropout_seed * 1000003 + self._step * 16 + hi * 4 + (1 if second else 0)

    def _forward_impl(self, x, training, masks=None):
        lib = self._lib()
        if not x.is_cuda:
            raise RuntimeError("SpacecraftPoseNet runs on the MI355X only (no CPU path)")
        self._ensure_arena()
        B = x.shape[0]
        fast = self._fast(B)
        dt, dc = self._dt(), (L.BF16 if self._half else L.F32)
        cp = self._build_copies(need_t=training, fast=fast)
        st = _st()
        sv = {"B": B, "fast": fast}
        p5 = self._trunk(x, cp, sv)                 # pool5 output, NHWC [B, 6, 6, 256]
        self.join_updates()                         # the previous step's update of the heads ran beside the trunk
        NC = self.num_classes
        outs = []
        if fast:
            MP = 32 if B <= 32 else 64
            f, fT = self._buf("f", (B, 9216), dt), self._buf("fT", (9216, MP), dt)
            L.check(lib.spb_spn_flatten(_p(p5), _p(f), _p(fT), B, 36, 256, st), "spb_spn_flatten")
            sv["f"], sv["fT"], sv["MP"] = f, fT, MP
            outs = [None, None]
            for hi, names in ((1, ("fc9", "fc10", "fc11")), (0, ("fc6", "fc7", "fc8"))):   # the forked head is enqueued first
              with self._head_ctx(hi):       # the two heads are independent chains of weight-streaming kernels: side by side
                acc = self._acc("acc%d" % hi, max(4096, NC), MP)
                h, K = f, 9216
                for j, name in enumerate(names):
                    N = NC if j == 2 else 4096
                    L.check(lib.spb_fc_fwd(_p(h), _p(self._sh(name + ".weight")), _p(acc), B, N, K, _st()), "spb_fc_fwd")
                    y = self._buf("h" + name, (B, N), dt)
                    bias = getattr(self, name).bias.detach()
                    if j == 2:
                        self._epi(B, N, 0, accT=acc, bias=bias, Y=y)
                    else:
                        yT = self._buf("hT" + name, (N, MP), dt)
                        m = None
                        pdrop, given = 0.0, 0
                        if training:
                            m = self._buf("m" + name, (B, N), torch.uint8)
                            pdrop = float(self.keep_prob)
                            if masks is not None:
                                m.copy_(masks[name].to(torch.uint8)); given = 1
                        self._epi(B, N, 0, accT=acc, bias=bias, Y=y, YT=yT, mask=m, relu=1, p=pdrop, seed=self._drop_seed(hi, j == 1),
                                  given=given)
                        sv["hT" + name] = yT
                    sv["in" + name], sv["h" + name] = h, y
                    h, K = y, N
                outs[hi] = h
            self._join_heads()
        else:
            # reference flatten order (NCHW): a small permuting copy on this general path
            f = p5.reshape(B, 36, 256).permute(0, 2, 1).reshape(B, 9216).contiguous()
            sv["f"] = f
            for hi, (a, b_, c_) in enumerate((("fc6", "fc7", "fc8"), ("fc9", "fc10", "fc11"))):
                h = f
                for name in (a, b_):
                    y = self._buf("h" + name, (B, 4096), dt)
                    self._gemm(h, cp[name], getattr(self, name).bias.detach(), y, relu=True)
                    if training:
                        m = self._buf("m" + name, (B, 4096), torch.uint8)
                        given = 0
                        if masks is not None:
                            m.copy_(masks[name].to(torch.uint8)); given = 1
                        L.check(lib.spb_dropout(dc, _p(y), _p(m), B * 4096, float(self.keep_prob), self._drop_seed(hi, name == b_), given, st),
                                "spb_dropout")
                    sv["in" + name] = h; sv["h" + name] = y
                    h = y
                o = self._buf("h" + c_, (B, NC), dt)
                self._gemm(h, cp[c_], getattr(self, c_).bias.detach(), o, relu=False)
                sv["in" + c_] = h
                outs.append(o)
        self._saved = sv if training else None
        return outs[0], outs[1]

    def forward(self, x, y=None):
        """spn.py:125-143: (class logits c, regression logits r), each [B, num_classes] on the device (float32)"""
        c, r = self._forward_impl(x, self.training)
        return c.float(), r.float()

    # ---- weight / bias gradients only feed the optimizer: they are queued and run on a side stream beside the input-gradient
    #      chain (forked at a few points only -- every fork costs the launch stream an event record), joined before returning
    def _on_side(self, fns, fc=False):
        """fc=True: the fully connected layers' weight gradients (six HBM-bound 20-50 us kernels) go to the regression head's
        stream -- its own three behind its input-gradient chain, the class head's three after the launch stream has joined that
        chain (loss_and_grads marks its end with an event: the launch stream does not wait for any of the six) -- so that the
        convolution weight gradients of the trunk start at once on the side stream.  Not a stream of their own: launch, head, side and update stream are four, and the HIP
        runtime multiplexes streams onto four hardware queues by default -- a fifth shared a queue with the 1.2 ms update
        and the step went from 1.72 to 2.52 ms."""
        if not fns:
            return
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self._gflat.device)
        if getattr(self, "_hs", None) is None:
            self._hs = torch.cuda.Stream(device=self._gflat.device)
        self._side_fc = self._hs
        if not self.side_wgrad:
            for f in fns:
                f(_st())
            return
        side = self._side_fc if fc else self._side
        self._fork(side)
        with torch.cuda.stream(side):
            sst = _st()
            for f in fns:
                f(sst)
        if fc:
            self._side_fc_used = True
        else:
            self._side_used = True

    def _fork(self, to):
        """stream `to` continues behind everything the current stream holds: ops.StreamFork (no event record on the current stream)"""
        f = getattr(self, "_fork_obj", None)
        if f is None:
            f = self._fork_obj = ops.StreamFork()
        f(to)

    def _side_streams_in_use(self):
        return ([self._side] if getattr(self, "_side_used", False) else []) + ([self._side_fc] if getattr(self, "_side_fc_used", False) else [])

    def _join_side(self):
        for sd in self._side_streams_in_use():
            torch.cuda.current_stream().wait_stream(sd)
        self._side_used = self._side_fc_used = False

    def _head_ctx(self, hi):
        """stream context of head `hi`: the class head stays on the launch stream, the regression head runs on a stream of its
        own forked here (each M <= 64 weight-streaming kernel alone reaches ~3 TB/s; two side by side share the HBM better,
        and the 5 us epilogue launches between them overlap)"""
        import contextlib
        if hi == 0 or not self.concurrent_heads:
            return contextlib.nullcontext()
        if getattr(self, "_hs", None) is None:
            self._hs = torch.cuda.Stream(device=self._gflat.device)
        self._fork(self._hs)
        self._heads_forked = True
        return torch.cuda.stream(self._hs)

    def _join_heads(self):
        if getattr(self, "_heads_forked", False):
            if getattr(self, "_heads_marked", False):    # backward: up to the end of the forked head's input-gradient chain
                torch.cuda.current_stream().wait_event(self._hev)
                self._heads_marked = False
            else:
                torch.cuda.current_stream().wait_stream(self._hs)
            self._heads_forked = False

    def _run_updates(self, optimizer, jobs, B):
        """the fully connected layers' weight gradient + parameter update kernels (SpnOptimizer.fused_fc_update) on a stream of
        their own, ordered after everything enqueued so far; the convolution weight gradients keep the side stream"""
        if getattr(self, "_upd", None) is None:
            self._upd = _low_priority_stream(self._gflat.device)
        self._fork(self._upd)
        with torch.cuda.stream(self._upd):
            for name, gT, xT in jobs:
                optimizer.fused_fc_update(name, gT, xT, B)
        self._early_on_upd = True

    def _shard_exchange(self, group, compress_bf16, lo, hi, optimizer, world_size):
        """on the communication stream: reduce-scatter gflat[lo:hi] (bfloat16 on the wire when compress_bf16), clip + update this
        rank's slice, all-gather the updated shadows (bf16 mode) or parameters (f32 mode).  Slices are 8-element aligned; the
        staging buffers are padded to world equal pieces."""
        import torch.distributed as dist
        from ..parallel import shard_slice
        rank = dist.get_rank(group)
        n = hi - lo
        per, my_lo, my_hi = shard_slice(lo, hi, rank, world_size)
        m = my_hi - my_lo
        part = self._gflat[lo:hi]
        wire = torch.bfloat16 if compress_bf16 else torch.float32
        stage = self._buf("shard_rs_%d" % lo, (per * world_size,), wire)
        if n < per * world_size:
            stage[n:].zero_()
        stage[:n].copy_(part)
        mine = self._buf("shard_rs_out_%d" % lo, (per,), wire)
        if dist.get_backend(group) == "nccl":                     # RCCL: a real reduce-scatter over all xGMI links
            dist.reduce_scatter_tensor(mine, stage, op=dist.ReduceOp.SUM, group=group)
        else:                                                     # gloo test rig (two ranks on one GPU): no reduce-scatter there
            dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=group)
            mine.copy_(stage[rank * per:(rank + 1) * per])
        if m > 0:
            self._gflat[my_lo:my_hi].copy_(mine[:m])
        optimizer.update_range_early(my_lo, my_hi, world_size, covers=(lo, hi), group=group)
        src = self._shadow if self._shadow is not None else self._flat
        g_in = self._buf("shard_ag_in_%d" % lo, (per,), src.dtype)
        if m > 0:
            g_in[:m].copy_(src[my_lo:my_hi])
        g_out = self._buf("shard_ag_%d" % lo, (per * world_size,), src.dtype)
        dist.all_gather_into_tensor(g_out, g_in, group=group)
        src[lo:hi].copy_(g_out[:n])
        if self._shadow is not None:      # the f32 masters of the other ranks' slices are stale until sync_sharded_params()
            self._shard_stale = getattr(self, "_shard_stale", {})
            self._shard_stale[lo] = (lo, hi, per, group, world_size)

    def sync_sharded_params(self):
        """COLLECTIVE (every rank, same order): after sharded steps in bf16 mode each rank holds the f32 master values of its own
        slices of the fully connected parameters only; this gathers them so that flat_parameters() / state_dict() are complete
        and identical on every rank.  No-op when nothing is stale."""
        stale, self._shard_stale = getattr(self, "_shard_stale", None), {}
        if not stale:
            return
        import torch.distributed as dist
        from ..parallel import shard_slice
        self.join_updates()
        if getattr(self, "_early_on_comm", False):
            torch.cuda.current_stream().wait_stream(self._comm)
        for lo, hi, per, group, world in stale.values():
            rank = dist.get_rank(group)
            n = hi - lo
            _, my_lo, my_hi = shard_slice(lo, hi, rank, world); m = my_hi - my_lo
            g_in = torch.zeros(per, dtype=torch.float32, device=self._flat.device)
            if m > 0:
                g_in[:m].copy_(self._flat[my_lo:my_hi])
            g_out = torch.empty(per * world, dtype=torch.float32, device=self._flat.device)
            dist.all_gather_into_tensor(g_out, g_in, group=group)
            self._flat[lo:hi].copy_(g_out[:n])

    def _start_exchange(self, group, compress_bf16, lo, hi, optimizer=None, world_size=1, sharded=False):
        """all-reduce of gflat[lo:hi] on the communication stream, ordered after everything enqueued so far (launch stream and
        the weight-gradient side stream).  compress_bf16: the bucket travels as bfloat16 -- half the bytes on the xGMI links;
        the sum is then rounded to bfloat16 per hop, like torch's bf16_compress_hook.  With an optimizer the bucket's
        parameters are updated on the communication stream as soon as the sum has arrived."""
        from ..parallel import allreduce_sum_async
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream(device=self._gflat.device)
        part = self._gflat[lo:hi]
        self._comm.wait_stream(torch.cuda.current_stream())
        for sd in self._side_streams_in_use():
            self._comm.wait_stream(sd)                # the fc weight gradients come from a side stream
        with torch.cuda.stream(self._comm):
            if sharded and optimizer is not None:
                self._shard_exchange(group, compress_bf16, lo, hi, optimizer, world_size)
                self._early_on_comm = True
                return ("done", None, None, part)
            if compress_bf16:
                buf = self._buf("ddp_bf16_%d" % lo, (part.numel(),), torch.bfloat16)
                buf.copy_(part)
                work = ("bf16", allreduce_sum_async(buf, group), buf, part)
            else:
                work = ("f32", allreduce_sum_async(part, group), None, part)
            if optimizer is None:
                return work
            self._land(work)
            optimizer.update_range_early(lo, hi, world_size)
            self._early_on_comm = True
            return ("done", None, None, part)

    @staticmethod
    def _land(work):
        """on the communication stream: wait for the collective, bring a compressed bucket back into the f32 arena"""
        kind, w, buf, part = work
        if w is not None:
            w.wait()
        if kind == "bf16":
            part.copy_(buf)

    def finish_early_updates(self):
        """SpnOptimizer.step: the launch stream waits for the parameter updates issued beside backward -- except the heads'
        update on the single-GPU path, which may run on into the next step's trunk (join_updates)"""
        if getattr(self, "_early_on_comm", False):
            torch.cuda.current_stream().wait_stream(self._comm)
            self._early_on_comm = False
        self._join_side()

    def fp16_check_operands(self):
        """the 16-bit [F][MP] gradient operands of the six fully connected weight gradients of the backward pass just run (fast path), or
        None when that pass took the generic path: what SpnOptimizer's inf / nan check reads instead of the f32 gradients they produce"""
        sv = self._saved
        if not sv or not sv.get("fast") or not getattr(self, "_bwd_fast", False):
            return None
        out = [self._ws.get("gT" + n) for n in ("fc6", "fc7", "fc8", "fc9", "fc10", "fc11")]
        return None if any(t is None for t in out) else out

    def update_heads_on_side_stream(self, fn):
        """runs fn() (the heads' share of the optimizer step) on the update stream, ordered after everything enqueued so far; the next
        forward waits for it before fc6 (join_updates), state_dict() / flat_parameters() / invalidate() too"""
        if getattr(self, "_upd", None) is None:
            self._upd = _low_priority_stream(self._gflat.device)
        self._fork(self._upd)
        with torch.cuda.stream(self._upd):
            fn()
        self._early_on_upd = True

    def join_updates(self):
        """The heads' parameter update started by loss_and_grads(optimizer=...) may still be in flight on its own stream after
        optimizer.step() returned: the next forward only needs the convolution parameters until pool5, so that stretch of
        the trunk runs beside it.  This makes the current stream wait for it.  Called by the forward pass before the first
        fully connected layer, by state_dict(), flat_parameters(), load_state_dict() and invalidate(); call it yourself
        before reading parameter tensors directly on another stream (torch.cuda.synchronize() also does)."""
        if getattr(self, "_early_on_upd", False):
            torch.cuda.current_stream().wait_stream(self._upd)
            self._early_on_upd = False

    def state_dict(self, *a, **k):
        self.join_updates()
        self.sync_sharded_params()       # collective when sharded steps left stale master slices (every rank must call it)
        return super().state_dict(*a, **k)

    def finish_gradient_exchange(self, group=None):
        """called by SpnOptimizer.step before the update (idempotent per backward pass): waits for the overlapped buckets and
        reduces what has not been exchanged yet (the convolution bucket; everything when no exchange was started)"""
        from ..parallel import allreduce_sum_
        if getattr(self, "_ddp_finished", True):
            return
        self._ddp_finished = True
        works, self._ddp_works = getattr(self, "_ddp_works", None), None
        if not works:
            allreduce_sum_(self._gflat, group)
            return
        for work in works:
            if work[0] != "done":
                with torch.cuda.stream(self._comm):
                    self._land(work)              # the communication stream waits for the collective
        torch.cuda.current_stream().wait_stream(self._comm)
        allreduce_sum_(self._gflat[:self._conv_end], group)

    # ---- one training step's loss + gradients (trainer.py:146-177): loss = softCE(c, yClasses) + 10 softCE(r, yWeights)
    def loss_and_grads(self, x, y_classes, y_weights, masks=None, world_size=1, group=None, compress_bf16=None, optimizer=None,
                       sharded=False):
        """Runs forward (training mode), the loss and the backward pass; gradients land in p.grad of every parameter
        (views of the flat gradient arena).  Returns a device tensor (loss, loss_class, loss_regress).

        world_size > 1 (one process per GPU): the fully connected layers' gradients -- the tail of the arena, 600 of the 609 MB,
        final before the convolution trunk's backward starts -- are summed across ranks on a communication stream while the
        trunk's backward runs, in two buckets (the class head's as soon as its weight gradients are queued, beside the regression
        head's backward; the regression head's beside the trunk); SpnOptimizer.step waits for them and reduces the small
        convolution bucket itself.  compress_bf16 (default: on in bf16 mode) sends those buckets as bfloat16: half the bytes
        on the xGMI links, the sum rounded to bfloat16 per hop like torch's bf16_compress_hook; False keeps float32.

        optimizer (the model's SpnOptimizer): the caller promises to call optimizer.step() next, as trainer.py:177-184 always
        does.  The two heads' parameters -- 98 % of the arena, an HBM-bound 0.8 ms update -- are then updated from inside
        this call as soon as their gradients are final, beside the rest of backward; optimizer.step() updates the convolution
        parameters.  The result is the same as without it: clip_grad_value_ and the update rules are elementwise.  On one GPU the heads' update runs
        on a stream of its own and may still be in flight when optimizer.step() returns -- the next forward needs only the
        convolution parameters until pool5 and waits for it there; see join_updates() for who else waits.

        sharded (world_size > 1, with optimizer): the fully connected buckets are REDUCE-SCATTERED instead of all-reduced; every
        rank clips and updates only its 1/world slice of the 150 M fully connected parameters (an 0.8 ms HBM-bound pass becomes
        0.8 / world) and the updated compute-dtype shadows (the f32 parameters in f32 mode) are ALL-GATHERED.  Same bytes on the
        wire as the all-reduce, but the two halves are separate collectives that use every xGMI link at once instead of one ring,
        and the gather only has to land before the next forward reaches fc6.  The f32 master copy of a slice lives on its owner
        only; sync_sharded_params() (collective; state_dict() / flat_parameters() call it) gathers the masters.  The convolution
        parameters (2 % of the arena) stay replicated and all-reduced."""
        if self.precision == "fp16":
            # GradScaler.step() is all or nothing: every gradient must exist (and be checked for inf / nan) before any parameter
            # moves, so the heads' early update and the sharded exchange are off; SpnOptimizer.step() does check + update
            optimizer, sharded = None, False
            compress_bf16 = False          # scaled float16-range gradients travel as float32
        if sharded and (world_size <= 1 or optimizer is None):
            sharded = False
        self._sharded = bool(sharded)
        lib = self._lib()
        dt, dc = self._dt(), (L.BF16 if self._half else L.F32)
        c, r = self._forward_impl(x, True, masks)
        self._step += 1
        self._ddp_finished = False       # SpnOptimizer.step(world_size > 1) exchanges whatever has not been exchanged yet
        self._ddp_works = []
        if compress_bf16 is None:
            compress_bf16 = self.precision == "bf16"     # (float16 gradients carry the loss scale: they travel as float32)
        head2_lo = self._offs["fc9.weight"][0]
        sv, cp = self._saved, self._copies
        B, fast = sv["B"], sv["fast"]
        st = _st()
        NC = self.num_classes
        out = torch.zeros(3, dtype=torch.float32, device=c.device)
        dcg, drg = self._buf("dc", (B, NC), dt), self._buf("dr", (B, NC), dt)
        ycf, ywf = y_classes.float().contiguous(), y_weights.float().contiguous()
        if not fast:
            if self.precision == "fp16":   # scaler.scale(loss).backward(): SpnOptimizer._step_fp16 divides every gradient by the scale
                sc = _p(self.amp_state()[L.AMP_SCALE:L.AMP_SCALE + 1])
                L.check(lib.spb_softce_scaled(dc, _p(c), _p(ycf), _p(dcg), _p(out), 1, B, NC, 1.0, sc, st), "spb_softce_scaled")
                L.check(lib.spb_softce_scaled(dc, _p(r), _p(ywf), _p(drg), _p(out), 2, B, NC, 10.0, sc, st), "spb_softce_scaled")
            else:
                L.check(lib.spb_softce(dc, _p(c), _p(ycf), _p(dcg), _p(out), 1, B, NC, 1.0, st), "spb_softce")
                L.check(lib.spb_softce(dc, _p(r), _p(ywf), _p(drg), _p(out), 2, B, NC, 10.0, st), "spb_softce")
        ident = ops.bnref
        scale = 1.0 / (1.0 - self.keep_prob)
        self._bwd_fast = bool(fast)
        self._gflat[:self._conv_end].zero_()          # conv bias gradients are accumulated with atomics
        if fast:
            MP = sv["MP"]
            accF = self._acc("accF", 9216, MP)       # both heads' gradient of the flattened pool5 output (float atomics)
            jobs = []
            # weight gradient + update of a layer in one kernel (spb_fc_wgrad_update): measured SLOWER than the two passes -- the
            # matrix-core output layout gives every lane 16 bytes of a different weight row, and seven streams with that pattern
            # run at a third of the arena-wide update's bandwidth (1.7 ms against 0.2 + 0.8 ms) -- so it is an experiment knob
            fuse = optimizer is not None and world_size == 1 and _FUSED_FC_UPDATE
            for hi, names, g, lg, tgt_, slot, wgt in ((1, ("fc9", "fc10", "fc11"), drg, r, ywf, 2, 10.0),     # forked head first
                                                      (0, ("fc6", "fc7", "fc8"), dcg, c, ycf, 1, 1.0)):
              with self._head_ctx(hi):       # loss gradient + input-gradient chain of each head on its own stream
                st = _st()
                acc = self._acc("acc%d" % hi, max(4096, NC), MP)
                pend = []
                if self.precision == "fp16":   # scaler.scale(loss).backward(): the gradient carries the loss scale, the loss does not
                    L.check(lib.spb_softce_scaled(dc, _p(lg), _p(tgt_), _p(g), _p(out), slot, B, NC, wgt,
                                                  _p(self.amp_state()[L.AMP_SCALE:L.AMP_SCALE + 1]), st), "spb_softce_scaled")
                else:
                    L.check(lib.spb_softce(dc, _p(lg), _p(tgt_), _p(g), _p(out), slot, B, NC, wgt, st), "spb_softce")
                a, b_, c_ = names
                gT = self._buf("gT" + c_, (NC, MP), dt)
                self._epi(B, NC, 1, src=g, YT=gT, db=getattr(self, c_).bias.grad)
                for name, prev, xT in ((c_, b_, sv["hT" + b_]), (b_, a, sv["hT" + a]), (a, None, sv["fT"])):
                    lay = getattr(self, name)
                    N, K = lay.weight.shape
                    if fuse:
                        jobs.append((name, gT, xT))
                    else:
                        pend.append(lambda s_, gT=gT, xT=xT, lay=lay, N=N, K=K:
                                    L.check(lib.spb_fc_wgrad(_p(gT), _p(xT), _p(lay.weight.grad), B, N, K, s_), "spb_fc_wgrad"))
                    tgt = acc if prev is not None else accF
                    L.check(lib.spb_fc_dgrad(_p(g), _p(self._sh(name + ".weight")), _p(tgt), B, N, K, st), "spb_fc_dgrad")
                    if prev is not None:     # through inverted dropout and ReLU of the previous fc, + its bias gradient
                        g = self._buf("g" + prev, (B, 4096), dt)
                        gT = self._buf("gT" + prev, (4096, MP), dt)
                        self._epi(B, 4096, 1, accT=acc, H=sv["h" + prev], Y=g, YT=gT, db=getattr(self, prev).bias.grad, scale=scale)
                if hi == 1 and getattr(self, "_heads_forked", False) and self.side_wgrad:
                    # (side_wgrad off: the three weight gradients below run on this head's stream right here, and the launch
                    # stream must wait for ALL of it -- _join_heads then waits for the stream, not for the event)
                    # the launch stream needs the forked head's INPUT-gradient chain only: mark its end before the three
                    # HBM-bound weight-gradient kernels (20-50 us each) are queued behind it on the same stream
                    if getattr(self, "_hev", None) is None:
                        self._hev = torch.cuda.Event()
                    self._hev.record(torch.cuda.current_stream())
                    self._heads_marked = True
                if hi == 1:
                    self._on_side(pend, fc=True)     # this head's weight gradients: behind its own chain, beside what follows
                else:
                    pend0 = pend                     # the class head's: queued after the join (they would sit in front of it)
            self._join_heads()
            self._on_side(pend0, fc=True)
            st = _st()
            if world_size > 1:               # the class head's bucket travels first, the regression head's follows below
                self._ddp_works.append(self._start_exchange(group, compress_bf16, self._conv_end, head2_lo, optimizer, world_size, sharded))
            g_act = self._buf("dp5", (B, 6, 6, 256), dt)   # the two heads met in accF
            L.check(lib.spb_spn_unflatten_grad(_p(accF), _p(g_act), B, 36, 256, st), "spb_spn_unflatten_grad")
            if fuse:
                self._run_updates(optimizer, jobs, B)       # beside the trunk's backward (and the next step's trunk forward)
            elif optimizer is not None and world_size == 1:
                # both heads' update beside the trunk's backward (and the next step's trunk forward), on a stream of its own --
                # the convolution weight gradients keep the side stream.  Not earlier: the second head's chain streams its
                # weights from HBM like the update does.
                if getattr(self, "_upd", None) is None:
                    self._upd = _low_priority_stream(self._gflat.device)
                self._fork(self._upd)
                for sd in self._side_streams_in_use():
                    self._upd.wait_stream(sd)
                with torch.cuda.stream(self._upd):
                    optimizer.update_range_early(self._conv_end, self._gflat.numel(), max_blocks=_BACKGROUND_BLOCKS)
                self._early_on_upd = True
        else:
            df = None
            for (a, b_, c_), g in ((("fc6", "fc7", "fc8"), dcg), (("fc9", "fc10", "fc11"), drg)):
                for name, prev in ((c_, b_), (b_, a), (a, None)):
                    lay = getattr(self, name)
                    xin = sv["in" + name]
                    N, K = g.shape[1], xin.shape[1]
                    lay.weight.grad.zero_(); lay.bias.grad.zero_()
                    ops.pwconv_wgrad(g, xin, lay.weight.grad, ident(N), ident(K))
                    L.check(lib.spb_colsum(dc, _p(g), _p(lay.bias.grad), B, N, st), "spb_colsum")
                    dx = self._buf("dx" + name, (B, K), dt)
                    ops.pwconv_gemm(g, cp[name + "T"], dx, ident(N), 0, 0, out_scale=1.0)
                    if prev is not None:
                        gn = self._buf("g" + prev, (B, 4096), dt)
                        L.check(lib.spb_relu_bwd(dc, _p(dx), _p(sv["h" + prev]), None, _p(gn), B * 4096, scale, st), "spb_relu_bwd")
                        g = gn
                    else:
                        df = dx if df is None else df + dx
            g_act = df.view(B, 256, 36).permute(0, 2, 1).contiguous()    # NCHW flatten order -> NHWC
        if world_size > 1:
            lo = head2_lo if self._ddp_works else self._conv_end
            self._ddp_works.append(self._start_exchange(group, compress_bf16, lo, self._gflat.numel(), optimizer if fast else None,
                                                        world_size, sharded and fast))
        # trunk, last to first
        masked = False
        if "image" in sv:       # conv1 ran without a column matrix: its weight gradient's operand, built beside the whole trunk
            name, cout, cin, grp, k, stride, pad = _CONVS[0]
            Hc, Wc, _ = sv["in" + name]
            OH, OW = (Hc - k) // stride + 1, (Wc - k) // stride + 1
            col1 = self._buf("col" + name, (B * OH * OW, cp[name].shape[1]), dt)
            self._on_side([lambda s_, col1=col1, Hc=Hc, Wc=Wc, k=k, stride=stride: L.check(
                lib.spb_im2col_rgb(dc, _p(sv["image"]), _p(col1), B, Hc, Wc, k, k, stride, col1.shape[1], s_), "spb_im2col_rgb")])
            sv["col" + name] = col1
        dwp = self._buf("dWp", (sum(cp[n].numel() for n, *_ in _CONVS),), torch.float32)
        dwp.zero_()
        woff = 0
        for name, cout, cin, grp, k, stride, pad in reversed(_CONVS):
            lay = getattr(self, name)
            Hc, Wc, Cc = sv["in" + name]
            y = sv["y" + name]
            OH, OW = (Hc + 2 * pad - k) // stride + 1, (Wc + 2 * pad - k) // stride + 1
            if "lrn" + name in sv:
                xin, _ = sv["lrn" + name]
                t = self._buf("dl" + name, tuple(xin.shape), dt)
                L.check(lib.spb_lrn2_bwd(dc, _p(xin), _p(g_act), _p(t), xin.shape[0] * xin.shape[1] * xin.shape[2], xin.shape[3],
                                         LRN_ALPHA, LRN_BETA, LRN_K, st), "spb_lrn2_bwd")
                g_act = t
            if "pool" + name in sv:         # pool right after this layer's ReLU: both backward passes in one kernel
                _, arg, PHin, PWin = sv["pool" + name]
                t = self._buf("dp" + name, (B, PHin, PWin, cout), dt)
                L.check(lib.spb_maxpool3s2_relu_bwd(dc, _p(g_act), _p(arg), _p(y), _p(t), B, PHin, PWin, cout, st), "spb_maxpool3s2_relu_bwd")
                g_act, masked = t, True
            if masked:          # the input-gradient kernel of the layer above already applied this layer's ReLU mask
                g = g_act.view(B * OH * OW, cout)
            else:
                g = self._buf("gy" + name, (B * OH * OW, cout), dt)
                L.check(lib.spb_relu_bwd(dc, _p(g_act), _p(y), None, _p(g), g.numel(), 1.0, st), "spb_relu_bwd")
            masked = False
            kg = cp[name].shape[1]
            kpad, cog = kg * grp, cout // grp
            col = sv.get("col" + name)
            xin_ = sv.get("x" + name)           # implicit-GEMM layer: its NHWC input instead of a column matrix
            dW = dwp[woff:woff + cout * kg].view(cout, kg)
            woff += cout * kg
            def conv_wgrad(s_, g=g, col=col, xin_=xin_, dW=dW, lay=lay, cout=cout, cin=cin, grp=grp, k=k, kg=kg, cog=cog, name=name,
                           Hc=Hc, Wc=Wc, Cc=Cc, stride=stride, pad=pad):
                if col is None:
                    a = L.SpnConvArgs()
                    a.X = _p(xin_).value
                    a.B, a.H, a.W, a.Cx, a.KH, a.KW, a.stride, a.pad = B, Hc, Wc, Cc, k, k, stride, pad
                    a.groups, a.Cg, a.Ng, a.Kp = grp, cin // grp, cog, kg
                    L.check(lib.spb_spn_conv_wgrad(C.byref(a), _p(g), _p(dW), s_), "spb_spn_conv_wgrad")
                if col is not None and grp == 1 and self._implicit():     # conv1: the same LDS-DMA kernel on its column matrix
                    L.check(lib.spb_spn_col_wgrad(_p(g), _p(col), _p(dW), g.shape[0], cout, kg, kg, kg, s_), "spb_spn_col_wgrad")
                    col = None
                for gi in range(grp if col is not None else 0):   # ops.* launch on the current stream = the side stream inside _on_side
                    ops.pwconv_wgrad(g[:, gi * cog:(gi + 1) * cog], col[:, gi * kg:(gi + 1) * kg], dW[gi * cog:(gi + 1) * cog], ident(cog), ident(kg))
                L.check(lib.spb_spn_unpack_conv_grad(_p(dW), _p(lay.weight.grad), cout, cin, grp, k, k, kg, 1 if name == "conv1" else 0, s_),
                        "spb_spn_unpack_conv_grad")
                if name != "conv1":
                    L.check(lib.spb_colsum(dc, _p(g), _p(lay.bias.grad), g.shape[0], cout, s_), "spb_colsum")
            self._on_side([conv_wgrad])
            if name == "conv1":     # the last layer: nothing follows on the launch stream, so its bias gradient runs there, beside
                L.check(lib.spb_colsum(dc, _p(g), _p(lay.bias.grad), g.shape[0], cout, st), "spb_colsum")    # the weight gradient
            if name != "conv1" and (name + "D") in cp:
                # implicit GEMM over (mirrored tap, output channel); conv5 -> conv4 -> conv3 feed each other directly, so the
                # ReLU mask of the layer below goes into the store
                below = {"conv5": "conv4", "conv4": "conv3"}.get(name)
                dx = self._buf("dxin" + name, (B, Hc, Wc, Cc), dt)
                self._conv(g, cp[name + "D"], None, sv["y" + below] if below else None, dx, B, OH, OW, cout, k, 1, k - 1 - pad, grp,
                           cog, Cc // grp, relu=False)
                masked = below is not None
                g_act = dx
            elif name != "conv1":
                dcol = self._buf("dcol" + name, tuple(col.shape), dt)
                for gi in range(grp):
                    ops.pwconv_gemm(g[:, gi * cog:(gi + 1) * cog], cp[name + "T"][gi], dcol[:, gi * kg:(gi + 1) * kg], ident(cog), 0, 0,
                                    out_scale=1.0)
                dx = self._buf("dxin" + name, (B, Hc, Wc, Cc), dt)
                L.check(lib.spb_col2im(dc, _p(dcol), _p(dx), B, Hc, Wc, Cc, k, k, pad, kpad, grp, st), "spb_col2im")
                g_act = dx
        self._join_side()          # every gradient is in the arena before the caller's next launch (the optimizer)
        return out
