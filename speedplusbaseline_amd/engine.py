"""Python side of the C++ network plan (csrc/krn_plan.hip): owns the flat f32 parameter / gradient / BN-buffer arenas
on the GPU and the per-batch-size activation workspaces, and exposes forward / backward as single asynchronous calls.

PyTorch is used for device memory and streams only.  There is no CPU or eager-PyTorch fallback: every entry raises if
the HIP library is missing or the tensors are not on the GPU.
"""
import ctypes as C

import torch

from . import _lib as L

# "fp16": the IEEE-half build of the library (libspb_hip_f16.so); its 16-bit storage code is the same SPB_BF16 slot of the C-ABI
PRECISIONS = {"fp32": L.F32, "f32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16, "fp16": L.BF16, "float16": L.BF16}
HALF = ("fp16", "float16")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KrnEngine:
    """One KeypointRegressionNet (optionally with RevGrad's domain classifier) bound to device arenas."""

    def __init__(self, num_keypoints, dann=False, deterministic=False):
        """deterministic=True: the reproducible twin library (libspb_hip_det.so; include/spb_hip.h "reproducible mode") -- batch sums and
        weight gradients are accumulated exactly (order-independent), every launch stays on the caller's stream: a training run is
        bit-identical in every process.  Slower (one flush launch per kernel launch); the reference offers no such mode
        (utils.py:297-298 sets cudnn.deterministic = False)."""
        self.deterministic = bool(deterministic)
        self.lib = L.lib_det() if self.deterministic else L.lib()
        self._det_regions = {}          # float pointer -> shadow tensor (kept alive while registered)
        self.half = False               # attach(device, "fp16"): IEEE-half build + device-side dynamic loss scale (self.amp)
        self.amp = None
        self.num_keypoints = int(num_keypoints)
        self.dann = bool(dann)
        self.h = C.c_void_p()
        L.check(self.lib.spb_krn_create(int(num_keypoints), 1 if dann else 0, C.byref(self.h)), "spb_krn_create")
        ti = L.TensorInfo()
        self.param_infos, self.buffer_infos, self.bn_names = [], [], []
        for i in range(self.lib.spb_krn_num_params(self.h)):
            L.check(self.lib.spb_krn_param_info(self.h, i, C.byref(ti)), "param_info")
            self.param_infos.append((ti.name.decode(), tuple(ti.shape[: ti.ndim]), int(ti.offset), int(ti.numel)))
        for i in range(self.lib.spb_krn_num_buffers(self.h)):
            L.check(self.lib.spb_krn_buffer_info(self.h, i, C.byref(ti)), "buffer_info")
            self.buffer_infos.append((ti.name.decode(), tuple(ti.shape[: ti.ndim]), int(ti.offset), int(ti.numel)))
        buf = C.create_string_buffer(96)
        for i in range(self.lib.spb_krn_num_bn(self.h)):
            L.check(self.lib.spb_krn_bn_name(self.h, i, buf), "bn_name")
            self.bn_names.append(buf.value.decode())
        self.n_params = int(self.lib.spb_krn_param_numel(self.h))
        self.n_buffers = int(self.lib.spb_krn_buffer_numel(self.h))
        self.device = None
        self.dtype_code = None
        self._ctx = {}

    def __del__(self):
        try:
            self._det_release()
            for h, _ws in self._ctx.values():
                self.lib.spb_krn_ctx_destroy(h)
            if self.h:
                self.lib.spb_krn_destroy(self.h)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ reproducible mode
    def _det_register(self, ptr, n_floats):
        shadow = torch.zeros(4 * int(n_floats), dtype=torch.int64, device=self.device)
        torch.cuda.synchronize(self.device)
        L.check(self.lib.spb_det_register(C.c_void_p(ptr), int(n_floats), _p(shadow)), "spb_det_register")
        self._det_regions[ptr] = shadow

    def _det_release(self):
        for ptr, shadow in list(getattr(self, "_det_regions", {}).items()):
            self.lib.spb_det_unregister_if(C.c_void_p(ptr), _p(shadow))    # (only OUR registration: the address may have a new owner by now)
        self._det_regions = {}

    def det_misses(self):
        """float atomics of the reproducible library that hit no registered region since the last call (0: the run was exact)"""
        if not self.deterministic:
            raise RuntimeError("not a deterministic engine")
        return int(self.lib.spb_det_misses())

    # ------------------------------------------------------------------------------------------------ arenas
    def attach(self, device, precision):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the KRN engine runs on the MI355X only; there is no CPU path (got device %s)" % device)
        code = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        self._det_release()
        for h, _ws in self._ctx.values():
            self.lib.spb_krn_ctx_destroy(h)
        self._ctx = {}
        half = isinstance(precision, str) and precision in HALF
        keep_amp = self.amp.detach().cpu() if (half and self.half and self.amp is not None) else None   # re-attach in the same precision
        if half != self.half:            # the 16-bit format is a compile-time property of the library: rebuild the plan handle on the other build
            if half and self.deterministic:
                raise RuntimeError("the reproducible library has no float16 build")
            self.lib.spb_krn_destroy(self.h)
            self.lib = L.lib_f16() if half else L.lib()
            self.h = C.c_void_p()
            L.check(self.lib.spb_krn_create(self.num_keypoints, 1 if self.dann else 0, C.byref(self.h)), "spb_krn_create")
            self.half = half
        self.device, self.dtype_code = device, code
        with torch.cuda.device(device):
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=device)
            self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=device)
            self.buffers = torch.zeros(self.n_buffers, dtype=torch.float32, device=device)
            self.nbt = torch.zeros(len(self.bn_names), dtype=torch.int64, device=device)
            self.wc = torch.empty(int(self.lib.spb_krn_wcompute_bytes(self.h, code)), dtype=torch.uint8, device=device)
            self.tables = torch.empty(int(self.lib.spb_krn_tables_bytes(self.h)), dtype=torch.uint8, device=device)
            torch.cuda.synchronize(device)
            L.check(self.lib.spb_krn_bind(self.h, _p(self.params), _p(self.grads), _p(self.buffers), _p(self.nbt),
                                          _p(self.wc), _p(self.tables), code), "spb_krn_bind")
            if self.deterministic:
                self._det_register(self.grads.data_ptr(), self.n_params)
                L.check(self.lib.spb_krn_set_det(self.h, 1), "spb_krn_set_det")
            # float16: GradScaler's state on the device (include/spb_hip.h SPB_AMP_*): loss scale 65536 (torch.cuda.amp.GradScaler()'s
            # default, train.py:101-104), its reciprocal, growth tracker, found_inf, steps taken, lr / bias corrections, skip flag
            self.amp = None
            if self.half:
                self.amp = torch.zeros(L.AMP_STATE, dtype=torch.float32, device=device)
                self.amp[L.AMP_SCALE] = 65536.0
                self.amp[L.AMP_INV_SCALE] = 1.0 / 65536.0
                if keep_amp is not None:      # a second model.to() (or a device move) must not reset the loss scale / step count
                    self.amp.copy_(keep_amp.to(device))
        return self

    def param_view(self, info, arena=None):
        _name, shape, off, numel = info
        a = self.params if arena is None else arena
        return a[off: off + numel].view(shape)

    def context(self, batch, slot=0):
        key = (int(batch), int(slot))
        if key not in self._ctx:
            nbytes = int(self.lib.spb_krn_ctx_bytes(self.h, key[0], self.dtype_code))
            if nbytes <= 0:
                raise L.SpbError("spb_krn_ctx_bytes failed: %d" % nbytes)
            with torch.cuda.device(self.device):
                ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                h = C.c_void_p()
                torch.cuda.synchronize(self.device)
                L.check(self.lib.spb_krn_ctx_create(self.h, key[0], _p(ws), C.byref(h)), "spb_krn_ctx_create")
                if self.deterministic:
                    sp, sn = C.c_void_p(), C.c_longlong()
                    L.check(self.lib.spb_krn_ctx_stats(h, C.byref(sp), C.byref(sn)), "spb_krn_ctx_stats")
                    self._det_register(sp.value, sn.value)
                    L.check(self.lib.spb_krn_ctx_set_det(h, 1), "spb_krn_ctx_set_det")
                if self.half:           # every backward on this context multiplies the device-side loss scale onto the upstream gradient
                    L.check(self.lib.spb_krn_ctx_set_loss_scale(h, C.c_void_p(self.amp.data_ptr() + 4 * L.AMP_SCALE)), "spb_krn_ctx_set_loss_scale")
            self._ctx[key] = (h, ws)
        return self._ctx[key][0]

    def drop_context(self, batch, slot=0):
        """free the workspace of one (batch size, slot) context -- and, in reproducible mode, its region of the library's exact-
        accumulation table (64 regions per process: runs with many distinct batch sizes would exhaust it otherwise)"""
        key = (int(batch), int(slot))
        if key in self._ctx:
            h, _ws = self._ctx.pop(key)
            if self.deterministic:
                sp, sn = C.c_void_p(), C.c_longlong()
                if self.lib.spb_krn_ctx_stats(h, C.byref(sp), C.byref(sn)) == 0 and sp.value in self._det_regions:
                    torch.cuda.synchronize(self.device)
                    self.lib.spb_det_unregister_if(C.c_void_p(sp.value), _p(self._det_regions[sp.value]))
                    del self._det_regions[sp.value]
            self.lib.spb_krn_ctx_destroy(h)

    # ------------------------------------------------------------------------------------------------ passes
    def _check_input(self, x):
        if self.device is None:
            raise RuntimeError("model is not on the GPU: call model.to('cuda') first (no CPU path exists)")
        if not x.is_cuda:
            raise RuntimeError("input images must be on the GPU (got %s)" % x.device)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != 224 or x.shape[3] != 224:
            raise RuntimeError("KRN needs [B,3,224,224] images so the 7x7 head reduces to 1x1 (park2019.py:139-142); got %s"
                               % (tuple(x.shape),))
        return x.detach().to(torch.float32).contiguous()

    def forward(self, x, target=None, training=True, slot=0, domain=False, prepare=True, update_running=True,
                zero_grads=False, defer_running=False):
        """Enqueue one forward.  Returns (pred [B,2K], scalars [3] or None, domain_logits [B] or None); x is kept
        alive by the caller/ctx until backward.  prepare=False: the compute-dtype weight copies are already current
        (second pass of a DANN step); update_running=False: training forward whose BatchNorm running-statistics update is
        applied later by update_running(batch, slot).  zero_grads=True: the gradient arena is zeroed on the context's side stream
        (optimizer.zero_grad() off the launch stream); defer_running=True: the running-statistics update is enqueued by the next
        backward() on this context, on its side stream (the fused train step: nothing in it reads the running statistics)."""
        x = self._check_input(x)
        B = x.shape[0]
        ctx = self.context(B, slot)
        with torch.cuda.device(self.device):
            st = _stream()
            pred = torch.empty(B, 2 * self.num_keypoints, dtype=torch.float32, device=self.device)
            scalars = None
            if target is not None:
                target = target.detach().to(device=self.device, dtype=torch.float32).contiguous()
                if tuple(target.shape) != (B, 2, self.num_keypoints):
                    raise RuntimeError("target must be [B,2,%d], got %s" % (self.num_keypoints, tuple(target.shape)))
                scalars = torch.empty(3, dtype=torch.float32, device=self.device)
            dom = torch.empty(B, dtype=torch.float32, device=self.device) if (domain and self.dann) else None
            mode = ((1 if update_running else 2) if training else 0) | (4 if prepare else 0)   # | 4: refresh the weight copies on the side stream
            if zero_grads:
                mode |= 8
            if defer_running and training and update_running:
                mode |= 16
            L.check(self.lib.spb_krn_forward(ctx, _p(x), _p(target), mode, _p(pred),
                                             _p(scalars), _p(dom), st), "spb_krn_forward")
        self._last_x = getattr(self, "_last_x", {})
        self._last_x[(B, slot)] = (x, target)
        return pred, scalars, dom

    def bucket_split(self):
        """element offset where the early gradient bucket (blocks 14..17, extras, head, domain classifier) starts"""
        return int(self.lib.spb_krn_bucket_split(self.h))

    def set_bucket(self, batch, slot=0, on=True):
        L.check(self.lib.spb_krn_ctx_set_bucket(self.context(batch, slot), 1 if on else 0), "spb_krn_ctx_set_bucket")

    def wait_bucket(self, batch, slot, stream):
        """`stream` (a torch.cuda.Stream) waits until the last backward finished the early bucket's gradients"""
        L.check(self.lib.spb_krn_ctx_wait_bucket(self.context(batch, slot), C.c_void_p(stream.cuda_stream)), "spb_krn_ctx_wait_bucket")

    def prepare_weights(self):
        """refresh the compute-dtype weight copies from the f32 parameter arena (forward() does this unless prepare=False)"""
        with torch.cuda.device(self.device):
            L.check(self.lib.spb_krn_prepare_weights(self.h, _stream()), "spb_krn_prepare_weights")

    def update_running(self, batch, slot=0):
        with torch.cuda.device(self.device):
            L.check(self.lib.spb_krn_update_running(self.context(batch, slot), _stream()), "spb_krn_update_running")

    def backward(self, batch, slot=0, grads=None, gscale=1.0, with_pose=True, dlogit=None, alpha=0.0):
        if with_pose and int(batch) > self.max_train_batch():
            raise RuntimeError("per-GPU training batch %d is above this build's limit of %d (%s): the head's weight-gradient "
                               "kernel keeps one [B,7,7,8] slab of the feature map in LDS; split the batch over more GPUs or "
                               "accumulate two half batches" % (batch, self.max_train_batch(),
                                                                "bf16" if self.dtype_code == PRECISIONS["bf16"] else "fp32"))
        ctx = self.context(batch, slot)
        with torch.cuda.device(self.device):
            L.check(self.lib.spb_krn_backward(ctx, _p(grads), float(gscale), 1 if with_pose else 0, _p(dlogit), float(alpha),
                                              _stream()), "spb_krn_backward")

    def activations(self, batch, slot=0):
        """{BatchNorm name: raw convolution output z of the last forward as an NCHW VIEW of the context workspace} -- what the parity
        tests hand to the oracle to take its backward pass through this forward state (tests/test_parity_conditioned_gpu.py)"""
        h, ws = self._ctx[(int(batch), int(slot))]
        dt = (torch.float16 if self.half else torch.bfloat16) if self.dtype_code == L.BF16 else torch.float32
        es = 2 if self.dtype_code == L.BF16 else 4
        # 16-bit modes: the expanded tensors of blocks 2-4 are virtual (never stored; csrc/krn_plan.hip, Runner::virt) -- written out here,
        # rounded to the storage type, from the operands the forward pass kept (virtual_activations() names them)
        L.check(self.lib.spb_krn_ctx_materialize(h, _stream()), "spb_krn_ctx_materialize")
        out, ai = {}, L.ActInfo()
        for a in range(self.lib.spb_krn_num_acts(self.h)):
            L.check(self.lib.spb_krn_ctx_act_info(h, a, C.byref(ai)), "spb_krn_ctx_act_info")
            n = int(batch) * ai.H * ai.W * ai.C
            z = ws[ai.z_off: ai.z_off + n * es].view(dt).view(int(batch), ai.H, ai.W, ai.C).permute(0, 3, 1, 2)
            out[self.bn_names[ai.bn_index][: -len(".num_batches_tracked")]] = z
        return out

    def virtual_activations(self, batch, slot=0):
        """names (as in activations()) of the BatchNorm'd tensors that exist only as batch sums in this context: every kernel that needs
        their values recomputes them on the matrix cores from the expand convolution's input (spb_dw_args_t::Xe)"""
        h, _ = self._ctx[(int(batch), int(slot))]
        ai, out = L.ActInfo(), []
        for a in range(self.lib.spb_krn_num_acts(self.h)):
            if self.lib.spb_krn_ctx_virtual(h, a) == 1:
                L.check(self.lib.spb_krn_ctx_act_info(h, a, C.byref(ai)), "spb_krn_ctx_act_info")
                out.append(self.bn_names[ai.bn_index][: -len(".num_batches_tracked")])
        return out

    def use_loss_scale(self, batch, slot=0, on=True):
        """float16: whether backward() on this context multiplies the device-side loss scale onto the upstream gradient (default: yes).
        Off for the generic loss.backward() path, where the caller's own GradScaler scales the loss."""
        if self.half:
            ptr = C.c_void_p(self.amp.data_ptr() + 4 * L.AMP_SCALE) if on else None
            L.check(self.lib.spb_krn_ctx_set_loss_scale(self.context(batch, slot), ptr), "spb_krn_ctx_set_loss_scale")

    def max_train_batch(self):
        """largest per-GPU batch spb_head_bwd accepts (csrc/stem_head.hip: B*32 floats of upstream gradient + a [B,49,8] slab of
        z in 160 KB of LDS): 179 in bf16, 96 in fp32.  The reference's recipes use 48 (README.md:87) and 16 (DANN)."""
        el = 2 if self.dtype_code == PRECISIONS["bf16"] else 4
        return (160 * 1024) // (32 * 4 + 49 * 8 * el)

    # ------------------------------------------------------------------------------------------------ live timing
    def set_side_stream(self, batch, slot=0, on=True):
        """weight-gradient GEMMs on the context's side stream (default) or on the launch stream (hipGraph capture)"""
        L.check(self.lib.spb_krn_ctx_set_side_stream(self.context(batch, slot), 1 if on else 0), "spb_krn_ctx_set_side_stream")

    def prof_enable(self, batch, slot=0, on=True):
        L.check(self.lib.spb_krn_prof_enable(self.context(batch, slot), 1 if on else 0), "spb_krn_prof_enable")

    def prof_read(self, batch, slot=0):
        """{kernel family: dict(launches, ms, bytes, flops)} summed over the launches since the last read"""
        n = self.lib.spb_krn_prof_num_categories()
        la = (C.c_int * n)(); ms = (C.c_float * n)(); by = (C.c_double * n)(); fl = (C.c_double * n)()
        L.check(self.lib.spb_krn_prof_read(self.context(batch, slot), la, ms, by, fl), "spb_krn_prof_read")
        out = {}
        for i in range(n):
            if la[i]:
                out[self.lib.spb_krn_prof_category_name(i).decode()] = dict(launches=int(la[i]), ms=float(ms[i]),
                                                                            bytes=float(by[i]), flops=float(fl[i]))
        return out

    def prof_launches(self, batch, slot=0, cap=512):
        """[(kernel family, ms, algorithmic bytes)] of every launch since the last prof_read, in launch order"""
        cat = (C.c_int * cap)(); ms = (C.c_float * cap)(); by = (C.c_double * cap)()
        n = self.lib.spb_krn_prof_launches(self.context(batch, slot), cap, cat, ms, by)
        if n < 0:
            L.check(n, "spb_krn_prof_launches")
        return [(self.lib.spb_krn_prof_category_name(cat[i]).decode(), float(ms[i]), float(by[i])) for i in range(min(n, cap))]

    def weight_prep_bytes(self):
        return int(self.lib.spb_krn_weight_prep_bytes(self.h))

    def bce_logits(self, logits, label, gscale=1.0):
        """mean BCE-with-logits against a constant label; returns (loss [1], dlogit [B])"""
        B = logits.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        dl = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.spb_bce_logits(_p(logits), float(label), B, _p(loss), _p(dl), float(gscale), _stream()),
                    "spb_bce_logits")
        return loss, dl
