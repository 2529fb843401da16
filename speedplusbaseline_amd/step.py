"""One fused KRN / DANN training step on the MI355X: forward -> zero_grad -> backward -> [gradient all-reduce] ->
global-norm clip -> optimizer update, in the reference's order (trainer.py:72-98; dann.py:68-100), enqueued as HIP
kernels with no host synchronisation.  The whole forward+backward is two C calls that enqueue ~200 launches, which
keeps the MI355X busy without a graph; weight-gradient GEMMs go to a side stream and overlap the input-gradient chain
(KRN: 3.2 ms / step at B=48 bf16 at the start of round 3; DANN 5.06 ms).  A captured hipGraph replay is available
(use_graph=True, KRN 3.65 ms: cross-stream edges inside a graph cost more than they hide, so the side stream is switched off
there).

Eager mode passes the per-step scalars (lr, Adam bias corrections) by value in the optimizer launch; only the graph mode keeps
them in a 3-float device buffer refreshed by an async H2D copy before each replay (that copy and the bubble behind it were
17 us of every eager step).  Step-boundary work that nothing on the launch stream waits for rides on the plan's side stream:
zero_grad with the weight-copy refresh at the start of forward, the BatchNorm running-statistics update with the first
weight-gradient batch of backward.  The clip's sum of squares is 256 per-workgroup partials that every optimizer workgroup
adds up itself (the one-scalar form spent 13 of its 20 us on 512 same-address ticket atomics).
"""
import os

import torch

from . import ops
from .parallel import allreduce_sum_, allreduce_sum_async, mean_scale

_KIND_DEFAULT_EPS = 1e-8


class FusedTrainStep:
    def __init__(self, engine, batch, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0,
                 clip_value=0.0, dist_group=None, world_size=1, use_graph=False, dann=False):
        self.e = engine
        self.B = int(batch)
        self.kind = kind
        self.lr = float(lr)
        self.momentum = float(momentum)
        self.weight_decay = float(weight_decay)
        self.max_norm = float(max_norm)
        self.clip_value = float(clip_value)
        self.world = int(world_size)
        self.group = dist_group
        self.use_graph = bool(use_graph)
        self.dann = bool(dann)
        self.dann_overlap = os.environ.get("SPB_DANN_OVERLAP", "1") != "0"   # source / target passes on two streams
        if getattr(engine, "deterministic", False):
            self.dann_overlap = False      # reproducible mode: one stream, both passes into the one (exactly accumulated) gradient arena
        self._g2 = self._s2 = self._fork = None
        self._works = None
        # data-parallel: the arena tail is all-reduced while blocks 13..1 are still in backward (eager mode, plain KRN)
        mode = os.environ.get("SPB_DDP_OVERLAP", "1")      # "0": one all-reduce after backward; "force": also with one rank (tests)
        self._force = mode == "force"
        self._overlap = ((world_size > 1 or self._force) and not self.dann and not use_graph and mode != "0")
        if self._overlap:
            self._split = engine.bucket_split()
            self._comm = torch.cuda.Stream(device=engine.device)
            engine.set_bucket(batch, 0, True)
        dev = engine.device
        n = engine.n_params
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sqp = torch.zeros(ops.SQ_PARTS, dtype=torch.float32, device=dev)
        ops.grad_sqnorm(engine.grads, self.sq)     # first call allocates the library's reduction scratch (never inside a graph capture)
        self.gmul = torch.full((1,), mean_scale(self.world), dtype=torch.float32, device=dev) if self.world > 1 else None
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        # pinned ring for the per-step scalars: the host may run several steps ahead of the GPU (nothing below synchronises),
        # so the slot of step t is rewritten only after the event recorded behind its upload has completed
        self._hslots = 8
        self.hyper_host = torch.zeros(self._hslots, 3, dtype=torch.float32).pin_memory()
        self._hyper_ev = [None] * self._hslots
        self.t = 0
        self._graphs = None
        self._static = None
        self.last_scalars = None

    # optimizer hyper-parameters of step t (build.py:60-78: cfg.momentum doubles as RMSprop alpha / Adam beta1)
    def _betas(self):
        if self.kind == "rmsprop":
            return 0.0, self.momentum
        if self.kind in ("adam", "adamw"):
            return self.momentum, 0.999
        return self.momentum, 0.0

    def _refresh_hyper(self):
        if not self.use_graph:
            return                      # eager: lr / bias corrections travel by value in the optimizer launch
        b1, b2 = self._betas()
        i = self.t % self._hslots
        ev = self._hyper_ev[i]
        if ev is not None:
            ev.synchronize()            # the upload that last read this slot (8 steps ago) is done
        row = self.hyper_host[i]
        row[0] = self.lr
        row[1] = 1.0 - b1 ** self.t if b1 > 0 else 1.0
        row[2] = 1.0 - b2 ** self.t if b2 > 0 else 1.0
        self.hyper.copy_(row, non_blocking=True)
        if ev is None:
            ev = self._hyper_ev[i] = torch.cuda.Event()
        ev.record()

    # ---- the three phases (phase 2, the collective, is never captured)
    def _fwd_bwd(self, x, y, xt=None, alpha=0.0):
        e = self.e
        if not self.dann:
            side = not self.use_graph   # graph capture keeps every launch on one stream
            _, scal, _ = e.forward(x, y, training=True, slot=0, zero_grads=side, defer_running=side)
            if not side:
                ops.arena_zero(e.grads)
            e.backward(self.B, slot=0)
            if self._overlap:   # early bucket: all-reduce on the communication stream beside the rest of the backward
                e.wait_bucket(self.B, 0, self._comm)
                with torch.cuda.stream(self._comm):
                    self._works = [allreduce_sum_async(e.grads[self._split:], self.group, self._force)]
                self._works.append(allreduce_sum_async(e.grads[:self._split], self.group, self._force))   # after backward, launch stream
            return scal
        # DANN: zero_grad, source pass (pose + domain=1), target pass (domain=0), one backward of the sum (dann.py:81-95).
        # The two passes only meet in the gradient arena and the BatchNorm running statistics, and both are chains of
        # ~100 dependent, latency-bound launches at bs=16: they run CONCURRENTLY on two streams.  The target pass
        # accumulates into its own gradient arena (summed in afterwards) and leaves its running-statistics update to the
        # main stream, so the shared buffers see source first, then target, as in the reference.
        if not self.dann_overlap:
            ops.arena_zero(e.grads)
            _, scal, dom_s = e.forward(x, y, training=True, slot=0, domain=True)
            loss_s, dl_s = e.bce_logits(dom_s, 1.0)
            _, _, dom_t = e.forward(xt, None, training=True, slot=1, domain=True)
            loss_t, dl_t = e.bce_logits(dom_t, 0.0)
            e.backward(self.B, slot=0, with_pose=True, dlogit=dl_s, alpha=alpha)
            e.backward(self.B, slot=1, with_pose=False, dlogit=dl_t, alpha=alpha)
            return torch.cat([scal, loss_s, loss_t])
        main = torch.cuda.current_stream()
        if self._g2 is None:
            self._g2 = torch.zeros_like(e.grads)
            self._s2 = torch.cuda.Stream(device=e.device)
        e.prepare_weights()
        ops.arena_zero(e.grads); ops.arena_zero(self._g2)
        if self._fork is None:
            self._fork = ops.StreamFork()
        self._fork(self._s2, main)          # the target pass continues behind the launch stream: no event record there (ops.StreamFork)
        with torch.cuda.stream(self._s2):
            _, _, dom_t = e.forward(xt, None, training=True, slot=1, domain=True, prepare=False, update_running=False)
            loss_t, dl_t = e.bce_logits(dom_t, 0.0)
            e.backward(self.B, slot=1, grads=self._g2, with_pose=False, dlogit=dl_t, alpha=alpha)
        _, scal, dom_s = e.forward(x, y, training=True, slot=0, domain=True, prepare=False)
        loss_s, dl_s = e.bce_logits(dom_s, 1.0)
        e.backward(self.B, slot=0, with_pose=True, dlogit=dl_s, alpha=alpha)
        main.wait_stream(self._s2)
        for t_ in (dom_t, loss_t, dl_t, xt):
            t_.record_stream(main)
        e.update_running(self.B, slot=1)
        ops.arena_add(e.grads, self._g2)
        return torch.cat([scal, loss_s, loss_t])

    def _allreduce(self):
        if self._works is not None:      # issued from _fwd_bwd, overlapped with backward
            for w in self._works:
                if w is not None:
                    w.wait()             # the launch stream waits for the communicator's stream
            self._works = None
        elif self.world > 1:
            allreduce_sum_(self.e.grads, self.group)  # RCCL sum; the 1/world mean is folded into gmul

    # GradScaler defaults (torch.cuda.amp.GradScaler(): growth 2, backoff 0.5, growth interval 2000; the initial scale lives in engine.amp)
    amp_growth, amp_backoff, amp_interval = 2.0, 0.5, 2000

    def _update_fp16(self):
        """float16 (the reference's autocast + GradScaler recipe for KRN, trainer.py:73-98): scaler.unscale_ + inf / nan check, clip_grad_norm_
        on the unscaled gradient, optimizer.step() or nothing at all, scaler.update() -- on the device, no host read of found_inf."""
        import ctypes as C
        from . import _lib as L
        e = self.e
        b1, b2 = self._betas()
        lib, st = e.lib, C.c_void_p(torch.cuda.current_stream().cuda_stream)
        amp = e.amp
        L.check(lib.spb_amp_check(C.c_void_p(e.grads.data_ptr()), e.grads.numel(), C.c_void_p(amp.data_ptr()), st), "spb_amp_check")
        L.check(lib.spb_amp_step(C.c_void_p(amp.data_ptr()), float(self.lr), float(b1), float(b2), float(self.amp_growth), float(self.amp_backoff),
                                 int(self.amp_interval), st), "spb_amp_step")
        gm = amp[L.AMP_INV_SCALE:L.AMP_INV_SCALE + 1]
        if self.world > 1:
            gm = gm * mean_scale(self.world)
        if self.max_norm > 0:
            ops.grad_sqnorm_partials(e.grads, self.sqp)
        ops.optim_step(self.kind, e.params, e.grads, m=self.m, v=self.v, sq_partials=self.sqp if self.max_norm > 0 else None, gmul=gm,
                       lr=self.lr, beta1=b1, beta2=b2, eps=_KIND_DEFAULT_EPS, weight_decay=self.weight_decay, max_norm=self.max_norm,
                       clip_value=self.clip_value, step=max(self.t, 1), first_step=False, hyper=amp[L.AMP_LR:L.AMP_LR + 3],
                       skip=amp[L.AMP_SKIP:L.AMP_SKIP + 1])

    def _update(self, plain=False):
        """plain=True: the gradients in the arena are already unscaled (the generic loss.backward() path with the caller's own GradScaler)"""
        e = self.e
        if getattr(e, "half", False) and not plain:
            return self._update_fp16()
        b1, b2 = self._betas()
        if self.max_norm > 0:
            ops.grad_sqnorm_partials(e.grads, self.sqp)
        ops.optim_step(self.kind, e.params, e.grads, m=self.m, v=self.v, sq_partials=self.sqp if self.max_norm > 0 else None,
                       gmul=self.gmul, lr=self.lr, beta1=b1, beta2=b2, eps=_KIND_DEFAULT_EPS,
                       weight_decay=self.weight_decay, max_norm=self.max_norm, clip_value=self.clip_value, step=max(self.t, 1),
                       first_step=False, hyper=self.hyper if self.use_graph else None)

    def static_inputs(self):
        """the graph's input buffers (write a batch into them to skip the per-step device copy)"""
        return None if self._static is None else (self._static["x"], self._static["y"])

    def _capture(self, x, y, xt, alpha):
        self._static = dict(x=x.clone(), y=y.clone(), xt=None if xt is None else xt.clone())
        s = self._static
        self.e.set_side_stream(self.B, 0, False)
        side = torch.cuda.Stream(device=self.e.device)
        side.wait_stream(torch.cuda.current_stream())
        # warm-up outside capture (lazy kernel attributes, context creation).  It is a real forward: undo its only
        # lasting side effect, the BatchNorm running statistics / num_batches_tracked update.
        keep_buf, keep_nbt = self.e.buffers.clone(), self.e.nbt.clone()
        with torch.cuda.stream(side):
            self._fwd_bwd(s["x"], s["y"], s["xt"], alpha)
        torch.cuda.current_stream().wait_stream(side)
        self.e.buffers.copy_(keep_buf); self.e.nbt.copy_(keep_nbt)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            scal = self._fwd_bwd(s["x"], s["y"], s["xt"], alpha)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            self._update()
        self._graphs = (g1, g2, scal)

    def __call__(self, x, y, xt=None, alpha=0.0):
        """x [B,3,224,224], y [B,2,K] on the GPU (xt: target-domain images for DANN).  Returns a device tensor
        (loss, loss_x, loss_y[, loss_domain_source, loss_domain_target]); nothing is synchronised."""
        self.t += 1
        if self.kind == "sgd" and self.t == 1 and self.momentum != 0:
            self.m.zero_()  # buf_1 = g_1 == momentum*0 + g_1
        self._refresh_hyper()
        if self.use_graph and not self.dann:
            if self._graphs is None:
                self._capture(x, y, xt, alpha)
            s = self._static
            if x.data_ptr() != s["x"].data_ptr():
                s["x"].copy_(x, non_blocking=True)
            if y.data_ptr() != s["y"].data_ptr():
                s["y"].copy_(y, non_blocking=True)
            g1, g2, scal = self._graphs
            g1.replay()
            self._allreduce()
            g2.replay()
        else:
            scal = self._fwd_bwd(x, y, xt, alpha)
            self._allreduce()
            self._update()
        self.last_scalars = scal
        return scal
