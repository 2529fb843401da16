"""Optimizer front end: a torch.optim.Optimizer (so StepLR, checkpoints and the reference's loops keep working) whose
update -- together with the trainers' clip_grad_norm_(1.0) -- is one fused HIP pass over the model's flat f32 parameter
arena (reference build.py:60-78 for the four kinds; trainer.py:90,97 / dann.py:99 for the clip).

Two ways in:
  * train_step(images, target)   fast path used by speedplusbaseline_amd.core.{trainer,dann}: forward, zero_grad, backward,
                                 clip, update as HIP launches (optionally a replayed hipGraph), no autograd;
  * step()                       generic path after loss.backward() / clip_grad_norm_: gathers p.grad into the arena if
                                 autograd did not already leave them there, then runs the same update kernel.
"""
import torch

from . import _lib as L
from .step import FusedTrainStep


class FusedOptimizer(torch.optim.Optimizer):
    def __init__(self, params, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.0, model=None, max_norm=1.0,
                 use_graph=False):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, kind=kind)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FusedOptimizer works on the model's single parameter group (as the reference builds it)")
        self._model = model
        self._max_norm = max_norm
        self._use_graph = use_graph
        self._ts = None
        self._pending_state = None

    # ------------------------------------------------------------------------------------------------ binding
    def _bind(self, batch, dann=False, world_size=1, group=None):
        m = self._model
        if m is None:
            raise RuntimeError("FusedOptimizer needs the HIP-backed model it was built for (get_optimizer(cfg, model))")
        eng = m.engine()
        g = self.param_groups[0]
        ts = self._ts
        if ts is None or ts.e is not eng or ts.B != batch or ts.dann != dann:
            old = ts
            ts = FusedTrainStep(eng, batch, kind=g["kind"], lr=g["lr"], momentum=g["momentum"], weight_decay=g["weight_decay"],
                                max_norm=self._max_norm, dist_group=group, world_size=world_size,
                                use_graph=self._use_graph, dann=dann)
            if old is not None and old.e is eng:  # keep the moments when only the batch size changed
                ts.m.copy_(old.m); ts.v.copy_(old.v); ts.t = old.t
            self._ts = ts
            if self._pending_state is not None:
                self._load_into(ts, self._pending_state)
                self._pending_state = None
            if self._pending_amp is not None and getattr(eng, "amp", None) is not None:     # resumed float16 run: the saved loss scale / step count
                eng.amp.copy_(self._pending_amp.to(eng.amp.device))
                self._pending_amp = None
        ts.lr = float(g["lr"])  # StepLR edits param_groups between epochs
        return ts

    # ------------------------------------------------------------------------------------------------ fast path
    def train_step(self, images, target, target_images=None, alpha=0.0, world_size=1, group=None):
        """one whole training step on the GPU; returns the device tensor (loss, loss_x, loss_y[, bce_src, bce_tgt])"""
        ts = self._bind(images.shape[0], dann=target_images is not None, world_size=world_size, group=group)
        return ts(images, target, target_images, alpha)

    # ------------------------------------------------------------------------------------------------ generic path
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._model.engine()
        names = {id(p): n for n, p in self._model.named_parameters()}
        infos = {i[0]: i for i in eng.param_infos}
        ts = self._ts
        if ts is None or ts.e is not eng:
            ts = self._bind(1)
        ts.lr = float(self.param_groups[0]["lr"])
        # the clip (if any) already happened in the caller (clip_grad_norm_): plain update here
        eng.grads.zero_()
        for p in self.param_groups[0]["params"]:
            if p.grad is not None:
                eng.param_view(infos[names[id(p)]], eng.grads).copy_(p.grad)
        ts.t += 1
        ts._refresh_hyper()
        keep = ts.max_norm
        ts.max_norm = 0.0
        try:
            ts._update(plain=True)     # (float16: loss.backward() through the module leaves UNSCALED gradients -- the caller's GradScaler owns the scale)
        finally:
            ts.max_norm = keep
        return loss

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim-compatible layout: per-parameter step / exp_avg / exp_avg_sq (or momentum_buffer / square_avg)"""
        g = self.param_groups[0]
        kind = g["kind"]
        state = {}
        ts = self._ts
        if ts is not None and ts.t > 0:
            eng = ts.e
            names = {id(p): n for n, p in self._model.named_parameters()}
            infos = {i[0]: i for i in eng.param_infos}
            for idx, p in enumerate(g["params"]):
                info = infos[names[id(p)]]
                st = {"step": torch.tensor(float(ts.t))}
                if kind in ("adam", "adamw"):
                    st["exp_avg"] = eng.param_view(info, ts.m).clone(); st["exp_avg_sq"] = eng.param_view(info, ts.v).clone()
                elif kind == "rmsprop":
                    st["square_avg"] = eng.param_view(info, ts.v).clone()
                elif g["momentum"] != 0:
                    st["momentum_buffer"] = eng.param_view(info, ts.m).clone()
                state[idx] = st
        pg = {k: v for k, v in g.items() if k != "params"}
        pg["params"] = list(range(len(g["params"])))
        out = {"state": state, "param_groups": [pg]}
        # float16: GradScaler's state lives on the device with the engine (loss scale, growth tracker, steps actually taken); torch keeps it
        # in scaler.state_dict() -- here it travels with the optimizer, like the SPN optimizer's
        amp = getattr(ts.e, "amp", None) if ts is not None else None
        if amp is not None:
            out["spb_amp"] = amp.detach().float().cpu()
        elif self._pending_amp is not None:
            out["spb_amp"] = self._pending_amp.clone()
        return out

    _pending_amp = None

    def load_state_dict(self, sd):
        g = self.param_groups[0]
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                g[k] = v
        self._pending_amp = sd.get("spb_amp")
        if self._ts is not None:
            self._load_into(self._ts, sd["state"])
        else:
            self._pending_state = sd["state"]

    def _load_into(self, ts, state):
        eng = ts.e
        had_amp = False
        if self._pending_amp is not None and getattr(eng, "amp", None) is not None:
            eng.amp.copy_(self._pending_amp.to(eng.amp.device))
            self._pending_amp = None
            had_amp = True
        names = {id(p): n for n, p in self._model.named_parameters()}
        infos = {i[0]: i for i in eng.param_infos}
        for idx, p in enumerate(self.param_groups[0]["params"]):
            st = state.get(idx, state.get(str(idx)))
            if not st:
                continue
            info = infos[names[id(p)]]
            ts.t = int(float(st.get("step", ts.t)))
            for key, arena in (("exp_avg", ts.m), ("exp_avg_sq", ts.v), ("square_avg", ts.v), ("momentum_buffer", ts.m)):
                if key in st and st[key] is not None:
                    eng.param_view(info, arena).copy_(st[key].to(arena.device))
        # float16 run resumed from a checkpoint WITHOUT GradScaler state (written by a bf16 / fp32 run, or by the reference): the Adam bias
        # corrections follow the device-side count of steps actually taken (AMP_STEPS) -- seed it from the loaded step, or warm moments
        # would meet the corrections of step 1 and inflate the first updates
        if not had_amp and getattr(eng, "amp", None) is not None and ts.t > 0:
            eng.amp[L.AMP_STEPS] = float(ts.t)


class SpnOptimizer(torch.optim.Optimizer):
    """Optimizer of the Spacecraft Pose Network (reference build.py:60-78 kinds; trainer.py:177-184: clip_grad_value_(1.0)
    then step).  The model keeps its 152 M parameters and their gradients in two flat f32 arenas, so the whole update is ONE
    HIP pass: (data-parallel mean,) clamp of every gradient element to [-clip_value, clip_value], sgd / rmsprop / adam /
    adamw in f32, and the refreshed bf16 shadow the next forward streams its weights from.  With world_size > 1 the gradient
    arena is summed across ranks first (one all-reduce, RCCL on the MI355X node)."""

    def __init__(self, params, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.0, model=None, clip_value=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, kind=kind))
        if model is None:
            raise RuntimeError("SpnOptimizer needs the HIP-backed model it was built for (get_optimizer(cfg, model))")
        self._model = model
        self.clip_value = clip_value
        self._t = 0
        self._m = self._v = self._gmul = None
        self._early = []       # arena ranges already updated for the step in flight (update_range_early)
        self._shard_buckets = {}   # lo -> (lo, hi, world, group) of buckets whose moments are rank-sharded right now (group: the one the sharded step ran on)

    def _betas(self, kind, momentum):
        if kind == "rmsprop":
            return 0.0, momentum
        if kind in ("adam", "adamw"):
            return momentum, 0.999
        return momentum, 0.0

    # ---- rank-sharded optimizer state (SpacecraftPoseNet.loss_and_grads(sharded=True)): after a sharded step the moments of a
    #      fully connected bucket are current on their owner's slice only, like the f32 master parameters
    def gather_sharded_state(self, group=None):
        """COLLECTIVE (every rank, same order): all-gathers the moments of every bucket that was updated rank-sharded, and the
        model's f32 master slices (sync_sharded_params), so that m / v / parameters are complete and identical on all ranks.
        state_dict() calls it -- under a sharded run EVERY rank must therefore call optimizer.state_dict() / model.state_dict()
        (train.py does; only rank 0 writes the file).  No-op when nothing is sharded."""
        buckets, self._shard_buckets = self._shard_buckets, {}
        mdl = self._model
        mdl.sync_sharded_params()
        if not buckets or self._m is None:
            return
        import torch.distributed as dist
        from .parallel import shard_slice
        mdl.join_updates()
        if getattr(mdl, "_early_on_comm", False):
            torch.cuda.current_stream().wait_stream(mdl._comm)
        for lo, hi, world, bgroup in buckets.values():
            grp = group if group is not None else bgroup          # the collective runs on the group the sharded steps were issued on
            rank = dist.get_rank(grp)
            per, my_lo, my_hi = shard_slice(lo, hi, rank, world)
            n, k = hi - lo, my_hi - my_lo
            for arena in (self._m, self._v):
                g_in = torch.zeros(per, dtype=torch.float32, device=arena.device)
                if k > 0:
                    g_in[:k].copy_(arena[my_lo:my_hi])
                g_out = torch.empty(per * world, dtype=torch.float32, device=arena.device)
                dist.all_gather_into_tensor(g_out, g_in, group=grp)
                arena[lo:hi].copy_(g_out[:n])

    def _unshard_if_needed(self, lo, hi):
        """a NON-sharded update of [lo, hi) after sharded steps (e.g. a ragged last batch that takes the generic path) must see
        complete masters and moments there: gather first (collective -- every rank takes the same path, the batch shape decides it)"""
        if any(a < hi and lo < b for a, b, _, _ in self._shard_buckets.values()):
            self.gather_sharded_state()

    def _state(self, flat):
        if self._m is not None and self._m.numel() == flat.numel() and self._m.device != flat.device:
            self._m, self._v = self._m.to(flat.device), self._v.to(flat.device)       # restored from a checkpoint (CPU)
        if self._m is None or self._m.numel() != flat.numel():
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)

    def _update(self, lo, hi, world_size, max_blocks=0):
        """clamp + update + bf16 shadow of arena elements [lo, hi) on the current stream (every piece of the update is
        elementwise, so a step may be made range by range)"""
        from . import ops
        from .parallel import mean_scale
        mdl = self._model
        flat, gflat = mdl._flat, mdl._gflat
        g = self.param_groups[0]
        b1, b2 = self._betas(g["kind"], g["momentum"])
        gmul = None
        if world_size > 1:
            if self._gmul is None:
                self._gmul = torch.full((1,), mean_scale(world_size), dtype=torch.float32, device=flat.device)
            gmul = self._gmul
        t = self._t
        ops.optim_step(g["kind"], flat[lo:hi], gflat[lo:hi], m=self._m[lo:hi], v=self._v[lo:hi], gmul=gmul, lr=g["lr"], beta1=b1,
                       beta2=b2, eps=1e-8, weight_decay=g["weight_decay"], max_norm=0.0, clip_value=self.clip_value, step=t,
                       first_step=(t == 1), shadow=None if mdl._shadow is None else mdl._shadow[lo:hi], max_blocks=max_blocks)

    @torch.no_grad()
    def update_range_early(self, lo, hi, world_size=1, max_blocks=0, covers=None, group=None):
        """Called by SpacecraftPoseNet.loss_and_grads(..., optimizer=self) from inside the backward pass, on the stream that
        produced (and, data parallel, exchanged) the gradients of arena elements [lo, hi): this step's update of that range,
        issued while the rest of backward still runs.  step() then updates what is left.  Ranges must not overlap.
        covers=(a, b): the sharded exchange -- this rank updates [lo, hi) of the bucket [a, b) and the other ranks the rest, so
        step() must leave the whole bucket alone."""
        self._model._ensure_arena()
        self._state(self._model._flat)
        if not self._early:
            self._t += 1
        self._early.append(tuple(covers) if covers is not None else (lo, hi))
        if covers is not None:      # rank-sharded bucket: this rank's moments are current on [lo, hi) only (gather_sharded_state)
            self._shard_buckets[int(covers[0])] = (int(covers[0]), int(covers[1]), int(world_size), group)
        else:
            self._unshard_if_needed(lo, hi)
        if hi > lo:
            self._update(lo, hi, world_size, max_blocks)

    @torch.no_grad()
    def fused_fc_update(self, name, gT, xT, M):
        """Called by SpacecraftPoseNet.loss_and_grads(..., optimizer=self) on its update stream: the weight gradient of fully
        connected layer `name` and this step's update of that weight in one kernel (spb_fc_wgrad_update) -- the gradient never
        reaches HBM, p.grad of that weight is not written -- followed by the bias' share (its gradient is already in the arena)."""
        from . import ops
        mdl = self._model
        mdl._ensure_arena()
        self._state(mdl._flat)
        if not self._early:
            self._t += 1
        lay = getattr(mdl, name)
        N, K = lay.weight.shape
        o, n = mdl._offs[name + ".weight"]
        g = self.param_groups[0]
        b1, b2 = self._betas(g["kind"], g["momentum"])
        self._early.append((o, o + (n + 7) // 8 * 8))
        ops.fc_wgrad_update(gT, xT, M, g["kind"], mdl._flat[o:o + n].view(N, K), None, self._m[o:o + n], self._v[o:o + n], None, lr=g["lr"],
                            beta1=b1, beta2=b2, eps=1e-8, weight_decay=g["weight_decay"], clip_value=self.clip_value, step=self._t,
                            first_step=(self._t == 1), shadow=None if mdl._shadow is None else mdl._shadow[o:o + n])
        o, n = mdl._offs[name + ".bias"]
        self.update_range_early(o, o + (n + 7) // 8 * 8)

    # GradScaler defaults (torch.cuda.amp.GradScaler(): init_scale 2**16 lives in the model's AMP state)
    amp_growth, amp_backoff, amp_interval = 2.0, 0.5, 2000

    def _step_fp16(self, world_size, group):
        """fp16: unscale, inf / nan check, clip, update and GradScaler.update() of trainer.py:175-181, all on the device: the
        optimizer pass multiplies every gradient by 1/scale before the clamp and does nothing at all when the check found a
        non-finite value; the scale halves then, and doubles after amp_interval clean steps in a row."""
        from . import ops
        from . import _lib as L
        import ctypes as C
        mdl = self._model
        flat, gflat = mdl._flat, mdl._gflat
        self._t += 1                                  # host-side count of step() calls; the bias corrections use the device count
        if world_size > 1:
            mdl.finish_gradient_exchange(group)
        mdl.finish_early_updates()
        st = mdl.amp_state()
        g = self.param_groups[0]
        b1, b2 = self._betas(g["kind"], g["momentum"])
        lib = L.lib_f16()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # inf / nan check.  Fast path (what loss_and_grads ran for this batch): the convolution range of the f32 gradient arena (9 MB) and
        # the six 16-bit [F][MP] gradient operands of the fully connected weight gradients -- dW = g^T x accumulates at most 64 products
        # in f32, so it is finite exactly when its operands are (the bias gradients are column sums of the same g) -- instead of reading
        # the 600 MB those layers' gradients occupy.  Exchanged gradients (world_size > 1) and the generic path: the whole arena.
        ops16 = mdl.fp16_check_operands() if world_size == 1 else None
        sg = L.AmpSegs()
        segs = [(gflat[:mdl._conv_end], 0)] + [(t_, 1) for t_ in ops16] if ops16 else [(gflat, 0)]
        if gflat.numel() & 3:
            segs = None
        if segs is not None and len(segs) <= L.AMP_SEGS:      # one launch: the checks and GradScaler's decision (last workgroup)
            for i, (t_, h) in enumerate(segs):
                sg.ptr[i], sg.n[i], sg.is16[i] = t_.data_ptr(), t_.numel(), h
            sg.nseg = len(segs)
            L.check(lib.spb_amp_decide(C.byref(sg), C.c_void_p(st.data_ptr()), float(g["lr"]), float(b1), float(b2), float(self.amp_growth),
                                       float(self.amp_backoff), int(self.amp_interval), stream), "spb_amp_decide")
        else:
            L.check(lib.spb_amp_check(C.c_void_p(gflat.data_ptr()), gflat.numel(), C.c_void_p(st.data_ptr()), stream), "spb_amp_check")
            L.check(lib.spb_amp_step(C.c_void_p(st.data_ptr()), float(g["lr"]), float(b1), float(b2), float(self.amp_growth),
                                     float(self.amp_backoff), int(self.amp_interval), stream), "spb_amp_step")
        gm = st[L.AMP_INV_SCALE:L.AMP_INV_SCALE + 1]
        if world_size > 1:                            # data-parallel mean on top of the unscale factor
            gm = gm / float(world_size)
        hyper, skip = st[L.AMP_LR:L.AMP_LR + 3], st[L.AMP_SKIP:L.AMP_SKIP + 1]

        def upd(lo, hi):
            ops.optim_step(g["kind"], flat[lo:hi], gflat[lo:hi], m=self._m[lo:hi], v=self._v[lo:hi], gmul=gm, lr=g["lr"], beta1=b1, beta2=b2,
                           eps=1e-8, weight_decay=g["weight_decay"], max_norm=0.0, clip_value=self.clip_value, step=max(self._t, 1),
                           first_step=False, hyper=hyper, shadow=mdl._shadow[lo:hi], skip=skip)
        # GradScaler.step() is all or nothing, so nothing can be updated before the check above.  But the 98 % of the arena that belongs to
        # the two heads (an 0.8 ms HBM-bound pass) need not hold up the launch stream: the next forward reads only convolution parameters
        # until pool5 and waits for the heads' update before fc6 (join_updates), exactly as in bf16 mode.
        ce = mdl._conv_end
        upd(0, ce)
        if world_size == 1 and self.overlap_heads_update:
            mdl.update_heads_on_side_stream(lambda: upd(ce, flat.numel()))
        else:
            upd(ce, flat.numel())
        mdl.optimizer_updated()

    overlap_heads_update = True

    @torch.no_grad()
    def step(self, closure=None, world_size=1, group=None):
        mdl = self._model
        mdl._ensure_arena()
        flat = mdl._flat          # not flat_parameters(): that would wait for the heads' update still in flight
        self._state(flat)
        if getattr(mdl, "precision", None) == "fp16":
            return self._step_fp16(world_size, group)
        early, self._early = sorted(self._early), []
        if not early:
            self._t += 1
        if world_size > 1:
            mdl.finish_gradient_exchange(group)     # the fc buckets were started from inside backward (loss_and_grads)
        mdl.finish_early_updates()                  # the launch stream waits for the updates made beside backward
        pos = 0
        for lo, hi in early + [(flat.numel(), flat.numel())]:
            if lo > pos:
                self._unshard_if_needed(pos, lo)
                self._update(pos, lo, world_size)
            pos = max(pos, hi)
        mdl.optimizer_updated()
        return None

    def state_dict(self):
        """COLLECTIVE after rank-sharded steps (gather_sharded_state): every rank calls it, rank 0 writes the file."""
        self._model.join_updates()      # the heads' moments may still be written on the update stream
        self.gather_sharded_state()
        sd = super().state_dict()
        sd["spn_fused"] = {"t": self._t, "m": None if self._m is None else self._m.detach().cpu(),
                           "v": None if self._v is None else self._v.detach().cpu()}
        # fp16: the GradScaler state lives on the device (loss scale, growth tracker, and the count of steps that were really
        # taken -- the Adam bias corrections use it, skipped steps do not advance it): torch keeps these in scaler.state_dict()
        # and in the optimizer's per-parameter `step`; here they travel with the optimizer
        pending, amp = getattr(self._model, "_amp_pending", None), getattr(self._model, "_amp", None)
        if pending is not None:
            sd["spn_fused"]["amp"] = pending.clone()
        elif amp is not None:
            sd["spn_fused"]["amp"] = amp.detach().float().cpu()
        return sd

    def load_state_dict(self, sd):
        """train.py keeps the reference's order (get_optimizer, load_checkpoint, model.to(device)), so the model may still be
        on the CPU here: the moments stay where the checkpoint put them and step() moves them next to the arena."""
        sd = dict(sd)
        st = sd.pop("spn_fused", None)
        super().load_state_dict(sd)
        if st is not None:
            self._t = int(st["t"])
            self._m = None if st["m"] is None else st["m"].detach().float().reshape(-1)
            self._v = None if st["v"] is None else st["v"].detach().float().reshape(-1)
            amp = st.get("amp")
            if amp is None and getattr(self._model, "precision", None) == "fp16" and self._t > 0:
                # a checkpoint written before the scaler state was saved: at least seed the device step count (bias corrections)
                from . import _lib as L
                amp = torch.zeros(L.AMP_STATE, dtype=torch.float32)
                amp[L.AMP_SCALE], amp[L.AMP_INV_SCALE], amp[L.AMP_STEPS] = 65536.0, 1.0 / 65536.0, float(self._t)
            if amp is not None and hasattr(self._model, "restore_amp_state"):
                # resume: loss scale, growth tracker and the count of steps actually taken go back to the device BEFORE the next
                # forward scales its loss gradient with them
                self._model.restore_amp_state(amp.detach().float().reshape(-1).clone())
