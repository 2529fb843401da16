"""Data-parallel helpers: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the MI355X node).

The reference has no distributed code; data parallelism is this build's addition: every rank runs the
full model on its own bs/GPU shard of the global batch; the flat f32 gradient arena (KRN: 22.6 MB, RevGrad: 24.2 MB) is summed
in TWO buckets: the tail of the arena (inverted-residual blocks 14..17, extras, head: 90 % of the elements) is final after the
7x7 part of the backward pass and is all-reduced on a communication stream while the backward of blocks 13..1 runs; the
small head of the arena follows after backward.  The 1/world mean is folded into the optimizer kernel's gradient multiplier.
BatchNorm statistics stay per-rank: the reference normalises over exactly the per-GPU batch, so this preserves its semantics
at bs=48/GPU.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    """[lo, hi) slice of a global batch owned by `rank` (equal shards; global_batch must divide evenly)"""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_slice(lo, hi, rank, world):
    """The sharded exchange of an arena bucket [lo, hi) (SpacecraftPoseNet._shard_exchange): `world` equal pieces of `per`
    elements (8-element aligned, the staging buffers are padded to world * per); returns (per, my_lo, my_hi) -- rank's slice
    [my_lo, my_hi), empty for trailing ranks of a tiny bucket."""
    n = hi - lo
    per = ((n + world - 1) // world + 7) // 8 * 8
    my_lo = min(hi, lo + rank * per)
    return per, my_lo, min(hi, my_lo + per)


def allreduce_sum_(flat, group=None):
    """in-place sum over ranks of a flat gradient arena; returns the tensor"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_sum_async(flat, group=None, force=False):
    """enqueue the in-place sum of `flat` over ranks (on the current stream's communicator); returns the work handle
    (call .wait() before the optimizer reads the arena) or None when there is nothing to do.  force: also with a single
    rank (exercises the stream plumbing on a one-GPU box)"""
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def mean_scale(world):
    """gradient multiplier that turns the all-reduced SUM into the data-parallel MEAN"""
    return 1.0 / float(world)


def broadcast_(tensors, src=0, group=None):
    """rank `src`'s copy of each tensor everywhere (identical initial replicas)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src, group=group)


def shared_coin(step, seed, p):
    """rank-synchronous Bernoulli(p) draw for per-batch decisions that change a rank's step time (style augmentation):
    derived from (seed, step) only, so all ranks take the same branch and nobody stalls the all-reduce"""
    g = torch.Generator().manual_seed(int(seed) * 1000003 + int(step))
    return bool(torch.rand(1, generator=g).item() < p)


# ------------------------------------------------------------------------------------------------ launcher side
# What train.py / adapt.py call when they are started as `python -m torch.distributed.run --nproc-per-node N train.py ...`
# (one process per GPU).  The reference has one process and `cuda:0` (train.py:50, adapt.py:48); a plain `python train.py`
# still takes exactly that path here.

class Job:
    """rank / world / device of this process plus the conventions the CLI scripts share"""

    def __init__(self, rank, world, device, group):
        self.rank, self.world, self.device, self.group = rank, world, device, group

    @property
    def is_main(self):          # the rank that writes checkpoints, config.txt, TensorBoard scalars and result files
        return self.rank == 0

    def seed(self, base):       # per-rank data stream (different samples on every rank), identical model seed
        return int(base) + 7919 * self.rank

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def close(self):
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def init_job(use_cuda=True):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (set by torch.distributed.run); with WORLD_SIZE > 1 binds this process to
    `cuda:LOCAL_RANK` and creates the process group -- backend "nccl" (= RCCL over xGMI on the MI355X node).  SPB_ONE_DEVICE=1 puts
    every rank on cuda:0 with gloo collectives (RCCL cannot host two ranks on one device): what the two-rank tests on a one-GPU
    box use.  SPB_DIST_BACKEND overrides the backend.  Without the launcher's variables: (rank 0, world 1, cuda:0), no group."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("SPB_ONE_DEVICE", "0") == "1"
    device = torch.device("cuda", 0 if (one_device or world == 1) else local) if use_cuda else torch.device("cpu")
    if world == 1:
        return Job(0, 1, device, None)
    if use_cuda:
        if not one_device and local >= torch.cuda.device_count():
            raise RuntimeError("LOCAL_RANK %d but only %d visible GPUs (one process per GPU)" % (local, torch.cuda.device_count()))
        torch.cuda.set_device(device)
    backend = os.environ.get("SPB_DIST_BACKEND") or ("gloo" if (one_device or not use_cuda) else "nccl")
    if not dist.is_initialized():
        # a generous collective timeout: rank 0 alone validates / writes checkpoints between epochs while the others wait in a barrier
        import datetime
        tmin = int(os.environ.get("SPB_DIST_TIMEOUT_MIN", "120"))
        # the library's stream-fork gates (csrc/elemwise.hip) may sit behind a launch stream that holds a collective waiting for a late
        # peer: they must not give up before the process group itself would (default 600 s, read once when the first gate is enqueued)
        os.environ.setdefault("SPB_FORK_TIMEOUT_S", str(60 * tmin + 60))
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=tmin))
    return Job(rank, world, device, dist.group.WORLD)


def sync_replicas(model, job, src=0):
    """identical replicas before the first step: rank `src`'s parameters and buffers everywhere.  Every rank builds the model
    from the same seed (set_all_seeds), so this only matters after a resume where rank 0 alone read the checkpoint, or when an
    initialisation drew from an unseeded source."""
    if job.world <= 1:
        return
    eng = model.__dict__.get("_engine") if hasattr(model, "__dict__") else None
    if eng is not None:                       # KRN / RevGrad: parameters and BatchNorm buffers are views into three flat arenas
        broadcast_([eng.params, eng.buffers, eng.nbt], src, job.group)
        return
    if callable(getattr(model, "flat_parameters", None)):     # SPN: one flat f32 arena (+ its 16-bit shadow, rebuilt from it)
        flat = model.flat_parameters()
        broadcast_([flat], src, job.group)
        if callable(getattr(model, "invalidate", None)):
            model.invalidate()                # compute copies and the 16-bit shadow are rebuilt from the arena
        return
    with torch.no_grad():
        broadcast_([p.data for p in model.parameters()] + [b.data for b in model.buffers()], src, job.group)


def replica_digest(model, device=None):
    """float64 (sum, sum of squares, count) over all parameters on `device`: cheap cross-rank equality check of the replicas
    (BatchNorm running statistics are left out: they are per-rank by design, the reference normalises over the per-GPU batch)"""
    with torch.no_grad():
        ps = [p.detach() for p in model.parameters()]
        dev = device if device is not None else (ps[0].device if ps else torch.device("cpu"))
        s = torch.zeros(3, dtype=torch.float64, device=dev)
        for q in ps:
            q = q.double()
            s[0] += q.sum().to(dev); s[1] += (q * q).sum().to(dev); s[2] += q.numel()
    return s


def check_replicas(model, job, what="parameters"):
    """COLLECTIVE: largest difference of the ranks' parameter digests (0.0 for identical replicas -- the summed gradient, the clip
    and the update are the same arithmetic on every rank).  Raises when the replicas have drifted apart; returns the difference."""
    if job.world <= 1:
        return 0.0
    mine = replica_digest(model, job.device)
    every = [torch.empty_like(mine) for _ in range(job.world)]
    dist.all_gather(every, mine, group=job.group)
    diff = max(float((e - every[0]).abs().max()) for e in every)
    if diff != 0.0:
        raise RuntimeError("data-parallel replicas differ (%s digest spread %.3e over %d ranks): a rank missed or doubled a "
                           "gradient exchange" % (what, diff, job.world))
    return diff
