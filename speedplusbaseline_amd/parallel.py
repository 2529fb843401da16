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
