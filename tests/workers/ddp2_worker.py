"""Two data-parallel ranks on ONE MI355X (gloo collectives on device tensors), launched by tests/test_ddp2_gpu.py with
`python -m torch.distributed.run --nproc-per-node 2`.  Prints one JSON line from rank 0.

KRN: FusedTrainStep with the bucketed exchange overlapped with backward (step.py) and without; SPN: loss_and_grads +
SpnOptimizer.step with the fully connected bucket exchanged beside the trunk's backward (nets/spn.py).  Each rank trains on
its own shard.  Reported: largest parameter difference between the replicas after the steps, and the exchanged (summed)
gradient of the first step against the sum of the two ranks' local gradients computed without any exchange."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def gather(t):
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return out


def krn(rank, world, dev):
    from oracle import krn_oracle as O
    from speedplusbaseline_amd.engine import KrnEngine
    from speedplusbaseline_amd.step import FusedTrainStep
    B, res = 8, {}
    g = torch.Generator().manual_seed(3 + rank)
    x = torch.rand(B, 3, 224, 224, generator=g).to(dev); y = torch.rand(B, 2, 11, generator=g).to(dev)

    def fresh():
        eng = KrnEngine(11).attach(dev, "fp32")
        sd = O.init_state(11)
        for info in eng.param_infos:
            eng.param_view(info).copy_(sd[info[0]].to(dev))
        for name, shape, off, numel in eng.buffer_infos:
            eng.buffers[off: off + numel].copy_(sd[name].flatten().to(dev))
        return eng
    # local gradient of this rank's shard, no exchange
    eng = fresh()
    eng.forward(x, y, training=True); eng.grads.zero_(); eng.backward(B)
    torch.cuda.synchronize()
    want = sum(gather(eng.grads.clone()))
    for mode in ("0", "1"):
        os.environ["SPB_DDP_OVERLAP"] = mode
        eng = fresh()
        ts = FusedTrainStep(eng, B, kind="sgd", lr=0.05, momentum=0.9, weight_decay=1e-4, max_norm=1.0, dist_group=dist.group.WORLD,
                            world_size=world)
        ts(x, y)
        torch.cuda.synchronize()
        got = eng.grads.clone()                      # the summed gradient the clip and the update saw
        for _ in range(2):
            ts(x, y)
        torch.cuda.synchronize()
        both = gather(eng.params.clone())
        sp = eng.bucket_split()
        res["overlap" + mode] = dict(active=bool(ts._overlap), replica_diff=float((both[0] - both[1]).abs().max()),
                                     grad_rel_shallow=float((got[:sp] - want[:sp]).norm() / want[:sp].norm()),
                                     grad_rel_deep=float((got[sp:] - want[sp:]).norm() / want[sp:].norm()),
                                     moved=float((both[0] - O_flat(eng, O)).abs().max()))
    return res


def O_flat(eng, O):
    sd = O.init_state(11)
    out = torch.zeros_like(eng.params)
    for info in eng.param_infos:
        eng.param_view(info, out).copy_(sd[info[0]].to(out.device))
    return out


def spn(rank, world, dev):
    from oracle import spn_oracle as S
    from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet
    from speedplusbaseline_amd.optim import SpnOptimizer
    NC, res = 64, {}
    init = S.init_state(NC)
    x, yc, yw = (t.to(dev) for t in S.synth_batch(4, NC, seed=11 + rank))
    masks = {k: v.to(dev) for k, v in S.synth_masks(4, seed=5 + rank).items()}

    def fresh():
        net = SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False, precision="bf16")
        net.load_state_dict(init, strict=True)
        return net.to(dev).train()
    net = fresh()
    net.loss_and_grads(x, yc, yw, masks=masks)
    torch.cuda.synchronize()
    want = sum(gather(net.flat_grads().clone()))
    for mode in ("plain", "overlap", "overlap_f32", "overlap_early", "sharded_f32", "sharded"):
        net = fresh()
        opt = SpnOptimizer(list(net.parameters()), kind="sgd", lr=0.05, momentum=0.9, weight_decay=1e-4, model=net)
        p0 = net.flat_parameters().clone()
        got = None
        for it in range(2):
            if mode == "plain":
                net.loss_and_grads(x, yc, yw, masks=masks)
            elif mode == "overlap_early":      # the heads' buckets are updated on the communication stream as they arrive
                net.loss_and_grads(x, yc, yw, masks=masks, world_size=world, group=dist.group.WORLD, compress_bf16=False,
                                   optimizer=opt)
            elif mode in ("sharded_f32", "sharded"):   # reduce-scatter, rank-sharded update, all-gather of the shadows
                net.loss_and_grads(x, yc, yw, masks=masks, world_size=world, group=dist.group.WORLD,
                                   compress_bf16=(mode == "sharded"), optimizer=opt, sharded=True)
            else:
                net.loss_and_grads(x, yc, yw, masks=masks, world_size=world, group=dist.group.WORLD,
                                   compress_bf16=None if mode == "overlap" else False)
            if it == 0:
                net.finish_gradient_exchange(dist.group.WORLD)
                torch.cuda.synchronize()
                got = net.flat_grads().clone()
                if mode.startswith("sharded"):     # a rank holds the reduced fc gradient of its own slices only: compare those
                    got[net._conv_end:] = want[net._conv_end:]
            opt.step(world_size=world, group=dist.group.WORLD)
            if it == 0:
                torch.cuda.synchronize()
                after1 = net.flat_parameters().clone()
                if mode in ("overlap_early", "sharded_f32", "sharded"):
                    # optimizer.state_dict() is a collective after sharded steps: the momentum buffers it returns must be
                    # complete and identical on both ranks, and equal the unsharded run's (compared after the FIRST step, like the
                    # parameters: later steps amplify the atomics noise of the convolution gradients)
                    mom1 = opt.state_dict()["spn_fused"]["m"].to(dev)
        torch.cuda.synchronize()
        both = gather(net.flat_parameters().clone())
        ce = net._conv_end
        if mode == "overlap_f32":
            f32_params = after1
        res[mode] = dict(replica_diff=float((both[0] - both[1]).abs().max()),
                         grad_rel_conv=float((got[:ce] - want[:ce]).norm() / want[:ce].norm()),
                         grad_rel_fc=float((got[ce:] - want[ce:]).norm() / want[ce:].norm()),
                         moved=float((both[0] - p0).abs().max()))
        if mode in ("overlap_early", "sharded_f32", "sharded"):   # same exchange arithmetic, same update, compared after the first step (later
            d = (after1 - f32_params).abs()    # steps amplify the atomics noise of the convolution gradients, see test_spn_gpu)
            res[mode]["diff_conv"], res[mode]["diff_fc"] = float(d[:ce].max()), float(d[ce:].max())
        if mode.startswith("sharded"):         # the bf16 shadows the next forward reads: identical on both ranks after the all-gather
            sh = gather(net._shadow.float().clone())
            res[mode]["shadow_diff"] = float((sh[0] - sh[1]).abs().max())
        if mode in ("overlap_early", "sharded_f32", "sharded"):
            moms = gather(mom1)
            if mode == "overlap_early":
                mom_ref = mom1
            res[mode]["mom_replica_diff"] = float((moms[0] - moms[1]).abs().max())
            res[mode]["mom_rel"] = float((mom1[ce:] - mom_ref[ce:]).norm() / mom_ref[ce:].norm())
            res[mode]["mom_zero_frac"] = float((mom1[ce:] == 0).float().mean())
            mom2 = gather(opt.state_dict()["spn_fused"]["m"].to(dev))       # and again after a sharded step that followed a gather
            res[mode]["mom2_replica_diff"] = float((mom2[0] - mom2[1]).abs().max())
    return res


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    out = {"krn": krn, "spn": spn}[sys.argv[1]](rank, world, dev)
    if rank == 0:
        print("DDP2 " + json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
