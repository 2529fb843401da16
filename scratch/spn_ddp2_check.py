"""two ranks on ONE GPU (gloo): SPN data-parallel step with the fc bucket's all-reduce overlapped with the trunk backward.
torchrun --nproc-per-node 2 scratch/spn_ddp2_check.py"""
import os, sys, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from oracle import spn_oracle as S
from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet
from speedplusbaseline_amd.optim import SpnOptimizer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device('cuda', 0)
dist.init_process_group("gloo")
NC = 64
res = {}
for mode in ("plain", "plain2", "overlap", "overlap_bf16"):
    net = SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False, precision="bf16")
    net.load_state_dict(S.init_state(NC), strict=True)
    net = net.to(dev).train()
    opt = SpnOptimizer(list(net.parameters()), kind="sgd", lr=0.05, momentum=0.9, weight_decay=1e-4, model=net)
    x, yc, yw = S.synth_batch(4, NC, seed=11 + rank)
    masks = {k: v.to(dev) for k, v in S.synth_masks(4, seed=5 + rank).items()}
    x, yc, yw = x.to(dev), yc.to(dev), yw.to(dev)
    p0 = net.flat_parameters().clone()
    for _ in range(2):
        if mode.startswith("plain"):
            net.loss_and_grads(x, yc, yw, masks=masks)
        else:
            net.loss_and_grads(x, yc, yw, masks=masks, world_size=world, group=dist.group.WORLD, compress_bf16=(mode == "overlap_bf16"))
        opt.step(world_size=world, group=dist.group.WORLD)
    torch.cuda.synchronize()
    mine = net.flat_parameters().clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    res[mode] = (float((both[0] - both[1]).abs().max()), mine - p0)
if rank == 0:
    for m, (d, _) in res.items():
        print("%-13s max |params rank0 - rank1| = %.3e" % (m, d))
    a = res["plain"][1]
    for m in ("plain2", "overlap", "overlap_bf16"):
        b = res[m][1]
        print("delta vs plain (%s): rel L2 %.3e" % (m, float((a - b).norm() / a.norm())))
dist.destroy_process_group()
