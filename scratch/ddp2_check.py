"""two ranks on ONE GPU (gloo collectives on device tensors): replicas must stay identical through the bucketed, overlapped
gradient exchange.  torchrun --nproc-per-node 2 scratch/ddp2_check.py"""
import os, sys, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
device = torch.device('cuda', 0)
dist.init_process_group("gloo")
B = 8
res = {}
for mode in ("0", "1"):
    os.environ["SPB_DDP_OVERLAP"] = mode
    g = torch.Generator().manual_seed(3 + rank)
    x = torch.rand(B, 3, 224, 224, generator=g).to(device); y = torch.rand(B, 2, 11, generator=g).to(device)
    eng = KrnEngine(11).attach(device, "fp32")
    sd = O.init_state(11)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(device))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(device))
    ts = FusedTrainStep(eng, B, kind="sgd", lr=0.05, momentum=0.9, weight_decay=1e-4, max_norm=1.0, dist_group=dist.group.WORLD, world_size=world)
    p0 = eng.params.clone()
    for _ in range(3):
        ts(x, y)
    torch.cuda.synchronize()
    mine = eng.params.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    res[mode] = (float((both[0] - both[1]).abs().max()), (mine - p0).clone(), ts._overlap, eng.grads.clone())
if rank == 0:
    for mode in ("0", "1"):
        print("overlap=%s (active %s): max |params rank0 - rank1| = %.3e" % (mode, res[mode][2], res[mode][0]))
    sp = eng.bucket_split()
    a, b = res["0"][3], res["1"][3]
    print("summed gradients, plain vs overlapped exchange: shallow rel %.3e deep rel %.3e" % (
        float((a[:sp] - b[:sp]).norm() / a[:sp].norm()), float((a[sp:] - b[sp:]).norm() / a[sp:].norm())))
dist.destroy_process_group()
