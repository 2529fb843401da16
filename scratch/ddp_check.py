import os, socket, sys, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from oracle import krn_oracle as O
from speedplusbaseline_amd.engine import KrnEngine
from speedplusbaseline_amd.step import FusedTrainStep
device = torch.device('cuda', 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
B = 8
g = torch.Generator().manual_seed(3)
x = torch.rand(B, 3, 224, 224, generator=g).to(device); y = torch.rand(B, 2, 11, generator=g).to(device)
def run(mode, steps):
    os.environ["SPB_DDP_OVERLAP"] = mode
    eng = KrnEngine(11).attach(device, "fp32")
    sd = O.init_state(11)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(device))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(device))
    ts = FusedTrainStep(eng, B, kind="sgd", lr=0.05, momentum=0.9, weight_decay=1e-4, max_norm=1.0, dist_group=dist.group.WORLD, world_size=1)
    p0 = eng.params.clone()
    for _ in range(steps):
        sc = ts(x, y)
    torch.cuda.synchronize()
    return eng.params - p0, eng.grads.clone(), eng.bucket_split(), sc
for steps in (1, 2):
    a = run("0", steps); b = run("0", steps); c = run("force", steps)
    sp = a[2]
    for nm, u, v in (("plain vs plain", a, b), ("plain vs overlap", a, c)):
        print(steps, nm, "delta rel: shallow %.3e deep %.3e | grads rel: shallow %.3e deep %.3e" % (
            float((u[0][:sp]-v[0][:sp]).norm()/u[0][:sp].norm()), float((u[0][sp:]-v[0][sp:]).norm()/u[0][sp:].norm()),
            float((u[1][:sp]-v[1][:sp]).norm()/u[1][:sp].norm()), float((u[1][sp:]-v[1][sp:]).norm()/u[1][sp:].norm())), u[3].tolist())
dist.destroy_process_group()
