"""A/B of the optimizer launch shape on a 152 M element arena (run on the GPU box)"""
import torch, sys
sys.path.insert(0, '.')
from speedplusbaseline_amd import ops, _lib as L
n = 152372368 // 8 * 8
dev = 'cuda'
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
def run(tag, vec, blocks, nt, shadow=True):
    L.lib().spb_debug_set_optim(vec, blocks, nt)
    for _ in range(3):
        ops.optim_step("adamw", p, g, m=m, v=v, lr=1e-3, clip_value=1.0, step=2, shadow=sh if shadow else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.optim_step("adamw", p, g, m=m, v=v, lr=1e-3, clip_value=1.0, step=2, shadow=sh if shadow else None)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    by = n * (28 + (2 if shadow else 0))
    print("%-34s %7.1f us  %5.2f TB/s" % (tag, t * 1e3, by / t / 1e9))
for vec in (1, 4):
    for blocks in (1, 2, 4):
        for nt in ((0, 1) if vec == 4 else (0,)):
            run("vec%d blocks=%d nt=%d" % (vec, blocks, nt), vec, blocks, nt)
run("vec1 U1 no shadow", 1, 1, 0, shadow=False)
# plain copy bandwidth for reference
a = torch.empty(n, device=dev); 
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): a.copy_(p)
e1.record(); torch.cuda.synchronize()
print("torch copy: %.2f TB/s" % (n * 8 / (e0.elapsed_time(e1) / 10) / 1e9))
