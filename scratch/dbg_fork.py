import os, time, ctypes as C, torch
os.environ["SPB_FORK_TIMEOUT_S"] = "0.1"
from speedplusbaseline_amd import _lib as L, ops
dev = torch.device("cuda:0")
lib = L.lib()
h = C.c_void_p(); L.check(lib.spb_fork_create(C.byref(h)), "create")
print("selftest", lib.spb_fork_selftest())
a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
big = torch.zeros(256 << 20, device=dev)
torch.cuda.synchronize()
t0 = time.time()
with torch.cuda.stream(a):
    for _ in range(800):
        big.add_(1.0)
t1 = time.time()
code = lib.spb_fork_streams(h, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream))
with torch.cuda.stream(b):
    y = big[:4].clone()
b.synchronize(); t2 = time.time()
a.synchronize(); t3 = time.time()
print("enqueue %.3f s, b done after %.3f s, a done after %.3f s, fork code %d, y %s" % (t1 - t0, t2 - t0, t3 - t0, code, y.tolist()))
print("next fork code", lib.spb_fork_streams(h, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream)))
torch.cuda.synchronize()
