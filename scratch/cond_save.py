"""one conditioning run of tests/test_parity_conditioned_gpu.py, saved as the probe state (not a test)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
import tests.test_parity_conditioned_gpu as T
dev = torch.device("cuda:0")
for i in range(4):
    try:
        sd = T._condition(dev)
        break
    except BaseException as e:
        print("retry:", str(e)[:200])
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "probe_state.pt")
torch.save({k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}, out)
print("saved", out)
