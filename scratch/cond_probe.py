"""the conditioning run of tests/test_parity_conditioned_gpu.py, N times: prints the loss history (not a test)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(min(32, os.cpu_count() or 1))
import tests.test_parity_conditioned_gpu as T
import pytest
dev = torch.device("cuda:0")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    try:
        T._condition(dev)
        print("run %d: settled" % i)
    except BaseException as e:   # pytest.fail raises Failed (BaseException subclass)
        print("run %d: %s" % (i, str(e)[:700]))
