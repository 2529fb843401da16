"""run pytest on the given arguments with the caching allocator's pool pre-filled with NaN (a kernel that relies on zero-initialised
workspace then shows up at once); not a test.   python scratch/polluted_run.py tests/test_parity_conditioned_gpu.py -q -m gpu -s"""
import sys, torch, pytest
junk = [torch.full((256 << 20,), float("nan"), device="cuda") for _ in range(8)]     # 8 GiB of NaN, returned to the pool
junk += [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(256)]
torch.cuda.synchronize()
del junk
sys.exit(pytest.main(sys.argv[1:]))
