for i in 1 2 3; do
  timeout 1200 python -m pytest tests/test_parity_conditioned_gpu.py tests/test_ddp2_gpu.py tests/test_spn_gpu.py tests/test_spn_fullsize_gpu.py -q -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -8
  echo "--- run $i"
done
