timeout 900 python -m pytest tests/test_ghiasi_gpu.py -q -x 2>&1 | grep -E "^E  |passed|failed" | head
for v in 1 0 1 0; do SPB_GCONV_UP2=$v timeout 300 python scratch/bench_ghiasi.py 2>&1 | grep -E "Ghiasi forward|u2" ; done
