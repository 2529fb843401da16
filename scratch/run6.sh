timeout 900 python -m pytest tests/test_ghiasi_gpu.py -q -x 2>&1 | grep -E "^E  |passed|failed" | head
for v in 1 1; do timeout 300 python scratch/bench_ghiasi.py 2>&1 | grep -E "Ghiasi forward|9x9|conv9" ; done
