set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "dw" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_krn_gpu.py -x -q 2>&1 | tail -5
for v in 0 32768 0 32768 200000; do SPB_DW_SPLIT=$v timeout 300 python bench.py --steps 200 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $v', d['value'], d['ms_per_step'])"; done
