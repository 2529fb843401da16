R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_spn_gpu.py tests/test_spn_fullsize_gpu.py -q -x 2>&1 | grep -E "^E|passed|failed" | head -20
for v in 1 1; do SPB_SPN_STEM=$v timeout 300 python bench.py --model spn --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stem $v', d['value'], d['ms_per_step'])"; done
mkdir -p $R/gpurun_out/spn1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o st -- python $R/bench.py --model spn --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/spn1/err.txt
cp $(find /tmp/ps -name "*kernel_trace.csv" | head -1) $R/gpurun_out/spn1/kernel_trace.csv
