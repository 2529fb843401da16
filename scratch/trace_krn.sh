#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/krn1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o st -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $R/gpurun_out/krn1/err.txt
cp $(find /tmp/pk -name "*kernel_trace.csv" | head -1) $R/gpurun_out/krn1/kernel_trace.csv
