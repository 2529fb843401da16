#!/bin/bash
# per-launch kernel trace of a few KRN steps -> gpurun_out/$OUT/kernel_trace.csv (scratch tooling, not the product)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${OUT:-krn1}
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o st -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/$OUT/bench.json 2> $R/gpurun_out/$OUT/err.txt
cp $(find /tmp/pk -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$OUT/kernel_trace.csv
