#!/bin/bash
# kernel trace of the KRN step under an env setting:  OUT=name ENVS="A=1 B=2" bash scratch/r3_trace2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${OUT:-r3t}
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
env $ENVS timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o st -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/$OUT/bench.json 2> $R/gpurun_out/$OUT/err.txt
cp $(find /tmp/pk -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$OUT/kernel_trace.csv
python $R/scratch/chain_table.py $R/gpurun_out/$OUT/kernel_trace.csv > $R/gpurun_out/$OUT/chain.txt 2>&1
tail -1 $R/gpurun_out/$OUT/chain.txt
