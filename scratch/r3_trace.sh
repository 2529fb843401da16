#!/bin/bash
# round 3: baseline bench + per-launch kernel trace of the KRN step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${OUT:-r3a}
mkdir -p $R/gpurun_out/$OUT
cd $R
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/$OUT/bench_plain.json 2> gpurun_out/$OUT/bench_plain.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o st -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $R/gpurun_out/$OUT/bench.json 2> $R/gpurun_out/$OUT/err.txt
cp $(find /tmp/pk -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$OUT/kernel_trace.csv
python $R/scratch/chain_table.py $R/gpurun_out/$OUT/kernel_trace.csv > $R/gpurun_out/$OUT/chain.txt 2>&1
tail -3 $R/gpurun_out/$OUT/chain.txt; cat $R/gpurun_out/$OUT/bench_plain.json | cut -c1-300
