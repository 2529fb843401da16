#!/bin/bash
# kernel trace of a few SPN steps (run through gpurun from the repo root); prints one step's timeline via scratch/trace_print.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/spn1
python $R/bench.py --model spn --steps 30 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -o st -- python $R/bench.py --model spn --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/spn1/err.txt
cp $(find /tmp/ps -name "*kernel_trace.csv" | head -1) $R/gpurun_out/spn1/kernel_trace.csv
