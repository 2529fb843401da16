#!/bin/bash
# Collect the committed profiles of a round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the bench command, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE), each with
#   --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o st -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
cp $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $OUT/fetch.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
cp $(find /tmp/p_write -name "*counter_collection.csv" | head -1) $OUT/write.csv
python $ROOT/scratch/pmc_summary.py $OUT/kernel_stats.csv $OUT/fetch.csv $OUT/write.csv $OUT/pmc_traffic.json 4 > $OUT/pmc_summary.txt 2>&1
ls -la $OUT
