#!/bin/bash
# A/B of debug-knob settings on the GPU box: bash scratch/ab.sh OUT "name|SPB_DEBUG value" ...   (name "base" with an empty value = defaults)
# per arm: the bench line (50 steps) and the plan profiler's per-launch table
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
for arm in "$@"; do
  name=${arm%%|*}; val=${arm#*|}
  SPB_DEBUG="$val" python $R/bench.py --steps ${STEPS:-60} --warmup 15 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "import json,sys; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['ms_per_step'])"
  if [ -z "${NOTABLE:-}" ]; then SPB_DEBUG="$val" python $R/scratch/launch_table.py > $OUT/launches_$name.txt 2>&1; grep -E "^# (total|pw_bwd|dw_|pw_|stem)" $OUT/launches_$name.txt; fi
done
