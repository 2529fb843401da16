#!/bin/bash
# A/B of the KRN step: env settings per line -> ms/step   (each argument: "VAR=val VAR2=val2")
cd ${GRAFT_REPO_ROOT:-.}
run() { echo -n "$* : "; env "$@" python bench.py --steps ${STEPS:-60} --warmup 15 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; }
for cfg in "$@"; do run $cfg; done
