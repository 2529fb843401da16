# A/B: depthwise backward row-unit kernels with their loop-invariant coefficients in registers (-DSPB_DW_HOIST=1) against the product library
#   hoist  = input-gradient instances only (tap weights + coefficient vectors);  hoist2 = + the weight-gradient-only instance's coefficient vectors
run() { printf "%-28s " "$1"; env $1 python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2 3; do
run "X=1"
run "SPB_LIB_VARIANT=hoist"
run "SPB_LIB_VARIANT=hoist2"
done
