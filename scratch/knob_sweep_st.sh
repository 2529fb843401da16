run() { printf "%-60s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "spb_debug_set_gemm_st:3,0,0"
run "spb_debug_set_gemm_st:3,0,0;spb_debug_set_gemm_sk:5,0,0"
run "spb_debug_set_gemm_st:3,0,0;spb_debug_set_gemm_sk:3,0,0"
run "spb_debug_set_gemm_st:3,0,0;spb_debug_set_gemm_sk:1,128,0"
run "spb_debug_set_gemm_st:3,0,0;spb_debug_set_gemm_sk:1,0,1"
