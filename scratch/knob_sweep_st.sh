run() { printf "%-50s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "spb_debug_set_gemm_st:0,0,0"
for w in 384 512 640 768 896 1024; do run "spb_debug_set_gemm_st:1,0,$w"; done
for w in 512 768; do run "spb_debug_set_gemm_st:1,0,$w"; done
