run() { printf "%-60s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run ""
for v in 256 320 448 512; do run "spb_debug_set_wgrad_target:$v"; done
for v in 6 10; do run "spb_debug_set_wgrad_batch:$v"; done
for v in 14 28 56; do run "spb_debug_set_dw_split:$v"; done
run "spb_debug_set_gemm_wg_cap:512"
run "spb_debug_set_gemm_wg_cap:1024"
run "spb_debug_set_pwb:16,-1,4"
run "spb_debug_set_pwb:16,-1,6"
run "spb_debug_set_dw_plane_max_w:28"
run "spb_debug_set_gemm_sk:1,0,0;spb_debug_set_gemm_big:1,320,512"
run ""
