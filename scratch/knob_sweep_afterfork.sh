# with forks at 0.24 us instead of 5-7 us: do the fork-related plan knobs still sit at their optimum?  (tuning build)
run() { printf "%-70s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "spb_debug_set_launch_events:1"
run "spb_debug_set_dw_split:0"
run "spb_debug_set_dw_split:28"
run "spb_debug_set_dw_split:56"
run "spb_debug_set_wgrad_min_flush:2"
run "spb_debug_set_wgrad_min_flush:4"
run "spb_debug_set_wgrad_batch:4"
run "spb_debug_set_wgrad_batch:16"
run "spb_debug_set_wgrad_batch:-4"
run "spb_debug_set_wgrad_batch:-16"
run "spb_debug_set_side_priority:1"
run "spb_debug_set_launch_events:1"
