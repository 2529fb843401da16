# A/B: system-fenced (bit 1 of spb_debug_set_side_priority) vs device-scope events between the launch stream and the side stream
run() { printf "%-60s " "$1 $2"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2 3; do
run "spb_debug_set_side_priority:0"
run "spb_debug_set_side_priority:2"
done
run "spb_debug_set_side_priority:0" "--precision fp16"
run "spb_debug_set_side_priority:2" "--precision fp16"
run "spb_debug_set_side_priority:0" "--model dann"
run "spb_debug_set_side_priority:2" "--model dann"
