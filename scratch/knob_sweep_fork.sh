# A/B: how the weight gradients are handed to the side stream (spb_debug_set_launch_events bits 3-4): 1 default = device word stored at the
# entry of the depthwise kernel, 9 = events (completion event on the preceding GEMM's packet), 17 = device word stored by a one-wave kernel
run() { printf "%-50s " "$1 $2"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2 3; do
run "spb_debug_set_launch_events:1"
run "spb_debug_set_launch_events:9"
run "spb_debug_set_launch_events:17"
done
run "spb_debug_set_launch_events:1" "--precision fp16"
run "spb_debug_set_launch_events:9" "--precision fp16"
run "spb_debug_set_launch_events:1" "--model dann"
run "spb_debug_set_launch_events:9" "--model dann"
run "spb_debug_set_launch_events:1" "--model dann --batch 16"
run "spb_debug_set_launch_events:9" "--model dann --batch 16"
