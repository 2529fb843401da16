#!/bin/bash
# A/B sweep of tuning knobs on the KRN bench (tuning build through SPB_DEBUG): prints ms/step per setting.
run() { printf "%-60s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run ""
run "spb_debug_set_dw_split:112"
for v in 0 14 28 56; do run "spb_debug_set_dw_split:$v"; done
for v in 30000 9000; do run "spb_debug_set_fused_pw_bwd:$v"; done
for v in 256 512 768; do run "spb_debug_set_wgrad_target:$v"; done
for v in 512 1024 1536; do run "spb_debug_set_gemm_wg_cap:$v"; done
for v in 4 12 16; do run "spb_debug_set_wgrad_batch:$v"; done
run "spb_debug_set_side_wgrad:0"
run "spb_debug_set_dw_tile:28,65536"
run "spb_debug_set_dw_plane_max_w:28"
run "spb_debug_set_pwb:32,-1,8"
run "spb_debug_set_pwb:16,0,8"
