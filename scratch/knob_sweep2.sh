run() { printf "%-70s " "$1"; SPB_DEBUG="$1" python bench.py --bare --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run ""
run "spb_debug_set_replica_rows:-2000"
run "spb_debug_set_replica_rows:-2000;spb_debug_set_replica_rows:-2"
run "spb_debug_set_replica_rows:-2000;spb_debug_set_replica_rows:-8"
run "spb_debug_set_replica_rows:-2"
run "spb_debug_set_replica_rows:-8"
run ""
