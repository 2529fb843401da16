# A/B of ops.StreamFork in the SPN and DANN steps: device-word forks (default) against events (SPB_EVENT_FORKS=1)
run() { printf "%-36s %-24s " "$1" "$2"; env $1 python bench.py --bare --steps 60 --warmup 15 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
for a in "--model spn" "--model spn --precision fp16" "--model dann" "--model dann --batch 16"; do
run "X=1" "$a"
run "SPB_EVENT_FORKS=1" "$a"
done; done
