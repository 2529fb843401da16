# is the step host-bound?  host enqueue time per step against the step time, product library; few steps = empty queues (the host's own pace)
for k in 3 10 100; do
python bench.py --bare --steps $k --warmup 20 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($k, d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"
done
