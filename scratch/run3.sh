for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 300 python bench.py --model spn --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q=$q spn', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q=$q krn', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --styleaug --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q=$q styleaug', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --model dann --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Q=$q dann', d['value'], d['ms_per_step'])"
done
