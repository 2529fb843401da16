#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the depthwise row kernels with SPB_DW_XCD=$1 (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq
  SPB_DW_XCD=$1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq -o q -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - "$c" $(find /tmp/pq -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
c, f = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'dwr_' not in n: continue
    key = 'dwr_bwd' if 'dwr_bwd' in n else 'dwr_fwd'
    agg[key][0] += 1; agg[key][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()):
    print("%s %s: %d launches, %.1f MB per launch (raw KB counter / 1024%s)" % (c, k, n, v / n / 1024 * (2 if c == 'FETCH_SIZE' else 1), ", x2" if c == 'FETCH_SIZE' else ""))
PY
done
