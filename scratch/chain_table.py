"""Per-launch critical-path table of one KRN train step from a rocprofv3 kernel trace (csv).

usage: chain_table.py kernel_trace.csv [first-kernel-marker]  > profiles/rN_krn_chain.txt
Columns: index, start offset (us), duration (us), gap to the previous kernel END on the same queue (us), queue, what the
launch waited on (same-queue predecessor unless another queue's kernel ended later and before this start), grid (workgroups),
LDS bytes, VGPRs, kernel.
"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'stem_fwd'
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
# bench.py ends with 5 instrumented steps (plan profiler on: everything fused on one stream): take the last product step before them
back = int(sys.argv[3]) if len(sys.argv) > 3 else 7
s, e = idx[-back], idx[-back + 1]
# a step starts with the memset / weight_prep before the stem: walk back to the previous optimizer launch
while s > 0 and 'optim_step' not in rows[s - 1]['Kernel_Name']: s -= 1
while e > 0 and 'optim_step' not in rows[e - 1]['Kernel_Name']: e -= 1
t0 = int(rows[s]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n[:78]
qmain = collections.Counter(r['Queue_Id'] for r in rows[s:e]).most_common(1)[0][0]
last_end = {}
print("# one KRN train step (bs=48 bf16): %d launches; main queue %s" % (e - s, qmain))
print("%4s %9s %8s %7s %5s %-6s %7s %6s %4s  %s" % ("i", "start_us", "dur_us", "gap_us", "queue", "waited", "wgs", "lds", "vgpr", "kernel"))
tot = collections.defaultdict(float); gaps = 0.0; nmain = 0; busy_main = 0.0
for i, r in enumerate(rows[s:e]):
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r['Queue_Id']
    prev = last_end.get(q)
    gap = (st - prev) / 1e3 if prev else 0.0
    # did another queue's kernel end between our predecessor's end and our start?  then we probably waited on it
    waited = 'prev'
    for q2, e2 in last_end.items():
        if q2 != q and prev and e2 > prev and e2 <= st: waited = 'q' + q2
    wg = int(r.get('Workgroup_Size_X', 1) or 1)
    wgs = int(r.get('Grid_Size_X', 0) or 0) // max(wg, 1) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) * max(int(r.get('Grid_Size_Z', 1) or 1), 1)
    d = (en - st) / 1e3
    print("%4d %9.1f %8.1f %7.1f %5s %-6s %7d %6s %4s  %s" % (i, (st - t0) / 1e3, d, gap, q, waited, wgs, r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), short(r['Kernel_Name'])))
    last_end[q] = en
    if q == qmain: nmain += 1; busy_main += d; gaps += max(gap, 0.0) if prev else 0.0
    tot[short(r['Kernel_Name'])[:40]] += d
span = (int(rows[e - 1]['End_Timestamp']) - t0) / 1e3
print("# span %.1f us; main-queue launches %d, busy %.1f us, gaps %.1f us" % (span, nmain, busy_main, gaps))
