"""print the last step of a rocprofv3 kernel trace (csv): start offset, duration, kernel name, grid"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'im2col_rgb'
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
s = idx[-2] if len(idx) > 1 else idx[-1]
e = idx[-1] if len(idx) > 1 else len(rows)
t0 = int(rows[s]['Start_Timestamp'])
busy = 0.0
for r in rows[s:e]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    print("%8.1f %7.1f  %s  grid=%s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d,
          r['Kernel_Name'][:90].replace('(anonymous namespace)::', '').replace('void ', ''), r.get('Grid_Size_X', '')))
print("step span %.1f us, kernel-busy %.1f us, %d launches" % ((int(rows[e - 1]['End_Timestamp']) - t0) / 1e3 if e <= len(rows) else 0, busy, e - s))
