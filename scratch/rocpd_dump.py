"""Dump a rocprofv3 rocpd sqlite trace: per-kernel stats and (optionally) the dispatch sequence of one step."""
import sqlite3, sys, re, collections
db = sys.argv[1]
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]; ks = [x for x in t if 'kernel_symbol' in x][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.group_segment_size, s.arch_vgpr_count, s.sgpr_count from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void ', '', n)
    return n[:90]
st = collections.OrderedDict()
for r in rows:
    k = short(r[0]); d = (r[2]-r[1])/1e3
    a = st.setdefault(k, [0, 0.0, 1e18, 0.0, r[7], r[8]])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in st.values())
print("%-92s %6s %10s %9s %9s %9s %5s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "vgpr", "sgpr"))
for k, a in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print("%-92s %6d %10.1f %9.2f %9.2f %9.2f %5s %5s  %.1f%%" % (k, a[0], a[1], a[1]/a[0], a[2], a[3], a[4], a[5], 100*a[1]/tot))
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    print("\nlast %d dispatches:" % n)
    for r in rows[-n:]:
        print("%-80s %9.2f us grid %7d x %3d wg %4d lds %6d gap_prev" % (short(r[0])[:80], (r[2]-r[1])/1e3, r[3]//max(r[5],1), r[4], r[5], r[6]))
