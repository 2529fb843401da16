"""per-kernel averages of the SQ counter passes collected by scratch/pmc_sq.sh"""
import csv, sys, collections, re, glob, os
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in sorted(glob.glob(os.path.join(d, "pass*.csv"))):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)[:60] + " g" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 0])[1]):
    v = {c: agg[k][c][1] / max(agg[k][c][0], 1) for c in agg[k]}
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-75s n=%d" % (k, agg[k][next(iter(agg[k]))][0]))
    print("    waves %.0f  wave_cyc %.3g  busy %.3g | wait_any %.2f  wait_inst %.2f  active %.2f (valu %.2f lds %.2f vmem %.2f) | insts valu %.3g lds %.3g vmrd %.3g vmwr %.3g salu %.3g mfma %.3g | lds conflict %.3g / idx %.3g | gui %.3g" % (
        v.get("SQ_WAVES", 0), wc, v.get("SQ_BUSY_CYCLES", 0), v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc,
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_ACTIVE_INST_LDS", 0) / wc, v.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_SALU", 0),
        v.get("SQ_INSTS_MFMA", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("GRBM_GUI_ACTIVE", 0)))
