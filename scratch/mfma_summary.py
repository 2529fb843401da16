"""Matrix-core utilisation per kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16,
SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE; kernel-trace only).  MI355X_MICROARCH.md: MFMA_BUSY counts cycles, summed over the chip's
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); MOPS counts
512-FLOP units (bf16), so TFLOP/s = MOPS * 512 / time -- time from the wall clock of the launch in the same trace."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
        agg[k]["ns"] += int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)
print("%-72s %6s %10s %12s %10s %10s" % ("kernel", "calls", "us/call", "mfma busy", "TFLOP/s", "of 2500"))
for k in sorted(agg, key=lambda k: -agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"]):
    a, n = agg[k], max(cnt[k], 1)
    if a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] <= 0:
        continue
    gui = a["GRBM_GUI_ACTIVE"] / 8.0
    busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if gui > 0 else float("nan")
    us = a["ns"] / n / 1e3
    tf = a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (a["ns"] * 1e-9) / 1e12 if a["ns"] > 0 else float("nan")
    print("%-72s %6d %10.1f %11.1f%% %10.1f %9.1f%%" % (k, n, us, 100 * busy, tf, 100 * tf / 2500))
