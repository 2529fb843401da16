"""Style decoder: HBM traffic from two rocprofv3 PMC passes over scratch/bench_ghiasi.py (FETCH_SIZE, WRITE_SIZE; separate runs,
--kernel-trace only; corrections as scratch/pmc_summary.py: KB counters, FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported).
bench_ghiasi.py runs 3 + 10 + 2 = 15 restyles of 48 images: per-kernel bytes per launch and the HBM bytes of ONE restyle
(every decoder launch of it) = total / 15."""
import csv, json, re, sys, collections
RESTYLES = 15


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[(]", n)[0][:60]


def read(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out_json):
    fetch, write = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    fam, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        nf, kbf = fetch.get(k, [0, 0.0]); nw, kbw = write.get(k, [0, 0.0])
        total += (2 * kbf + kbw) * 1024
        fam[k] = {"launches": max(nf, nw), "hbm_bytes_per_launch": round((2 * kbf / max(nf, 1) + kbw / max(nw, 1)) * 1024)}
    out = {"_note": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; restyle_hbm_bytes = all kernels of the run / %d restyles (B=48, 224x224)" % RESTYLES,
           "restyle_hbm_bytes": round(total / RESTYLES), "families": fam}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    print("HBM bytes per restyle of 48 images: %.1f MB" % (total / RESTYLES / 1e6))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print("  %-60s x%4d  %8.2f MB/launch" % (k, v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
