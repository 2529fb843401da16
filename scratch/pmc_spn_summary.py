"""SPN: HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only).
Same corrections as scratch/pmc_summary.py (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): counters in KB, FETCH_SIZE
doubled on gfx950 for wide coalesced streams, WRITE_SIZE as reported.  Launches are keyed by (kernel, grid size): the
arena-wide optimizer pass ("optim_step_full", the largest grid) is the SPN step's dominant kernel."""
import csv, json, re, sys, collections


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n)[0]


def read(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out_json, steps=0):
    fetch, write = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    fam = {}
    for key in sorted(set(fetch) | set(write)):
        nf, kbf = fetch.get(key, [0, 0.0]); nw, kbw = write.get(key, [0, 0.0])
        fam["%s@%d" % key] = {"launches": max(nf, nw), "fetch_KB_raw_per_launch": round(kbf / max(nf, 1), 2),
                              "write_KB_per_launch": round(kbw / max(nw, 1), 2),
                              "hbm_bytes_per_launch": round((2 * kbf / max(nf, 1) + kbw / max(nw, 1)) * 1024)}
    full = max((k for k in fam if k.startswith("optim_step_kernel@")), key=lambda k: int(k.split("@")[1]), default=None)
    if full:
        fam["optim_step_full"] = dict(fam[full], grid=int(full.split("@")[1]))
    out = {"_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, averaged over the launches with that grid; scratch/pmc_spn_summary.py",
           "families": fam}
    if int(steps) > 0:   # the passes ran `bench.py --model spn --bare --steps S --warmup W`: every launch belongs to one of the S + W steps
        tot = sum((2 * kbf + kbw) * 1024 for (kbf, kbw) in
                  ((fetch.get(k, [0, 0.0])[1], write.get(k, [0, 0.0])[1]) for k in set(fetch) | set(write)))
        out["steps"] = int(steps)
        out["step_hbm_bytes"] = round(tot / int(steps))
        print("whole step: %.1f MB of HBM traffic (all launches of %d steps / %d)" % (tot / int(steps) / 1e6, int(steps), int(steps)))
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in fam.items():
        if v["hbm_bytes_per_launch"] > 20e6 or k == "optim_step_full":
            print("%-44s n=%3d  %8.1f MB / launch" % (k, v["launches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
