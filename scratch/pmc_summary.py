"""Summarise rocprofv3 passes into profiles/: per-kernel-family time (kernel trace) and HBM traffic (PMC FETCH_SIZE /
WRITE_SIZE, separate passes).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced stream (16 B/lane), so reads are doubled;
WRITE_SIZE is used as reported (uncalibrated)."""
import csv, json, re, sys, collections

def family(name):
    m = re.search(r"pw_gemm_kernel<[^>]*?, *(\d+), *(\d+)>", name)   # last two template ints: PRO, EPI
    if m:
        pro, epi = int(m.group(1)), int(m.group(2))
        return "pw_gemm_dgrad" if pro == 2 else ("pw_gemm_fwd" if epi == 1 else "pw_gemm_plain")
    m = re.search(r"pw_(?:sk|os|big)_kernel<(\d+), *(\d+)", name)    # split-K / one-shot / 128x128 kernels: PRO, EPI lead the template list
    if m:
        return "pw_gemm_dgrad" if int(m.group(1)) == 2 else "pw_gemm_fwd"
    m = re.search(r"dw[rp]_bwd_kernel<[^,]*, *\d+, *(true|false), *(true|false)(?:, *(true|false))?", name)
    if m:                                                             # row-unit: <T, ST, WG, EPI, DG>; plane: <T, ST, WG, EPI>
        return "dw_wgrad" if (m.group(3) == "false") else "dw_dgrad"
    if "dwp_fwd_kernel" in name or "dwt_fwd_kernel" in name:
        return "dw_fwd"
    if "dwt_dgrad" in name:
        return "dw_dgrad"
    m = re.search(r"pw_st_kernel<(\d+)", name)                        # streaming kernel of the large maps (gemm_st.hip): forward only (PRO 1 / 3),
    if m:                                                             # <PRO, NJ, KS, SO>: SO = true are the statistics-only passes of the virtual
        return "pw_stats" if re.search(r"pw_st_kernel<\d+, *\d+, *\d+, *true>", name) else "pw_gemm_fwd"   # expand convolutions (plan category pw_stats)
    m = re.search(r"pw_rs_kernel<(\d+)", name)                        # row-slab kernel: PRO leads the template list (1 / 3 forward, 2 input gradient)
    if m:
        return "pw_gemm_dgrad" if int(m.group(1)) == 2 else "pw_gemm_fwd"
    for pat, fam in (("pwb_kernel", "pw_bwd_fused"), ("dwr_fwd_kernel", "dw_fwd"), ("dwr_bwd_kernel", "dw_dgrad"),
                     ("stem_fwd_mfma", "stem_fwd"), ("stem_wgrad_mfma", "stem_wgrad"), ("pw_gemm_dma_kernel", "pw_gemm_dma")):
        if pat in name:
            return fam
    for k in ("pw_wgrad", "dw_fwd", "dw_dgrad", "dw_wgrad", "stem_fwd", "stem_wgrad", "head_fwd", "head_reduce", "head_bwd",
              "bn_apply", "bn_bwd_prep", "bn_running_update", "bn_param_grads", "bn_load_running", "weight_prep", "grad_sqnorm",
              "optim_step", "domain_tail", "bce_logits", "partial_reduce", "gconv_slab", "gconv_up2", "gconv_kernel", "conv9", "in_coef",
              "in_apply", "style_fc", "final_sigmoid"):
        if k in name:
            return k
    for k in ("fork_gate", "fork_set", "det_flush", "arena_zero", "arena_add", "amp_", "softce", "fillBuffer", "copyBuffer"):
        if k in name:
            return "misc:" + k
    bare = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    return "other:" + re.sub(r"[<(].*", "", bare)[:40]

def counters(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg

def main(stats_csv, fetch_csv, write_csv, out_json, steps_in_pmc):
    fetch, write = counters(fetch_csv, "FETCH_SIZE"), counters(write_csv, "WRITE_SIZE")
    out = {"_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over the launches of the family; see scratch/pmc_summary.py",
           "families": {}}
    for fam in sorted(set(fetch) | set(write)):
        nf, kbf = fetch.get(fam, [0, 0.0]); nw, kbw = write.get(fam, [0, 0.0])
        n = max(nf, nw, 1)
        out["families"][fam] = {"launches": n, "fetch_KB_raw_per_launch": round(kbf / max(nf, 1), 2), "write_KB_per_launch": round(kbw / max(nw, 1), 2),
                                "hbm_bytes_per_launch": round((2 * kbf / max(nf, 1) + kbw / max(nw, 1)) * 1024)}
    # whole-step HBM traffic: every launch of the PMC passes (steps_in_pmc = timed + warm-up steps of the profiled command; the one-off
    # launches of engine construction -- weight copies, arena fills -- are a few MB and ride along)
    tot = sum(2 * v[1] for v in fetch.values()) + sum(v[1] for v in write.values())
    # passes in the profiled command = optimizer launches (bench.py: warm-up + timed + its 5 instrumented steps), whatever the caller assumed
    if "optim_step" in fetch and fetch["optim_step"][0] > 0:
        steps_in_pmc = fetch["optim_step"][0]
    out["step_hbm_bytes"] = round(tot * 1024 / max(steps_in_pmc, 1))
    out["steps_in_pmc"] = steps_in_pmc
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["families"], indent=1)[:3000])

if __name__ == "__main__":
    main(*sys.argv[1:5], int(sys.argv[5]) if len(sys.argv) > 5 else 3)
