"""Static instruction mix of a kernel's main loop (the longest backward-branch span) from the compiler's assembly: how DESIGN section 5
("Round 5, second half") counts instructions per loop iteration of the depthwise row-unit backward kernels with and without the register hoist.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S [-DSPB_DW_HOIST=0] speedplusbaseline_amd/csrc/dwconv_rows.hip -o /tmp/dwr.s
  python scratch/loop_mix.py /tmp/dwr.s dwr_bwd_kernelItLi1ELb0ELb1ELb1E [more mangled-name fragments ...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
for frag in sys.argv[2:]:
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and frag in l.split(":")[0]][0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels, best = {}, (0, 0, 0)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        b = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if b and b.group(1) in labels and i - labels[b.group(1)] > best[0]:
            best = (i - labels[b.group(1)], labels[b.group(1)], i)
    c = collections.Counter()
    for l in body[best[1]:best[2] + 1]:
        l = l.strip()
        if not l or l[0] in ".;":
            continue
        op = l.split()[0]
        kind = ("vector" if op.startswith("v_") else "lds" if op.startswith("ds_") else "waitcnt" if op.startswith("s_waitcnt") else
                "scalar" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_")) else "other")
        c[kind] += 1
        if "dpp" in l:
            c["(of the vector: dpp)"] += 1
    print("%s: %d instructions per iteration %s" % (frag, sum(v for k, v in c.items() if not k.startswith("(")), dict(c)))
