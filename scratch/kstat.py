import sys, json
d = json.loads(sys.stdin.read()); k = d["kernels"]
print(d["value"], d["ms_per_step"], " ".join("%s=%.1f" % (n, k[n]["ms_per_step"] * 1e3) for n in sys.argv[1:] if n in k))
