"""Kernel experiments: build libspb_hip.<name>.so with extra compiler flags (e.g. -DSPB_DW_RING_CAP=8); select it at run
time with SPB_LIB_VARIANT=<name>.   usage: python scratch/build_variant.py name [-Dflag ...]"""
import os, sys, subprocess
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speedplusbaseline_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
od = os.path.join("/tmp", "spb_var_" + name); os.makedirs(od, exist_ok=True)
def one(s):
    o = os.path.join(od, s + ".o")
    subprocess.check_call([B._hipcc()] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, s), "-o", o], stderr=subprocess.DEVNULL)
    return o
only = os.environ.get("ONLY")   # comma list of sources to recompile with the flags; the rest come from the product objects
srcs = B.SOURCES
with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(lambda s: one(s) if (not only or s in only.split(",")) else os.path.join(B.OBJDIR, s + ".o"), srcs))
out = os.path.join(B.HERE, "libspb_hip.%s.so" % name)
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
