"""In-tree build of libspb_hip.so (hipcc, gfx950 only).

The shared object is git-ignored but travels with the repo snapshot to the GPU box; nothing is JIT-compiled at
run time and there is no non-HIP code path behind it.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libspb_hip.so")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["gemm_pw.hip", "gemm_sk.hip", "gemm_os.hip", "gemm_big.hip", "gemm_rs.hip", "gemm_st.hip", "pw_bwd_fused.hip", "dwconv_rows.hip", "dwconv_plane.hip", "dwconv_tile.hip", "stem_head.hip", "stem_mfma.hip", "elemwise.hip", "ghiasi.hip", "ghiasi_wide.hip", "ghiasi_f32.hip", "spn.hip", "spn_fc.hip", "spn_conv.hip", "preproc.hip", "krn_plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]
# IEEE-half twin for the SPN fp16 recipe (csrc/common.h, -DSPB_F16): the SPN kernels, the pointwise GEMMs they use and the
# elementwise / optimizer kernels.
LIB_F16 = os.path.join(HERE, "libspb_hip_f16.so")
# Round 5: + the KRN / DANN kernels and the plan -- the reference's own AMP recipe for KRN is float16 autocast + GradScaler
# (train.py:101-104, trainer.py:73-94); bfloat16 stays the benchmarked substitution (BASELINE configs[1]).
SOURCES_F16 = ["gemm_pw.hip", "elemwise.hip", "spn.hip", "spn_fc.hip", "spn_conv.hip",
               "gemm_sk.hip", "gemm_os.hip", "gemm_big.hip", "gemm_rs.hip", "gemm_st.hip", "pw_bwd_fused.hip", "dwconv_rows.hip", "dwconv_plane.hip", "dwconv_tile.hip",
               "stem_head.hip", "stem_mfma.hip", "krn_plan.hip",
               # round 6: the style decoder in IEEE half (Ghiasi(precision="fp16")): the reference runs this module in float32 (trainer.py:68-69);
               # half has eight times bfloat16's mantissa at the same matrix-core rate
               "ghiasi.hip", "ghiasi_wide.hip", "ghiasi_f32.hip"]
# Reproducible twin (csrc/common.h, -DSPB_DET): the KRN / DANN kernels and the plan with exact (order-independent) accumulation in
# place of float atomics.  KrnEngine(..., deterministic=True) and tests/test_parity_conditioned_gpu.py use it.
# Tuning twin (-DSPB_TUNING): the only build that exports the spb_debug_set_* knobs (include/spb_hip_tuning.h).  Measurement scripts and
# the kernel-variant tests load it; the product library above has no mutable tuning state.
LIB_TUNE = os.path.join(HERE, "libspb_hip_tune.so")
LIB_DET = os.path.join(HERE, "libspb_hip_det.so")
SOURCES_DET = ["gemm_pw.hip", "gemm_sk.hip", "gemm_os.hip", "gemm_big.hip", "gemm_rs.hip", "gemm_st.hip", "pw_bwd_fused.hip", "dwconv_rows.hip", "dwconv_plane.hip", "dwconv_tile.hip", "stem_head.hip", "stem_mfma.hip", "elemwise.hip", "krn_plan.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr((SOURCES, SOURCES_F16, SOURCES_DET)).encode())     # a source added to one of the libraries relinks it
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "spb_hip.h"))
    headers.append(os.path.join(ROOT, "include", "spb_hip_tuning.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    stamp = os.path.join(OBJDIR, "stamp.txt")
    want = _digest(srcs + headers)
    if not force and os.path.exists(LIB) and os.path.exists(LIB_F16) and os.path.exists(LIB_DET) and os.path.exists(LIB_TUNE) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        dig = os.path.join(OBJDIR, os.path.basename(src) + ".sha")
        d = _digest([src] + headers)
        if not force and os.path.exists(obj) and os.path.exists(dig) and open(dig).read().strip() == d:
            return obj
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(dig, "w") as f:
            f.write(d)
        return obj

    def compile_twin(src, tag, define):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".%s.o" % tag)
        dig = os.path.join(OBJDIR, os.path.basename(src) + ".%s.sha" % tag)
        d = _digest([src] + headers)
        if not force and os.path.exists(obj) and os.path.exists(dig) and open(dig).read().strip() == d:
            return obj
        cmd = [hipcc] + FLAGS + [define, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(dig, "w") as f:
            f.write(d)
        return obj

    srcs16 = [os.path.join(CSRC, s) for s in SOURCES_F16]
    srcsdet = [os.path.join(CSRC, s) for s in SOURCES_DET]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
        objs16 = list(ex.map(lambda s_: compile_twin(s_, "f16", "-DSPB_F16"), srcs16))
        objsdet = list(ex.map(lambda s_: compile_twin(s_, "det", "-DSPB_DET"), srcsdet))
        objstune = list(ex.map(lambda s_: compile_twin(s_, "tune", "-DSPB_TUNING"), srcs))
    for out, ob in ((LIB, objs), (LIB_F16, objs16), (LIB_DET, objsdet), (LIB_TUNE, objstune)):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + ob
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
