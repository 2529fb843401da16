"""KRN in float16 with dynamic loss scaling: the reference's own mixed-precision recipe for the keypoint network -- torch.cuda.amp.autocast +
GradScaler around forward / backward / clip / step (train.py:101-104, trainer.py:73-98) -- on the IEEE-half build of the KRN kernels
(libspb_hip_f16.so: the same sources compiled with -DSPB_F16, csrc/common.h) with GradScaler's arithmetic on the device
(FusedTrainStep._update_fp16: scaler.unscale_ + inf / nan check, clip_grad_norm_ on the unscaled gradient, optimizer.step() or nothing,
scaler.update()).  bfloat16 stays the benchmarked substitution (BASELINE configs[1]); this is the recipe a user of the reference's
--use_fp16 gets with --precision fp16.  Gradient fidelity on trained states: tests/test_parity_conditioned_gpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import krn_oracle as O  # noqa: E402
from speedplusbaseline_amd import _lib as L  # noqa: E402
from speedplusbaseline_amd.engine import KrnEngine  # noqa: E402
from speedplusbaseline_amd.step import FusedTrainStep  # noqa: E402

K, B = 11, 8


def _load(eng, sd):
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].detach().to(device=eng.device, dtype=torch.float32))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].detach().flatten().to(device=eng.device, dtype=torch.float32))


def test_fp16_forward_and_step_track_the_f32_path(device):
    """random init, bs=8: loss of the float16 pass against float32 HIP (and the float64 oracle's loss); one AdamW step of both: the parameter
    update of the float16 path is closer to the float32 one than the bfloat16 path's (random init amplifies rounding ~350x, tests/test_krn_gpu.py)"""
    sd = O.init_state(K)
    x, y = O.synth_batch(B, K, seed=7)
    out = {}
    for prec in ("fp32", "fp16", "bf16"):
        eng = KrnEngine(K).attach(device, prec)
        _load(eng, sd)
        ts = FusedTrainStep(eng, B, kind="sgd", lr=0.05, momentum=0.0, weight_decay=0.0, max_norm=1.0)
        p0 = eng.params.clone()
        for attempt in range(20):       # float16 at random init: the first scale (65536) overflows, GradScaler skips and halves until a step is taken
            s = ts(x.to(device), y.to(device))
            torch.cuda.synchronize()
            if prec != "fp16" or float(eng.amp[L.AMP_STEPS]) == 1.0:
                break
        out[prec] = (float(s[0]), (eng.params - p0).double().cpu(), eng, attempt)
    ref = float(O.krn_forward({k: v.clone() for k, v in sd.items()}, x, y, training=True)[0])
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    c16, cbf = cos(out["fp16"][1], out["fp32"][1]), cos(out["bf16"][1], out["fp32"][1])
    print("loss: oracle %.5f  fp32 %.5f  fp16 %.5f  bf16 %.5f;  clipped SGD update vs fp32: fp16 cosine %.4f (norm ratio %.4f), bf16 cosine %.4f"
          % (ref, out["fp32"][0], out["fp16"][0], out["bf16"][0], c16, float(out["fp16"][1].norm() / out["fp32"][1].norm()), cbf))
    assert abs(out["fp32"][0] - ref) <= 1e-3 * ref
    assert abs(out["fp16"][0] - ref) <= 0.08 * ref                     # (measured 1-5 % run to run; bf16 at random init: 5-15 %)
    # random init amplifies any rounding ~350x (tests/test_krn_gpu.py): float16's 2^-12 becomes ~0.7 in the update's cosine, bfloat16's 2^-9 ~0.3
    assert c16 > 0.55 and c16 > cbf
    amp = out["fp16"][2].amp.cpu()
    print("float16: %d skipped steps before the first one was taken, loss scale now %g" % (out["fp16"][3], float(amp[L.AMP_SCALE])))
    assert float(amp[L.AMP_SCALE]) == 65536.0 * 0.5 ** out["fp16"][3] and float(amp[L.AMP_STEPS]) == 1.0 and float(amp[L.AMP_SKIP]) == 0.0


def test_fp16_overflow_skips_the_step_and_halves_the_scale_then_recovers(device):
    """GradScaler semantics on the device: with the loss scale forced to 2^40 the float16 backward overflows -> found_inf -> the optimizer
    step is skipped ENTIRELY (parameters, moments, step count untouched), the scale halves; after enough halvings a step is taken."""
    sd = O.init_state(K)
    x, y = O.synth_batch(B, K, seed=9)
    eng = KrnEngine(K).attach(device, "fp16")
    _load(eng, sd)
    ts = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    eng.amp[L.AMP_SCALE] = 2.0 ** 40
    eng.amp[L.AMP_INV_SCALE] = 2.0 ** -40
    p0, m0 = eng.params.clone(), ts.m.clone()
    skipped = 0
    for it in range(40):
        ts(x.to(device), y.to(device))
        torch.cuda.synchronize()
        amp = eng.amp.cpu()
        if float(amp[L.AMP_STEPS]) == 0.0:
            skipped += 1
            assert torch.equal(eng.params, p0) and torch.equal(ts.m, m0)          # nothing moved
            assert float(amp[L.AMP_SCALE]) == 2.0 ** (40 - skipped)
        else:
            break
    assert 1 <= skipped < 40
    assert float(eng.amp[L.AMP_STEPS]) == 1.0 and not torch.equal(eng.params, p0)
    assert torch.isfinite(eng.params).all() and torch.isfinite(ts.m).all()
    print("skipped %d steps, scale now 2^%d" % (skipped, int(math.log2(float(eng.amp[L.AMP_SCALE])))))


def test_fp16_scale_grows_after_the_interval(device):
    sd = O.init_state(K)
    x, y = O.synth_batch(B, K, seed=11)
    eng = KrnEngine(K).attach(device, "fp16")
    _load(eng, sd)
    ts = FusedTrainStep(eng, B, kind="adamw", lr=1e-4, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    ts.amp_interval = 3
    eng.amp[L.AMP_SCALE] = 1024.0
    eng.amp[L.AMP_INV_SCALE] = 1.0 / 1024.0
    for it in range(3):
        ts(x.to(device), y.to(device))
    torch.cuda.synchronize()
    assert float(eng.amp[L.AMP_SCALE]) == 2048.0 and float(eng.amp[L.AMP_STEPS]) == 3.0
