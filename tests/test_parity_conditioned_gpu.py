"""bf16 -- the benchmarked compute type -- against the float64 oracle at the north-star tolerance, on a CONDITIONED state, at the
BASELINE batch size (48).  Round 5: ONE REPRODUCIBLE state, fixed hard bars, no skip paths, no warning tier.

What rounds 2-4 got wrong, and what was found (scratch/bf16_state_analysis.py; HISTORY.md section 4 "Round 5"):

  * The conditioning run (f32 HIP training on structured synthetic frames) used the float-atomic library, so every run ended on a
    different state -- after 40 steps two f32 runs already differ by 6 % in the parameters (scratch/det_probe.py) -- and the bars were
    then fitted to whatever states turned up.  Now the run uses the REPRODUCIBLE twin library (libspb_hip_det.so: exact,
    order-independent accumulation in place of float atomics; include/spb_hip.h "reproducible mode"): the conditioned state is
    bit-identical in every process and on every box (its sha256 is printed; the first steps are run twice here and compared).
  * The frames carried per-pixel clutter of amplitude 0.25 under the blob.  A network trained on them is CHAOTIC: on the float64 CPU
    oracle a 2^-9 relative perturbation at the 112x112 layers grows to 15-22 % in the 7x7 activations, loss spikes of several hundred
    appear in the (deterministic, f32) training run, and rounds 2-5 found the gradient of their bfloat16 pass unrelated to the float64 one
    (the oracle with bf16 rounding at the storage / operand points: cosine 0.58, norm ratio 0.46; PyTorch's own CPU bf16 autocast of the
    oracle: 0.84 / 0.68; float16 rounding instead: 0.98 / 1.00; the HIP pass anywhere between -0.16 and 0.47 depending on the order of its
    float additions).  ROUND 6 FOUND THE CAUSE: it is the rounding of the network's INPUT.  One rounding family at a time on the oracle
    (scratch/bf16_state_analysis_r6.py, profiles/r6_bf16_state_analysis.txt): every stored tensor and every other operand costs 0.02-0.05
    of cosine each; the stem's operands -- the float32 image and the 864 stem weights rounded to bfloat16 -- cost more than all of them
    together: with ONLY them left unrounded the oracle's bf16 gradient goes from cosine 0.49 to 0.90.  The stem kernels now feed image
    and weights to the matrix cores as hi + lo pairs (csrc/stem_mfma.hip: exact to ~1e-5, +4 us), and the HIP bf16 pass on the chaotic state
    measures 0.909 (exact accumulation) / 0.911 (float atomics), norm ratio 0.96 / 0.92 -- end to end, asserted below.  The well-conditioned
    state (clutter 0.05: no loss spike) went from 0.9918 to 0.9945; it stays the state of the tight end-to-end bars.
  * A backward-kernel defect and forward chaos were indistinguishable in an end-to-end cosine.  They are separated now: the oracle
    takes its backward pass THROUGH THE HIP FORWARD STATE (oracle._Net.forced: the stored bf16 convolution outputs and predictions of
    the HIP pass replace its own), so the comparison holds exactly the backward kernels to float64 -- on the well-conditioned AND on the
    chaotic state.

Tests (all bars hard):
  * reproducibility of the conditioning run (two short runs bit-identical, no float atomic outside the exact regions);
  * eval forward: keypoint MSE <= 1e-4 (BASELINE.json north_star: "keypoint MSE within 1e-4 of reference");
  * train forward: loss within the keypoint budget; per-layer BatchNorm batch means (bounded error growth with depth);
  * backward through the HIP forward state vs float64: cosine >= 0.999, norm ratio 0.99 .. 1.01, every tensor >= 0.99, on both states;
  * end to end (identical inputs, targets shifted by 0.05 -- a coherent upstream gradient like that of a network still in training):
    cosine >= 0.97 (mean of 8 float-atomic passes; 0.95 for every single pass and for the exact-accumulation pass), norm ratio 0.9 .. 1.1;
  * the DANN step FusedTrainStep(dann=True) as bench.py --model dann runs it (two streams) and back to back, against oracle.DannTrainer
    (dann.py:68-100): pose loss within budget, domain terms 2e-2, clipped SGD update cosine >= 0.93, ratio 0.9 .. 1.1.
Reference: park2019.py:126-165, trainer.py:72-98, dann.py:68-100.
"""
import hashlib
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import krn_oracle as O  # noqa: E402
from speedplusbaseline_amd.engine import KrnEngine  # noqa: E402
from speedplusbaseline_amd.step import FusedTrainStep  # noqa: E402

B = 48
K = 11
SCHEDULE = {0: 1e-3, 300: 3e-4, 600: 1e-4, 1000: 3e-5, 1600: 1e-5}   # AdamW learning rate from step ...
STEPS = 2000
TARGET_SHIFT = 0.05        # the gradient bars are taken against targets shifted by this constant (module docstring)
N_RUNS = 8                 # identical float-atomic bf16 passes (mean and every single one are held to the bars)
CLEAN, CLUTTERED = 0.05, 0.25     # amplitude of the per-pixel clutter under the blob (module docstring)
CHAOTIC_E2E_COSINE = 0.80         # end-to-end bf16 gradient cosine on the chaotic state (PyTorch's CPU bf16 autocast of the oracle: 0.84; the review's bar: that - 0.05)
NOISE_AMP = float(os.environ.get("SPB_COND_NOISE", CLEAN))    # scratch/cond_explore.py sweeps it


def structured_batch(n, seed, device=None, noise=None):
    """frames with a soft blob at (cx, cy) of size s over per-pixel clutter; the 11 keypoints sit on a fixed constellation
    around the blob, so the targets are a learnable function of the image (as real keypoints are).  device=None: CPU
    generator (the batches both sides are compared on); a cuda device: generated there (conditioning stream only)"""
    dev = torch.device("cpu") if device is None else torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *sh: torch.rand(*sh, generator=g, device=dev)
    cx = 0.25 + 0.5 * r(n); cy = 0.25 + 0.5 * r(n)
    s = 0.08 + 0.08 * r(n)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 224, device=dev), torch.linspace(0, 1, 224, device=dev), indexing="ij")
    d2 = (xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2
    blob = torch.exp(-d2 / (2 * s[:, None, None] ** 2))
    amp = NOISE_AMP if noise is None else noise
    img = amp * r(n, 3, 224, 224) + 0.7 * blob[:, None] * torch.tensor([1.0, 0.8, 0.6], device=dev)[None, :, None, None]
    ang = torch.arange(K, device=dev) * (2 * math.pi / K)
    kx = cx[:, None] + 1.5 * s[:, None] * torch.cos(ang)[None]
    ky = cy[:, None] + 1.5 * s[:, None] * torch.sin(ang)[None]
    y = torch.stack([kx, ky], dim=1).clamp(0, 1)
    return img, y


def load_state(eng, sd):
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].detach().to(device=eng.device, dtype=torch.float32))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].detach().flatten().to(device=eng.device, dtype=torch.float32))
    for i, n in enumerate(eng.bn_names):
        eng.nbt[i] = int(sd[n])


def dump_state(eng, dtype=torch.float64):
    sd = {}
    for info in eng.param_infos:
        sd[info[0]] = eng.param_view(info).detach().cpu().to(dtype).clone()
    for name, shape, off, numel in eng.buffer_infos:
        sd[name] = eng.buffers[off: off + numel].detach().cpu().to(dtype).view(shape).clone()
    for i, n in enumerate(eng.bn_names):
        sd[n] = torch.tensor(int(eng.nbt[i]), dtype=torch.int64)
    # the oracle walks the reference's state_dict order
    order = list(O.krn_param_shapes(K, dann=eng.dann).keys())
    return {k: sd[k] for k in order}


def digest(eng):
    h = hashlib.sha256()
    for t in (eng.params, eng.buffers, eng.nbt):
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def _train(device, steps, noise):
    """the conditioning run: f32 HIP path on the reproducible library, AdamW (wd 0.01, clip 1.0) on SCHEDULE, a fresh structured batch
    every step (seed 100 + step).  Returns (engine, losses [steps] on the host)"""
    eng = KrnEngine(K, deterministic=True).attach(device, "fp32")
    load_state(eng, O.init_state(K))
    ts = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0)
    losses = []
    for it in range(steps):
        if it in SCHEDULE:
            ts.lr = SCHEDULE[it]
        x, y = structured_batch(B, 100 + it, device, noise)        # a fresh batch every step: the network has to generalise
        losses.append(ts(x, y)[0:1].clone())
    torch.cuda.synchronize()
    return eng, torch.cat(losses).cpu()


def _condition(device, noise):
    eng, losses = _train(device, STEPS, noise)
    misses = eng.det_misses()
    tail = losses[-50:]
    print("conditioned state (clutter %.2f): sha256 %s after %d reproducible f32 steps; loss every 200 steps %s; last 50: median %.5f max %.5f; "
          "largest loss of the run %.3f; float atomics outside the exact regions: %d"
          % (noise, digest(eng), STEPS, [round(float(v), 4) for v in losses[::200]], float(tail.median()), float(tail.max()), float(losses.max()), misses))
    assert misses == 0
    assert torch.isfinite(losses).all()
    return dump_state(eng)


@pytest.fixture(scope="module")
def conditioned(device):
    """the well-conditioned state (clutter 0.05): ONE reproducible state per session, shared by the tests of this module"""
    sd = _condition(device, CLEAN)
    return sd


@pytest.fixture(scope="module")
def conditioned_cluttered(device):
    """the chaotic state (clutter 0.25; module docstring)"""
    return _condition(device, CLUTTERED)


@pytest.fixture(scope="module")
def conditioned_dann(conditioned):
    """RevGrad state for the DANN step: the conditioned backbone under the reference's `net.` prefix (revgrad.py:62-64) + the domain
    classifier at its initial state (what adapt.py starts from when it loads a KRN checkpoint: adapt.py:92-94)"""
    init = O.init_state(K, dann=True)
    sd = {}
    for k in O.krn_param_shapes(K, dann=True).keys():
        sd[k] = conditioned[k[4:]].clone() if k.startswith("net.") else init[k].double().clone() if init[k].is_floating_point() else init[k].clone()
    return sd


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu().flatten(); b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


def test_conditioning_run_is_reproducible(device):
    """two runs of the first 25 conditioning steps from scratch: parameters, BatchNorm buffers and every loss bit-identical; no float
    atomic escaped the exact accumulation.  (Across processes and boxes: the printed sha256 of the conditioned states, profiles/r5_*.)"""
    a, la = _train(device, 25, CLEAN)
    da, ma = digest(a), a.det_misses()
    del a
    b, lb = _train(device, 25, CLEAN)
    assert da == digest(b) and torch.equal(la, lb), (da, digest(b))
    assert ma == 0 and b.det_misses() == 0
    # ... while the float-atomic library (what bench.py times) is NOT reproducible run to run: that is what the twin is for
    print("25-step digest %s (twice); losses %.6f .. %.6f" % (da, float(la[0]), float(la[-1])))


def test_bf16_eval_keypoints_within_1e4_of_float64_oracle_at_bs48(device, conditioned):
    x, y = structured_batch(B, 7, noise=CLEAN)         # a batch the network has not seen
    sd = {k: v.clone() for k, v in conditioned.items()}
    with torch.no_grad():
        xc, yc = O.krn_forward(sd, x.double(), None, training=False)
    ref = torch.stack([xc, yc], dim=1)                  # [B,2,K]
    out = {}
    for prec in ("fp32", "bf16"):
        eng = KrnEngine(K).attach(device, prec)
        load_state(eng, conditioned)
        pred, _, _ = eng.forward(x.to(device), None, training=False)
        torch.cuda.synchronize()
        p = torch.stack([pred[:, 0::2], pred[:, 1::2]], dim=1).double().cpu()
        out[prec] = (float(((p - ref) ** 2).mean()), float((p - ref).abs().max()))
    fit = float(((ref - y.double()) ** 2).mean())
    print("conditioned eval, B=48: keypoint MSE vs float64 oracle  fp32 %.3e (max |d| %.2e)   bf16 %.3e (max |d| %.2e);  "
          "oracle-vs-target MSE %.3e, reference mean square %.3e" % (out["fp32"] + out["bf16"] + (fit, float((ref ** 2).mean()))))
    assert fit < 1e-3                               # the state is a trained one: it predicts the keypoints of an unseen batch
    assert out["fp32"][0] <= 1e-8
    assert out["bf16"][0] <= 1e-4, out             # the north-star bar, in the benchmarked dtype


def _oracle_grad(state, x, y, forced=None, forced_out=None):
    """float64 loss and gradient (flat, state-dict order) of the oracle; forced / forced_out: through another forward state (module docstring)"""
    sd = {k: v.clone() for k, v in state.items()}
    names = O._leafify(sd)
    O._Net.forced = forced
    try:
        out, _ = O.krn_predict(sd, x.double(), True, "")
    finally:
        O._Net.forced = None
    if forced_out is not None:
        out = out + (forced_out.double() - out).detach()
    loss = O.krn_loss(out, y.double())[0]
    loss.backward()
    return float(loss), torch.cat([sd[k].grad.flatten() for k in names]), sd, names


def _hip_grad(eng, state, x, y, device):
    load_state(eng, state)                         # (the forward moves the running statistics)
    eng.grads.zero_()
    pred, scal, _ = eng.forward(x.to(device), y.to(device), training=True)
    eng.backward(B)
    torch.cuda.synchronize()
    unscale = 1.0 / float(eng.amp[0]) if getattr(eng, "half", False) else 1.0      # float16: the arena holds loss-scale x gradient
    return (scal.cpu().double(), torch.cat([eng.param_view(i, eng.grads).double().cpu().flatten() for i in eng.param_infos]) * unscale, pred)


@pytest.mark.parametrize("which", ["clean", "cluttered"])
def test_bf16_backward_through_its_own_forward_state_vs_float64(device, conditioned, conditioned_cluttered, which):
    """the bf16 backward kernels alone: the float64 oracle takes its backward pass through the forward state the HIP pass stored (raw
    convolution outputs of all 58 BatchNorm'd tensors + the predictions), so forward rounding -- which the chaotic state amplifies
    -- is common to both sides.  Holds on every state; bars fixed."""
    state, noise = (conditioned, CLEAN) if which == "clean" else (conditioned_cluttered, CLUTTERED)
    x, y = structured_batch(B, 8, noise=noise)
    y = (y + TARGET_SHIFT).clamp(0, 1.2)
    eng = KrnEngine(K).attach(device, "bf16")
    s, g_hip, pred = _hip_grad(eng, state, x, y, device)
    forced = {k: v.detach().double().cpu() for k, v in eng.activations(B).items()}
    assert len(forced) == 58
    loss_f, g_forced, sd, names = _oracle_grad(state, x, y, forced, pred.cpu())
    loss_0, g_free, _, _ = _oracle_grad(state, x, y)
    cos, ratio = _cos(g_hip, g_forced), float(g_hip.norm() / g_forced.norm())
    print("%s state: bf16 loss %.5f, float64 through the HIP forward state %.5f, float64 on its own %.5f" % (which, float(s[0]), loss_f, loss_0))
    print("%s state: HIP bf16 gradient vs float64 THROUGH THE SAME FORWARD STATE: cosine %.4f, norm ratio %.4f   (end to end, for information: "
          "cosine %.4f, norm ratio %.4f; the two float64 gradients: cosine %.4f)"
          % (which, cos, ratio, _cos(g_hip, g_free), float(g_hip.norm() / g_free.norm()), _cos(g_forced, g_free)))
    gn = float(g_forced.norm())
    off, per = 0, []
    for k in names:
        n = sd[k].numel()
        a, b = g_hip[off: off + n], g_forced[off: off + n]
        off += n
        if float(b.norm()) > 1e-3 * gn:
            per.append((_cos(a, b), float(a.norm() / b.norm()), k))
    per.sort()
    print("   eight lowest per-tensor cosines among tensors above 1e-3 of |g| (cosine, norm ratio, tensor): " + "; ".join("%.3f %.2f %s" % t for t in per[:8]))
    assert abs(float(s[0]) - loss_f) <= 1e-3 * loss_f + 1e-6          # same predictions -> same loss
    # measured: cosine 1.0000 / ratio 1.0000 on the well-conditioned state, 0.9998 / 0.9995 on the chaotic one; lowest per-tensor cosine 0.997
    assert cos >= 0.999 and 0.99 <= ratio <= 1.01, (cos, ratio)
    assert per[0][0] >= 0.99, per[0]
    # the same float-atomic pass END TO END (round 6, since the stem takes image and weights unrounded: module docstring): 0.9947 / 0.998 on the
    # well-conditioned state, 0.911 / 0.917 on the chaotic one (rounds 2-5: 0.12 ... 0.47 there)
    e2e = (_cos(g_hip, g_free), float(g_hip.norm() / g_free.norm()))
    assert e2e[0] >= (0.97 if which == "clean" else CHAOTIC_E2E_COSINE) and 0.8 <= e2e[1] <= 1.25, e2e


@pytest.mark.parametrize("which", ["clean", "cluttered"])
def test_fp16_gradient_end_to_end_and_through_its_own_forward_state(device, conditioned, conditioned_cluttered, which):
    """float16 -- the reference's OWN mixed-precision recipe for KRN (autocast + GradScaler, train.py:101-104, trainer.py:73-94) -- through
    the IEEE-half build of the same kernels (KrnEngine.attach(device, "fp16")).  Three more mantissa bits than bfloat16: the CPU analysis
    (module docstring) predicted that float16 rounding keeps the gradient of even the CHAOTIC state at cosine 0.98 where every bfloat16
    evaluation gives 0.1 - 0.6.  Measured here on the HIP path, end to end, on both states; and the backward kernels alone through the
    pass's own forward state.  The gradient arena holds loss-scale x gradient (scale 65536 on the device); finite, no overflow."""
    state, noise = (conditioned, CLEAN) if which == "clean" else (conditioned_cluttered, CLUTTERED)
    x, y = structured_batch(B, 8, noise=noise)
    y = (y + TARGET_SHIFT).clamp(0, 1.2)
    eng = KrnEngine(K).attach(device, "fp16")
    assert eng.half and float(eng.amp[0]) == 65536.0
    s, g_hip, pred = _hip_grad(eng, state, x, y, device)
    assert torch.isfinite(g_hip).all() and torch.isfinite(pred).all()
    forced = {k: v.detach().double().cpu() for k, v in eng.activations(B).items()}
    assert next(iter(eng.activations(B).values())).dtype == torch.float16
    loss_f, g_forced, _, _ = _oracle_grad(state, x, y, forced, pred.cpu())
    loss_0, g_free, sd, names = _oracle_grad(state, x, y)
    e2e = (_cos(g_hip, g_free), float(g_hip.norm() / g_free.norm()))
    thr = (_cos(g_hip, g_forced), float(g_hip.norm() / g_forced.norm()))
    per = _per_tensor(g_hip, g_free, sd, names)
    print("%s state, float16: loss %.5f (float64 %.5f); gradient vs float64 END TO END: cosine %.4f, norm ratio %.4f; through its own forward "
          "state: cosine %.4f, norm ratio %.4f; lowest per-tensor cosines end to end: %s"
          % (which, float(s[0]), loss_0, e2e[0], e2e[1], thr[0], thr[1], "; ".join("%.3f %.2f %s" % t for t in per[:5])))
    # per tensor (above 1e-3 of |g|), end to end, both states (bf16: 0.60 / 0.65, test below)
    assert per[0][0] >= PER_TENSOR_FLOOR_FP16, per[:4]
    assert thr[0] >= 0.999 and 0.99 <= thr[1] <= 1.01, thr
    assert abs(float(s[0]) - loss_0) <= 0.02 * loss_0 + 1e-5
    # end to end: the well-conditioned state to 0.995 (measured 0.9992); the chaotic state above 0.95 (0.9875; bfloat16: 0.91)
    assert e2e[0] >= (0.995 if which == "clean" else 0.95) and 0.9 <= e2e[1] <= 1.1, e2e


def test_bf16_train_pass_tracks_float64_oracle_at_bs48(device, conditioned):
    x, y = structured_batch(B, 8, noise=CLEAN)         # a batch the network has not seen
    y_shift = (y + TARGET_SHIFT).clamp(0, 1.2)
    sd = {k: v.clone() for k, v in conditioned.items()}
    names = O._leafify(sd)
    O._Net.momentum = 1.0                              # running statistics := this batch's statistics (per-layer probe)
    try:
        out, _ = O.krn_predict(sd, x.double(), True, "")
    finally:
        O._Net.momentum = O.BN_MOM
    loss = O.krn_loss(out, y.double())[0]
    loss_shift = O.krn_loss(out, y_shift.double())[0]
    loss_shift.backward()
    g_ref = torch.cat([sd[k].grad.flatten() for k in names])
    eng = KrnEngine(K).attach(device, "bf16")
    s, _, _ = _hip_grad(eng, conditioned, x, y, device)
    print("conditioned train pass, B=48: loss bf16 %.6f float64 %.6f" % (float(s[0]), float(loss)))
    # a per-coordinate keypoint budget of 1e-4 (mean square) moves the summed loss by at most 2 sqrt(L * 2K * 1e-4) + 2K * 1e-4
    budget = 2 * math.sqrt(float(loss) * 2 * K * 1e-4) + 2 * K * 1e-4
    assert abs(float(s[0]) - float(loss)) <= budget, (float(s[0]), float(loss), budget)
    # per-layer batch means (running_mean after a momentum-0.1 update from the same start): error growth with depth
    errs = []
    for name, shape, off, numel in eng.buffer_infos:
        if not name.endswith("running_mean"):
            continue
        got = (eng.buffers[off: off + numel].double().cpu() - 0.9 * conditioned[name].flatten()) / 0.1     # batch mean seen by HIP
        errs.append(_rel(got, sd[name]))
    print("per-layer batch-mean relative error (58 BN layers): first %.2e  median %.2e  max %.2e  last %.2e"
          % (errs[0], sorted(errs)[len(errs) // 2], max(errs), errs[-1]))
    # bf16 operand rounding (2^-9) at the stem, bounded growth after; the largest per-layer error sits in the last, near-zero-mean layers
    assert errs[0] < 5e-3 and max(errs) < 0.15 and errs[-1] < 0.15 and sorted(errs)[len(errs) // 2] < 5e-3
    # ---- the gradient end to end, FIXED bars, against targets shifted by TARGET_SHIFT (module docstring)
    runs = [_hip_grad(eng, conditioned, x, y_shift, device) for _ in range(N_RUNS)]
    g_hip = torch.stack([r_[1] for r_ in runs]).mean(0)
    single = [(_cos(r_[1], g_ref), float(r_[1].norm() / g_ref.norm())) for r_ in runs]
    print("single float-atomic bf16 passes vs float64 (cosine, norm ratio): " + "  ".join("%.4f %.3f" % t for t in single))
    cos, ratio = _cos(g_hip, g_ref), float(g_hip.norm() / g_ref.norm())
    print("gradient vs float64 (targets + %.2f), mean of %d passes: cosine %.4f, norm ratio %.4f   (|g| %.3e; loss bf16 %.5f float64 %.5f)"
          % (TARGET_SHIFT, N_RUNS, cos, ratio, float(g_ref.norm()), float(runs[0][0][0]), float(loss_shift)))
    det = KrnEngine(K, deterministic=True).attach(device, "bf16")
    sx, g_det, _ = _hip_grad(det, conditioned, x, y_shift, device)
    sx2, g_det2, _ = _hip_grad(det, conditioned, x, y_shift, device)
    cos_d, ratio_d = _cos(g_det, g_ref), float(g_det.norm() / g_ref.norm())
    print("exact-accumulation bf16 pass: cosine %.4f, norm ratio %.4f, loss %.5f (a second pass is bit-identical: %s)"
          % (cos_d, ratio_d, float(sx[0]), bool(torch.equal(g_det, g_det2))))
    gn = float(g_ref.norm())
    per = sorted((_cos(eng.param_view(i, eng.grads).double().cpu().flatten(), sd[i[0]].grad.flatten()),
                  float(eng.param_view(i, eng.grads).double().cpu().norm() / (sd[i[0]].grad.norm() + 1e-30)), i[0]) for i in eng.param_infos
                 if float(sd[i[0]].grad.norm()) > 1e-3 * gn)
    print("eight lowest per-tensor cosines of the last pass among tensors above 1e-3 of |g| (cosine, norm ratio, tensor): "
          + "; ".join("%.3f %.2f %s" % t for t in per[:8]))
    assert torch.equal(g_det, g_det2) and torch.equal(sx, sx2)                 # one state, one batch -> one gradient
    # measured on this (reproducible) state, round 6: mean 0.993 / 1.016, singles 0.991 .. 0.994 / 1.009 .. 1.026, exact accumulation 0.991 / 1.022
    # (round 5: 0.990 / 1.023, 0.985 .. 0.991, and 0.981 for the exact pass -- which then lacked a flush, see the per-tensor test below)
    assert cos >= 0.97 and 0.9 <= ratio <= 1.1, (cos, ratio)
    assert min(c for c, _ in single) >= 0.95 and all(0.9 <= r_ <= 1.1 for _, r_ in single), single
    assert cos_d >= 0.97 and 0.9 <= ratio_d <= 1.1, (cos_d, ratio_d)


PER_TENSOR_FLOOR_FP16 = 0.88   # measured: lowest 0.969 (clean state, base.0.0.weight) / 0.974 (chaotic state); before the stem took its operands unrounded: 0.935 / 0.970


def _per_tensor(g, g_ref, sd, names, floor_share=1e-3):
    gn, off, per = float(g_ref.norm()), 0, []
    for k in names:
        n = sd[k].numel()
        a, b = g[off: off + n], g_ref[off: off + n]
        off += n
        if float(b.norm()) > floor_share * gn:
            per.append((_cos(a, b), float(a.norm() / b.norm()), k))
    return sorted(per)


# bars of the test below; measured values in its docstring
PER_TENSOR_FLOOR_CLEAN = 0.60
PER_TENSOR_FLOOR_CHAOTIC = 0.25   # single small tensors still move on this state: lowest 0.82 and 0.45 with two builds of the stem kernel that differ in the order of two f32 additions


def test_bf16_end_to_end_per_tensor_floor_and_chaotic_state_bar(device, conditioned, conditioned_cluttered):
    """Round-5 review, "parity soft spots": (1) on the well-conditioned state the end-to-end bf16 gradient was only held as a whole (cosine
    0.99, carried by the large tensors) while the first BatchNorm affines sat at 0.56-0.63; (2) on the chaotic state nothing was asserted
    end to end.  Both are asserted here on the EXACT-ACCUMULATION bf16 pass -- one state, one batch, one bit-reproducible gradient per
    build, so the bars carry no run-to-run noise:
      * clean state: whole gradient >= 0.95 (measured 0.9945 / ratio 0.999); every tensor above 1e-3 of |g| >= PER_TENSOR_FLOOR_CLEAN
        (measured: lowest 0.721 base.0.0.weight, then 0.778 / 0.805 / 0.805; before the stem took its operands unrounded: 0.635);
      * chaotic state: whole gradient >= CHAOTIC_E2E_COSINE and norm ratio 0.8 .. 1.25 (measured 0.9087 / 0.964; float-atomic pass 0.911 /
        0.917), every tensor >= PER_TENSOR_FLOOR_CHAOTIC (a weak floor: the whole gradient is stable there -- 0.9087 / 0.9093 exact accumulation, 0.911 / 0.905
        float atomics over two builds of the stem kernel -- single small tensors are not: lowest 0.821 base.2.conv.3.weight with one, 0.452 base.7.conv.0.1.weight with the other).  Rounds 2-5 measured anything between -0.16
        and 0.58 here and concluded that no bfloat16 evaluation of this state can track float64.  That was wrong: what the state amplifies
        is the rounding of the INPUT IMAGE (and the 864 stem weights) to bfloat16 -- module docstring -- and the stem no longer rounds them.
    History of this test: (a) round 6 first removed the rounding of the expanded tensors of blocks 2-4 (recomputed in f32, Runner::virt) --
    no visible effect on this state; (b) the test found a bug of the reproducible build: the fused pointwise backward returned without
    folding its exact batch sums, so the depthwise backward of blocks 1-3 rebuilt dz from zero sums (BatchNorm weight gradients of the first
    layers 8-17x too large, whole-gradient cosine 0.983 instead of 0.991; f32 unaffected)."""
    det = KrnEngine(K, deterministic=True).attach(device, "bf16")
    out = {}
    for which, state, noise in (("clean", conditioned, CLEAN), ("cluttered", conditioned_cluttered, CLUTTERED)):
        x, y = structured_batch(B, 8, noise=noise)
        y = (y + TARGET_SHIFT).clamp(0, 1.2)
        _, g_ref, sd, names = _oracle_grad(state, x, y)
        _, g_det, _ = _hip_grad(det, state, x, y, device)
        # (_hip_grad returns the arena in the plan's parameter order == state-dict order of `names`)
        assert [i[0] for i in det.param_infos] == list(names)
        per = _per_tensor(g_det, g_ref, sd, names)
        out[which] = (_cos(g_det, g_ref), float(g_det.norm() / g_ref.norm()), per)
        print("%s state, exact-accumulation bf16 pass END TO END vs float64: cosine %.4f, norm ratio %.4f; lowest per-tensor cosines (tensors above "
              "1e-3 of |g|): %s" % (which, out[which][0], out[which][1], "; ".join("%.3f %.2f %s" % t for t in per[:6])))
    assert out["clean"][2][0][0] >= PER_TENSOR_FLOOR_CLEAN, out["clean"][2][:4]
    assert out["clean"][0] >= 0.95
    assert out["cluttered"][0] >= CHAOTIC_E2E_COSINE and 0.8 <= out["cluttered"][1] <= 1.25, out["cluttered"][:2]
    assert out["cluttered"][2][0][0] >= PER_TENSOR_FLOOR_CHAOTIC, out["cluttered"][2][:4]


def test_bf16_dann_step_overlapped_streams_vs_oracle(device, conditioned_dann):
    """one DANN step through FusedTrainStep(dann=True) -- source and target passes concurrently on two streams, as
    bench.py --model dann times it -- in bf16 (and f32), against oracle.DannTrainer in float64 from the same state; fixed bars"""
    NB = B      # the batch size the backbone was conditioned at: its BatchNorm layers expect 48-image statistics (the README's DANN recipe
                # uses 16; bench.py times both).  At 16 images the 7x7 layers have 784 samples per channel and the training-mode loss of a
                # network conditioned at 48 is dominated by that mismatch, in float64 as much as in bf16
    xs, ys = structured_batch(NB, 21, noise=CLEAN); xt = structured_batch(NB, 22, noise=CLEAN)[0].flip(3) * 0.8
    ys = (ys + TARGET_SHIFT).clamp(0, 1.2)             # a coherent pose gradient (module docstring)
    alpha = 0.7
    sd = {k: v.clone() for k, v in conditioned_dann.items()}
    tr = O.DannTrainer(sd, kind="sgd", lr=0.05, momentum=0.0, weight_decay=0.0)
    p0 = torch.cat([sd[k].detach().flatten().clone() for k in tr.names])
    lp, ls, lt, gn = _dann_step_f64(tr, xs.double(), ys.double(), xt.double(), alpha)
    d_ref = torch.cat([sd[k].detach().flatten() for k in tr.names]) - p0
    budget = 2 * math.sqrt(lp * 2 * K * 1e-4) + 2 * K * 1e-4
    for prec, overlap in (("fp32", "1"), ("bf16", "1"), ("bf16", "0")):
        os.environ["SPB_DANN_OVERLAP"] = overlap
        try:
            eng = KrnEngine(K, dann=True).attach(device, prec)
            load_state(eng, conditioned_dann)
            ts = FusedTrainStep(eng, NB, kind="sgd", lr=0.05, momentum=0.0, weight_decay=0.0, max_norm=1.0, dann=True)
            q0 = eng.params.clone()
            s = ts(xs.to(device), ys.to(device), xt.to(device), alpha=alpha)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("SPB_DANN_OVERLAP", None)
        s = s.cpu().double()
        print("DANN %s (overlap=%s): pose %.5f / %.5f   bce source %.5f / %.5f   bce target %.5f / %.5f  (hip / float64)"
              % (prec, overlap, float(s[0]), lp, float(s[3]), ls, float(s[4]), lt))
        d_hip = torch.cat([(eng.param_view(i) - eng.param_view(i, q0)).double().cpu().flatten() for i in eng.param_infos])
        cos, ratio = _cos(d_hip, d_ref), float(d_hip.norm() / d_ref.norm())
        print("   SGD update after clip: cosine to float64 %.4f, norm ratio %.4f" % (cos, ratio))
        if prec == "fp32":
            assert abs(float(s[0]) - lp) <= 1e-3 * lp + 1e-5 and abs(float(s[3]) - ls) <= 1e-4 and abs(float(s[4]) - lt) <= 1e-4
            assert cos > 0.99 and 0.98 < ratio < 1.02
            continue
        # bf16: the training-mode pose loss within the keypoint budget (+ 5 %), the two domain terms to 2e-2, the clipped SGD update at
        # fixed bars.  (The update mixes the pose gradient with the gradients of the two domain terms, reversed at the 7x7 feature and
        # coming from a domain classifier at its initial state: an incoherent component whose bf16 evaluation is noise-dominated.)
        assert abs(float(s[0]) - lp) <= budget + 0.05 * lp, (float(s[0]), lp, budget)
        assert abs(float(s[3]) - ls) <= 2e-2 and abs(float(s[4]) - lt) <= 2e-2
        assert cos >= 0.93 and 0.9 <= ratio <= 1.1, (cos, ratio)          # measured: 0.964 / 1.0000 (both launch modes)


def _dann_step_f64(tr, xs, ys, xt, alpha):
    """oracle.DannTrainer.step with float64 domain labels (F.binary_cross_entropy_with_logits needs matching dtypes)"""
    import torch.nn.functional as F
    n = xs.shape[0]
    tr.opt.zero_grad(set_to_none=True)
    (lp, lx, ly), ds = O.revgrad_forward(tr.sd, xs, ys, alpha, True)
    l_src = F.binary_cross_entropy_with_logits(ds, torch.ones(n, dtype=ds.dtype))
    _, dt = O.revgrad_forward(tr.sd, xt, None, alpha, True)
    l_tgt = F.binary_cross_entropy_with_logits(dt, torch.zeros(n, dtype=dt.dtype))
    (lp + l_src + l_tgt).backward()
    gn = torch.nn.utils.clip_grad_norm_([tr.sd[k] for k in tr.names], 1.0)
    tr.opt.step()
    return float(lp), float(l_src), float(l_tgt), float(gn)
