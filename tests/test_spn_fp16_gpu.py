"""SPN in float16 with dynamic loss scaling -- BASELINE.json configs[5] ("SPN ... fp16") and SURVEY 8 row a13: the reference's
recipe is torch.cuda.amp autocast + GradScaler (train.py:101-104, trainer.py:146-181).  Here: IEEE-half activations and weight
shadows on v_mfma_f32_16x16x32_f16 (libspb_hip_f16.so, the SPN sources compiled with -DSPB_F16), f32 master weights, and
GradScaler's arithmetic on the device (spb_softce_scaled / spb_amp_check / spb_amp_step / spb_optim_step(skip)).
Golden vectors: the reference's own SpacecraftPoseNet on the CPU (tests/golden/make_golden_spn.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import spn_oracle as S
from speedplusbaseline_amd import _lib as L
from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet, softmax_cross_entropy_with_logits
from speedplusbaseline_amd.optim import SpnOptimizer

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "spn_golden.npz"))
NC, B = 5000, 32


def digest(t):
    return np.array(S.checksum(t.detach().float().cpu()))


@pytest.fixture(scope="module")
def init():
    return S.init_state(NC), S.synth_batch(B, NC, seed=23)


def test_full_size_forward_fp16(device, init):
    """logits and the three loss reductions at 5000 classes / bs=32 / 227x227 in float16 (10-bit mantissa: tighter than bf16's 3e-2)"""
    sd, (x, yc, yw) = init
    net = SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False, precision="fp16")
    net.load_state_dict(sd, strict=True)
    net = net.to(device).eval()
    c, r = net(x.to(device))
    torch.cuda.synchronize()
    assert c.shape == (B, NC) and net._shadow.dtype == torch.float16
    tol = 8e-3
    scale_c, scale_r = float(GOLD["full_c_sum"][1]), float(GOLD["full_r_sum"][1])
    assert np.abs(c[:4, :8].float().cpu().numpy() - GOLD["full_c_crop"]).max() < tol * 10 * scale_c
    assert np.abs(r[-4:, -8:].float().cpu().numpy() - GOLD["full_r_crop"]).max() < tol * 10 * scale_r
    assert abs(digest(c)[1] - scale_c) < tol * scale_c and abs(digest(r)[1] - scale_r) < tol * scale_r
    lc = softmax_cross_entropy_with_logits(c, yc.to(device), "mean"); lr = softmax_cross_entropy_with_logits(r, yw.to(device), "mean")
    loss, lc_ref, lr_ref = GOLD["full_losses"]
    print("fp16: class %.6f / %.6f  regress %.6f / %.6f (hip / reference)" % (float(lc), lc_ref, float(lr), lr_ref))
    assert abs(float(lc) - lc_ref) < tol * lc_ref and abs(float(lr) - lr_ref) < tol * lr_ref


def test_full_size_training_gradients_fp16_with_loss_scale(device, init):
    """loss_and_grads in float16: the gradient arena carries the loss scale (here 1024); unscaled it matches the reference's
    gradients -- per-tensor mean |g| within 3 %, scattered samples within a few % of the tensor's largest"""
    sd, (x, yc, yw) = init
    net = SpacecraftPoseNet(NC, keep_prob=0.0, pretrain=False, precision="fp16")
    net.load_state_dict(sd, strict=True)
    net = net.to(device).train()
    scale = 1024.0
    net.amp_state(init_scale=scale)
    out = net.loss_and_grads(x.to(device), yc.to(device), yw.to(device))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    loss, lc_ref, lr_ref = GOLD["full_losses"]
    assert abs(o[0] - loss) < 8e-3 * loss and abs(o[1] - lc_ref) < 8e-3 * lc_ref and abs(o[2] - lr_ref) < 8e-3 * lr_ref    # the loss is NOT scaled
    assert torch.isfinite(net.flat_grads()).all()
    worst = 0.0
    for k, p in net.named_parameters():
        g = p.grad.flatten() / scale
        got, want = digest(g), GOLD["full_grad_sum/" + k]
        n = p.numel()
        assert abs(got[1] - want[1]) < 0.03 * want[1], (k, got[1], want[1])
        idx = (torch.arange(64, dtype=torch.int64) * 7919) % n
        smp, ref = g[idx.to(g.device)].float().cpu().numpy(), GOLD["full_grad_samples/" + k]
        err, top = np.abs(smp - ref), np.abs(ref).max()
        worst = max(worst, float(np.median(err) / top))
        assert np.median(err) < 0.01 * top + 0.01 * want[1] and (err > 0.3 * top).mean() < 0.1, (k, np.median(err), err.max(), top)
    print("fp16 gradients: worst median sample error %.2e of the tensor's largest sample" % worst)


def _small(device, precision, seed=3):
    net = SpacecraftPoseNet(64, keep_prob=0.5, pretrain=False, precision=precision)
    net.load_state_dict(S.init_state(64), strict=True)
    net = net.to(device).train()
    x, yc, yw = (t.to(device) for t in S.synth_batch(4, 64, seed=seed))
    masks = {k: v.to(device) for k, v in S.synth_masks(4, seed=5).items()}
    return net, x, yc, yw, masks


def test_overflow_skips_the_step_and_halves_the_scale_then_training_resumes(device):
    """GradScaler semantics without a host sync: with an absurd scale every float16 gradient overflows -> found_inf -> the optimizer
    pass does nothing (parameters, moments and its step count unchanged) and the scale halves; once the scale is sane the step
    is taken, the step count advances and the update equals the float32 path's to float16 accuracy"""
    net, x, yc, yw, masks = _small(device, "fp16")
    opt = SpnOptimizer(list(net.parameters()), kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, model=net)
    st = net.amp_state(init_scale=2.0 ** 40)
    p0 = net.flat_parameters().clone()
    for k in range(3):
        net.loss_and_grads(x, yc, yw, masks=masks, optimizer=opt)
        opt.step()
        torch.cuda.synchronize()
        assert float(st[L.AMP_SCALE]) == 2.0 ** (39 - k) and float(st[L.AMP_SKIP]) == 1.0 and float(st[L.AMP_STEPS]) == 0.0
        assert torch.equal(net.flat_parameters(), p0)
        assert float(opt._m.abs().max()) == 0.0 and float(opt._v.abs().max()) == 0.0
    st[L.AMP_SCALE] = 4096.0
    net.loss_and_grads(x, yc, yw, masks=masks, optimizer=opt)
    opt.step()
    torch.cuda.synchronize()
    assert float(st[L.AMP_SKIP]) == 0.0 and float(st[L.AMP_STEPS]) == 1.0 and float(st[L.AMP_SCALE]) == 4096.0 and float(st[L.AMP_TRACKER]) == 1.0
    d16 = net.flat_parameters() - p0
    assert float(d16.abs().max()) > 0 and torch.isfinite(net.flat_parameters()).all()
    # the same first AdamW step in float32
    ref, x2, yc2, yw2, masks2 = _small(device, "fp32")
    opt32 = SpnOptimizer(list(ref.parameters()), kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, model=ref)
    q0 = ref.flat_parameters().clone()
    ref.loss_and_grads(x2, yc2, yw2, masks=masks2, optimizer=opt32)
    opt32.step()
    torch.cuda.synchronize()
    d32 = ref.flat_parameters() - q0
    cos = float((d16 * d32).sum() / (d16.norm() * d32.norm()))
    print("first AdamW update fp16 vs fp32: cosine %.4f, |d| %.4e / %.4e" % (cos, float(d16.norm()), float(d32.norm())))
    assert cos > 0.97 and abs(float(d16.norm()) / float(d32.norm()) - 1.0) < 0.05


def test_scale_grows_after_the_growth_interval(device):
    net, x, yc, yw, masks = _small(device, "fp16")
    opt = SpnOptimizer(list(net.parameters()), kind="sgd", lr=1e-4, momentum=0.9, weight_decay=1e-4, model=net)
    opt.amp_interval = 2
    st = net.amp_state(init_scale=256.0)
    scales = []
    for _ in range(5):
        net.loss_and_grads(x, yc, yw, masks=masks)
        opt.step()
        torch.cuda.synchronize()
        scales.append(float(st[L.AMP_SCALE]))
    assert scales == [256.0, 512.0, 512.0, 1024.0, 1024.0], scales
    assert float(st[L.AMP_STEPS]) == 5.0


def test_generic_path_batch_above_64_carries_the_loss_scale(device):
    """with B > 64 (or num_classes % 8 != 0) loss_and_grads takes its generic branch; in float16 the loss
    gradient must carry the scale there too, because SpnOptimizer._step_fp16 divides EVERY gradient by it.  Unscaled, the
    float16 gradients must equal the float32 path's on the same batch; the update must have the float32 update's size."""
    Bbig, NCs = 72, 64
    sd = S.init_state(NCs)
    x, yc, yw = (t.to(device) for t in S.synth_batch(Bbig, NCs, seed=31))
    grads, upd = {}, {}
    for prec in ("fp16", "fp32"):
        net = SpacecraftPoseNet(NCs, keep_prob=0.0, pretrain=False, precision=prec)
        net.load_state_dict(sd, strict=True)
        net = net.to(device).train()
        opt = SpnOptimizer(list(net.parameters()), kind="sgd", lr=0.05, momentum=0.0, weight_decay=0.0, model=net)
        if prec == "fp16":
            net.amp_state(init_scale=1024.0)
        assert not net._fast(Bbig)
        p0 = net.flat_parameters().clone()
        out = net.loss_and_grads(x, yc, yw, optimizer=opt)
        torch.cuda.synchronize()
        grads[prec] = net.flat_grads().clone() / (1024.0 if prec == "fp16" else 1.0)
        opt.step()
        torch.cuda.synchronize()
        upd[prec] = net.flat_parameters() - p0
        assert torch.isfinite(out).all()
    ce = net._conv_end
    for name, sl in (("fc", slice(ce, None)), ("conv", slice(0, ce))):
        g16, g32 = grads["fp16"][sl], grads["fp32"][sl]
        ratio = float(g16.norm() / g32.norm())
        cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
        print("B=72 generic path, %s gradients: |g16|/|g32| = %.4f, cosine %.4f" % (name, ratio, cos))
        assert abs(ratio - 1.0) < 0.05 and cos > 0.97, (name, ratio, cos)          # a missing scale shows as a ratio of 1/1024
    r = float(upd["fp16"].norm() / upd["fp32"].norm())
    assert abs(r - 1.0) < 0.05, r


def test_scaler_state_travels_with_the_optimizer_checkpoint(device):
    """loss scale, growth tracker and the count of steps really taken (Adam's bias corrections use it)
    live on the device; they are saved in SpnOptimizer.state_dict() and restored before the next forward scales a gradient."""
    net, x, yc, yw, masks = _small(device, "fp16")
    opt = SpnOptimizer(list(net.parameters()), kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, model=net)
    opt.amp_interval = 2
    st = net.amp_state(init_scale=512.0)
    for _ in range(3):
        net.loss_and_grads(x, yc, yw, masks=masks)
        opt.step()
    torch.cuda.synchronize()
    want = st.clone().cpu()
    assert float(want[L.AMP_STEPS]) == 3.0 and float(want[L.AMP_SCALE]) == 1024.0 and float(want[L.AMP_TRACKER]) == 1.0
    model_sd = {k: v.clone().cpu() for k, v in net.state_dict().items()}
    opt_sd = opt.state_dict()
    assert torch.equal(opt_sd["spn_fused"]["amp"], want)
    # resume in the reference's order: optimizer state restored while the model is still on the CPU (train.py:86-97)
    net2 = SpacecraftPoseNet(64, keep_prob=0.5, pretrain=False, precision="fp16")
    net2.load_state_dict(model_sd, strict=True)
    opt2 = SpnOptimizer(list(net2.parameters()), kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, model=net2)
    opt2.amp_interval = 2
    opt2.load_state_dict(opt_sd)
    assert torch.equal(opt2.state_dict()["spn_fused"]["amp"], want)              # saved again before any step: still there
    net2 = net2.to(device).train()
    st2 = net2.amp_state()
    assert torch.equal(st2.cpu(), want)
    # the 4th step of both runs: same scale on the loss gradient, same bias corrections -> same update
    p_a, p_b = net.flat_parameters().clone(), net2.flat_parameters().clone()
    assert torch.equal(p_a, p_b)
    for n_, o_ in ((net, opt), (net2, opt2)):
        n_.loss_and_grads(x, yc, yw, masks=masks)
        o_.step()
    torch.cuda.synchronize()
    assert float(st2[L.AMP_STEPS]) == 4.0 and float(st2[L.AMP_SCALE]) == 2048.0 and torch.equal(st2.cpu(), st.cpu())
    d_a, d_b = net.flat_parameters() - p_a, net2.flat_parameters() - p_b
    rel = float((d_a - d_b).norm() / d_a.norm())
    print("4th AdamW step, original vs resumed run: relative update difference %.2e" % rel)
    assert rel < 2e-2                                                             # float atomics in the gradients; a step count reset to 0 changes the update by tens of percent
