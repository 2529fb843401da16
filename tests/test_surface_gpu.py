"""GPU suite: the drop-in Python surface (src.nets / src.core import paths of the reference) on the MI355X -- the generic
autograd path (loss.backward(), clip_grad_norm_, optimizer.step()) and the fused fast path must agree with each other and
with the oracle; checkpoints round-trip; the epoch drivers run."""
import copy
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import krn_oracle as O  # noqa: E402

sys.argv = [sys.argv[0]]


def _cfg(**kw):
    c = types.SimpleNamespace(model_name="krn", num_keypoints=11, num_classes=5000, dann=False, optimizer="adamw", lr=1e-4,
                              momentum=0.9, weight_decay=0.01, fp16=False, precision="fp32", max_epochs=5, texture_ratio=0.5)
    c.__dict__.update(kw)
    return c


def _rel(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_generic_autograd_path_matches_fused_path_and_oracle(device):
    from src.nets import get_model, get_optimizer
    from torch.nn.utils import clip_grad_norm_
    x, y = O.synth_batch(4)
    sd0 = O.init_state(11)
    # oracle (float64): loss and one clipped SGD step
    sdo = O.init_state(11, dtype=torch.float64)
    tr = O.KrnTrainer(sdo, "sgd", lr=0.05, momentum=0.9, weight_decay=5e-5)
    loss_o = tr.step(x.double(), y.double())[0]
    results = {}
    for path in ("generic", "fused"):
        cfg = _cfg(optimizer="sgd", lr=0.05, weight_decay=5e-5)
        model = get_model(cfg)
        model.load_state_dict(sd0, strict=True)
        opt = get_optimizer(cfg, model)
        model = model.to(device)
        model.train()
        if path == "generic":
            loss, sm = model(x.to(device), y.to(device))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            gn = clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            results[path] = (float(loss), sm, copy.deepcopy({k: v.detach().cpu() for k, v in model.state_dict().items()}), float(gn))
        else:
            s = opt.train_step(x.to(device), y.to(device))
            results[path] = (float(s[0]), {"loss_x": float(s[1]), "loss_y": float(s[2])},
                             copy.deepcopy({k: v.detach().cpu() for k, v in model.state_dict().items()}), None)
    lg, lf = results["generic"][0], results["fused"][0]
    assert abs(lg - loss_o) < 1e-4 * loss_o and abs(lf - loss_o) < 1e-4 * loss_o
    assert isinstance(results["generic"][1]["loss_x"], float)
    for k in ("base.0.1.weight", "base.9.conv.1.1.weight", "extras.3.conv.4.weight", "head.0.bias", "head.0.weight"):
        d_g = results["generic"][2][k].double() - sd0[k].double()
        d_f = results["fused"][2][k].double() - sd0[k].double()
        d_o = sdo[k].detach().double() - sd0[k].double()
        assert _rel(d_g, d_f) < 0.05, k       # same kernels, different launch path (atomics order only)
        assert _rel(d_f, d_o) < 0.25, k       # vs oracle: fp32 mask-flip noise (see test_krn_gpu.py)
    assert int(results["fused"][2]["base.0.1.num_batches_tracked"]) == 1
    assert int(results["generic"][2]["base.0.1.num_batches_tracked"]) == 1


def test_eval_forward_and_checkpoint_roundtrip(device, tmp_path):
    from src.nets import get_model
    from speedplusbaseline_amd.utils import save_checkpoint
    x, _ = O.synth_batch(4)
    model = get_model(_cfg())
    model.load_state_dict(O.init_state(11), strict=True)
    model = model.to(device).eval()
    with torch.no_grad():
        xc, yc = model(x.to(device))
    assert xc.device.type == "cpu" and tuple(xc.shape) == (4, 11)
    sdo = O.init_state(11, dtype=torch.float64)
    with torch.no_grad():
        xo, yo = O.krn_forward(sdo, x.double(), None, training=False)
    assert _rel(xc, xo) < 2e-4 and _rel(yc, yo) < 2e-4
    save_checkpoint({"epoch": 1, "model": "krn", "state_dict": model.state_dict(), "best_score": 1, "optimizer": {}}, True, str(tmp_path))
    m2 = get_model(_cfg())
    m2.load_state_dict(torch.load(str(tmp_path / "model_best.pth.tar"), map_location="cpu"), strict=True)
    m2 = m2.to(device).eval()
    with torch.no_grad():
        xc2, _ = m2(x.to(device))
    assert torch.equal(xc, xc2)


def test_epoch_drivers_run_krn_and_dann(device, capsys):
    from src.nets import get_model, get_optimizer
    from src.core.trainer import train_single_epoch_krn
    from src.core.dann import train_dann_single_epoch_krn
    from speedplusbaseline_amd.data import SyntheticKeypointLoader
    cfg = _cfg(precision="bf16", lr=1e-3)
    model = get_model(cfg); opt = get_optimizer(cfg, model); model = model.to(device)
    p0 = model.head[0].bias.detach().clone()
    train_single_epoch_krn(1, cfg, model, SyntheticKeypointLoader(8, 3), opt, None, device)
    assert not torch.equal(p0, model.head[0].bias.detach()) and torch.isfinite(model.head[0].weight).all()
    assert int(model.state_dict()["base.0.1.num_batches_tracked"]) == 3
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 3 and "exp_avg" in sd["state"][0]
    cfgd = _cfg(dann=True, lr=1e-4)
    rg = get_model(cfgd); optd = get_optimizer(cfgd, rg); rg = rg.to(device)
    w0 = rg.domain_classifier[0].weight.detach().clone()
    train_dann_single_epoch_krn(1, cfgd, rg, SyntheticKeypointLoader(4, 2), SyntheticKeypointLoader(4, 2, labels=False, seed=7), optd, None, device)
    assert not torch.equal(w0, rg.domain_classifier[0].weight.detach())
    assert int(rg.state_dict()["net.base.0.1.num_batches_tracked"]) == 4  # both domains update the BN statistics
    out = capsys.readouterr().out
    assert "loss_x" in out and "loss_target" in out


def test_dann_generic_path_matches_fused_path(device):
    from src.nets import get_model, get_optimizer
    import torch.nn.functional as F
    from torch.nn.utils import clip_grad_norm_
    B = 4
    xs, ys = O.synth_batch(B, tag="src0"); xt, _ = O.synth_batch(B, tag="tgt0")
    sd0 = O.init_state(11, dann=True)
    finals = {}
    for path in ("generic", "fused"):
        cfg = _cfg(dann=True, optimizer="sgd", lr=0.05, weight_decay=5e-5)
        m = get_model(cfg); m.load_state_dict(sd0, strict=True); opt = get_optimizer(cfg, m); m = m.to(device).train()
        if path == "generic":
            opt.zero_grad(set_to_none=True)
            (lp, sm), ds = m(xs.to(device), y=ys.to(device), alpha=0.3)
            ls = F.binary_cross_entropy_with_logits(ds, torch.ones(B, device=device))
            _, dt = m(xt.to(device), alpha=0.3)
            lt = F.binary_cross_entropy_with_logits(dt, torch.zeros(B, device=device))
            (lp + ls + lt).backward()
            clip_grad_norm_(m.parameters(), 1.0)
            opt.step()
            losses = [float(lp), float(ls), float(lt)]
        else:
            s = opt.train_step(xs.to(device), ys.to(device), target_images=xt.to(device), alpha=0.3).tolist()
            losses = [s[0], s[3], s[4]]
        finals[path] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    np.testing.assert_allclose(finals["generic"][0], finals["fused"][0], rtol=2e-4)
    for k in ("domain_classifier.0.weight", "domain_classifier.3.weight", "net.base.17.conv.3.weight", "net.head.0.bias"):
        d_g = finals["generic"][1][k].double() - sd0[k].double()
        d_f = finals["fused"][1][k].double() - sd0[k].double()
        assert _rel(d_g, d_f) < 0.05, k


def test_style_augmentation_lookahead_keeps_the_batches(device):
    """AugLookahead (the trainer's one-batch-ahead restyling on a side stream) yields every batch once, in order, restyled
    exactly where the coin says so, with targets untouched"""
    import torch
    from speedplusbaseline_amd.core.trainer import AugLookahead

    class Aug:
        def __call__(self, x):
            return x * 0.5 + 0.25

    batches = [(torch.full((2, 3, 8, 8), float(i)), torch.full((2, 2, 11), float(i))) for i in range(5)]
    coins = [True, False, True, True, False]
    out = list(AugLookahead(batches, device, Aug(), lambda i: coins[i]))
    torch.cuda.synchronize()
    assert len(out) == 5
    for i, (x, y) in enumerate(out):
        want = i * 0.5 + 0.25 if coins[i] else float(i)
        assert x.is_cuda and float(x.mean()) == want and float(y.mean()) == float(i)


def test_fp16_module_generic_path_with_a_torch_gradscaler(device):
    """KeypointRegressionNet(precision="fp16") the way the reference's loop uses autocast + GradScaler (trainer.py:73-98): scaler.scale(loss)
    .backward() through the module (the upstream gradient carries the caller's scale; the engine's device-side scale stays out of it),
    scaler.unscale_, clip_grad_norm_, scaler.step(optimizer) with a plain torch optimizer.  Gradients finite and close to the fp32 module's."""
    from src.nets.park2019 import KeypointRegressionNet
    from torch.nn.utils import clip_grad_norm_
    x, y = O.synth_batch(4)
    sd0 = O.init_state(11)
    grads = {}
    for prec in ("fp32", "fp16"):
        model = KeypointRegressionNet(11, precision=prec)
        model.load_state_dict(sd0, strict=True)
        model = model.to(device).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.01)
        scaler = torch.amp.GradScaler("cuda", init_scale=256.0, enabled=(prec == "fp16"))
        loss, sm = model(x.to(device), y.to(device))
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        gn = clip_grad_norm_(model.parameters(), 1.0)
        p_before = model.head[0].bias.detach().clone()
        scaler.step(opt)
        scaler.update()
        assert torch.isfinite(gn) and not torch.equal(model.head[0].bias.detach(), p_before)      # the step was taken
        grads[prec] = (float(loss), torch.cat([p.grad.detach().flatten().double().cpu() for p in model.parameters()]))
    l32, g32 = grads["fp32"]; l16, g16 = grads["fp16"]
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    print("loss fp32 %.5f fp16 %.5f; clipped gradient cosine %.4f, norm ratio %.4f" % (l32, l16, cos, float(g16.norm() / g32.norm())))
    # random init amplifies rounding ~350x (tests/test_krn_gpu.py) and the float-atomic statistics make it run-to-run: float16 lands 1-5 % from
    # float32 here (bfloat16: 5-15 %); the trained-state bars are tests/test_parity_conditioned_gpu.py
    assert abs(l16 - l32) <= 0.08 * l32 and cos > 0.5 and 0.8 < float(g16.norm() / g32.norm()) < 1.25


def test_deterministic_module_two_runs_bit_identical(device):
    """KeypointRegressionNet(deterministic=True) through the fused optimizer path: two runs of three AdamW steps are bit-identical"""
    from src.nets import get_model, get_optimizer
    x, y = O.synth_batch(4)
    outs = []
    for rep in range(2):
        cfg = _cfg(optimizer="adamw", lr=1e-3, weight_decay=0.01)
        cfg.deterministic = True
        model = get_model(cfg)
        model.load_state_dict(O.init_state(11), strict=True)
        opt = get_optimizer(cfg, model)
        model = model.to(device).train()
        for _ in range(3):
            s = opt.train_step(x.to(device), y.to(device))
        torch.cuda.synchronize()
        assert model.engine().deterministic and model.engine().det_misses() == 0
        outs.append((s.cpu().clone(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])


def test_deterministic_module_generic_autograd_path(device):
    """KeypointRegressionNet / RevGrad(deterministic=True) through loss.backward() + a torch optimizer (the generic path
    ran backward into an arena the reproducible library had no shadow for -> SPB_E_STATE): two runs are bit-identical, every float
    atomic lands in an exact region, and the step agrees with the float-atomic library's."""
    from src.nets import get_model
    from torch.nn.utils import clip_grad_norm_
    x, y = O.synth_batch(4)
    sd0 = O.init_state(11)

    def run(det, dann=False, steps=2):
        cfg = _cfg(optimizer="sgd", lr=0.05, weight_decay=5e-5, dann=dann)
        cfg.deterministic = det
        torch.manual_seed(7)                # (the domain classifier is not part of sd0: its random init must be the same in every run)
        model = get_model(cfg)
        if dann:
            model.net.load_state_dict(sd0, strict=True)
        else:
            model.load_state_dict(sd0, strict=True)
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        model = model.to(device).train()
        for _ in range(steps):
            if dann:
                (loss, _), dom = model(x.to(device), y.to(device), alpha=0.3)
                loss = loss + torch.nn.functional.binary_cross_entropy_with_logits(dom, torch.ones_like(dom))
            else:
                loss, _ = model(x.to(device), y.to(device))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
        torch.cuda.synchronize()
        if det:
            assert model.engine().deterministic and model.engine().det_misses() == 0
        return float(loss), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    for dann in (False, True):
        a, b = run(True, dann), run(True, dann)
        assert a[0] == b[0] and all(torch.equal(a[1][k], b[1][k]) for k in a[1])
        a, c = run(True, dann, steps=1), run(False, dann, steps=1)     # one step: same forward, gradients up to the float atomics' order
        assert abs(a[0] - c[0]) <= 2e-4 * abs(c[0]) + 1e-6            # (float-atomic batch statistics: ~2e-5 run to run)
        for k in ("base.0.1.weight", "extras.3.conv.4.weight", "head.0.weight"):
            kk = ("net." + k) if dann else k
            assert _rel(a[1][kk] - sd0[k], c[1][kk] - sd0[k]) < 0.05, kk
