"""Whole-network parity on the MI355X: the HIP plan (C-ABI spb_krn_*) against the CPU oracle and the golden vectors
captured from the reference (tests/golden/krn_golden.npz).

f32 mode pins indexing/semantics (tolerances ~1e-4); bf16 mode is the shipped compute type and is held to the
north-star bar: keypoint MSE vs the fp32 reference <= 1e-4.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import krn_oracle as O  # noqa: E402
from speedplusbaseline_amd.engine import KrnEngine  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "krn_golden.npz"), allow_pickle=False)


def load_state(eng, sd):
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].detach().to(eng.device))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].detach().flatten().to(eng.device))
    for i, n in enumerate(eng.bn_names):
        eng.nbt[i] = int(sd[n])


def relerr(a, b):
    a = torch.as_tensor(a).double().cpu().flatten(); b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def oracle_run():
    """CPU oracle: train-mode loss + all gradients, eval-mode keypoints, B=4"""
    x, y = O.synth_batch(4)
    sd = O.init_state(11)
    with torch.no_grad():
        xc, yc = O.krn_forward(sd, x, None, training=False)
    sd = O.init_state(11)
    names = O._leafify(sd)
    loss, lx, ly = O.krn_forward(sd, x, y, training=True)
    loss.backward()
    return dict(x=x, y=y, xc=xc, yc=yc, loss=float(loss), lx=float(lx), ly=float(ly),
                grads={k: sd[k].grad.clone() for k in names}, sd_after=sd)


@pytest.mark.parametrize("prec,tol_pred,tol_loss,tol_grad", [("fp32", 2e-4, 1e-4, 2e-3), ("bf16", None, 5e-2, 0.25)])
def test_krn_forward_backward_vs_oracle(device, oracle_run, prec, tol_pred, tol_loss, tol_grad):
    o = oracle_run
    eng = KrnEngine(11).attach(device, prec)
    load_state(eng, O.init_state(11))
    x, y = o["x"].to(device), o["y"].to(device)
    # eval: predicted keypoints (config 1 of BASELINE.json, CPU reference vs GPU)
    pred, _, _ = eng.forward(x, None, training=False)
    torch.cuda.synchronize()
    xc, yc = pred[:, 0::2].cpu(), pred[:, 1::2].cpu()
    mse = float(((xc - o["xc"]) ** 2).mean() + ((yc - o["yc"]) ** 2).mean()) / 2
    assert mse <= 1e-4, mse  # north-star: keypoint MSE within 1e-4 of the reference
    if tol_pred is not None:
        assert relerr(xc, o["xc"]) < tol_pred and relerr(yc, o["yc"]) < tol_pred
        assert relerr(xc, G["g4_eval_xc"]) < tol_pred
    # train: loss, running stats, gradients
    eng.grads.zero_()
    pred, scal, _ = eng.forward(x, y, training=True)
    eng.backward(4)
    torch.cuda.synchronize()
    s = scal.cpu().numpy()
    assert abs(s[0] - o["loss"]) <= tol_loss * o["loss"], (s, o["loss"])
    assert abs(s[1] - o["lx"]) <= tol_loss * o["lx"] and abs(s[2] - o["ly"]) <= tol_loss * o["ly"]
    assert abs(s[0] - G["g4_train_loss"][0]) <= tol_loss * G["g4_train_loss"][0]
    sd_after = o["sd_after"]
    for name, shape, off, numel in eng.buffer_infos[:6] + eng.buffer_infos[-4:]:
        assert relerr(eng.buffers[off: off + numel], sd_after[name]) < (1e-4 if prec == "fp32" else 2e-2), name
    assert int(eng.nbt[0]) == 1 and int(eng.nbt[-1]) == 1
    worst = (0.0, None)
    gn2 = 0.0
    for info in eng.param_infos:
        g = eng.param_view(info, eng.grads)
        ref = o["grads"][info[0]]
        gn2 += float(g.double().pow(2).sum())
        e = relerr(g, ref)
        if e > worst[0]:
            worst = (e, info[0])
    assert worst[0] < tol_grad, worst
    gn_ref = float(G["g4_grad_norm"][0])
    assert abs(gn2 ** 0.5 - gn_ref) <= (1e-3 if prec == "fp32" else 5e-2) * gn_ref


def test_krn_train_steps_match_reference_trainer(device):
    """2 AdamW steps in the reference trainer's order (trainer.py:72-98) vs golden per-iteration losses + final state"""
    from speedplusbaseline_amd import ops
    eng = KrnEngine(11).attach(device, "fp32")
    load_state(eng, O.init_state(11))
    m = torch.zeros_like(eng.params); v = torch.zeros_like(eng.params)
    sq = torch.zeros(1, device=device)
    got = []
    for i in range(2):
        x, y = O.synth_batch(4, tag="it%d" % i)
        _, scal, _ = eng.forward(x.to(device), y.to(device), training=True)
        eng.grads.zero_()
        eng.backward(4)
        ops.grad_sqnorm(eng.grads, sq)
        ops.optim_step("adamw", eng.params, eng.grads, m=m, v=v, sqnorm=sq, lr=1e-4, beta1=0.9, beta2=0.999,
                       weight_decay=0.01, max_norm=1.0, step=i + 1)
        got.append(scal.cpu().numpy().copy())
    torch.cuda.synchronize()
    got = np.array(got)
    assert np.abs(got - G["g7_losses"]).max() <= 2e-3 * np.abs(G["g7_losses"]).max(), (got, G["g7_losses"])
    keys = [str(k) for k in G["g7_keys"]]
    cs = dict(zip(keys, G["g7_checksums"]))
    for info in eng.param_infos:
        t = eng.param_view(info).double()
        ref = cs[info[0]]
        assert abs(float(t.sum()) - ref[0]) <= 1e-3 * (abs(ref[0]) + 1e-2 * ref[1] ** 0.5 + 1e-6), info[0]
        assert abs(float((t * t).sum()) - ref[1]) <= 1e-3 * ref[1] + 1e-9, info[0]


def test_revgrad_forward_and_dann_step(device):
    """RevGrad forward (pose loss + domain logits) and one DANN step's gradients vs the CPU oracle (dann.py:68-100)"""
    import torch.nn.functional as F
    B = 4
    xs, ys = O.synth_batch(B, tag="src0"); xt, _ = O.synth_batch(B, tag="tgt0")
    alpha = 0.3
    sd = O.init_state(11, dann=True)
    names = O._leafify(sd)
    (lp, lx, ly), ds = O.revgrad_forward(sd, xs, ys, alpha, True)
    l_src = F.binary_cross_entropy_with_logits(ds, torch.ones(B))
    _, dt_ = O.revgrad_forward(sd, xt, None, alpha, True)
    l_tgt = F.binary_cross_entropy_with_logits(dt_, torch.zeros(B))
    (lp + l_src + l_tgt).backward()

    eng = KrnEngine(11, dann=True).attach(device, "fp32")
    load_state(eng, O.init_state(11, dann=True))
    eng.grads.zero_()
    _, scal, dom_s = eng.forward(xs.to(device), ys.to(device), training=True, slot=0, domain=True)
    loss_s, dl_s = eng.bce_logits(dom_s, 1.0)
    _, _, dom_t = eng.forward(xt.to(device), None, training=True, slot=1, domain=True)
    loss_t, dl_t = eng.bce_logits(dom_t, 0.0)
    eng.backward(B, slot=0, with_pose=True, dlogit=dl_s, alpha=alpha)
    eng.backward(B, slot=1, with_pose=False, dlogit=dl_t, alpha=alpha)
    torch.cuda.synchronize()
    assert relerr(dom_s, ds.detach()) < 1e-3 and relerr(dom_t, dt_.detach()) < 1e-3
    assert relerr(dom_s, G["g6_dom"]) < 1e-3
    assert abs(float(scal[0]) - float(lp)) < 1e-4 * float(lp)
    assert abs(float(loss_s) - float(l_src)) < 1e-4 and abs(float(loss_t) - float(l_tgt)) < 1e-4
    worst = (0.0, None)
    for info in eng.param_infos:
        e = relerr(eng.param_view(info, eng.grads), sd[info[0]].grad)
        if e > worst[0]:
            worst = (e, info[0])
    assert worst[0] < 5e-3, worst
    # BN running statistics were updated by BOTH domains, twice tracked (dann.py:81,89)
    assert int(eng.nbt[0]) == 2
    name, shape, off, numel = eng.buffer_infos[0]
    assert relerr(eng.buffers[off: off + numel], sd[name]) < 1e-4
