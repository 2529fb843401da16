"""CPU: the SPN oracle (oracle/spn_oracle.py) against golden vectors produced by the reference's own spn.py
(tests/golden/make_golden_spn.py).  No GPU, no reference at run time."""
import os

import numpy as np
import torch

from oracle import spn_oracle as S

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "spn_golden.npz"))
NC = 64


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_layout_and_param_count():
    assert list(S.init_state(NC).keys()) == list(GOLD["keys"])
    assert int(GOLD["n_params_5000"]) == 152372368


def test_eval_forward_and_losses_match_reference():
    sd = S.init_state(NC)
    x, yc, yw = S.synth_batch(2, NC, seed=11)
    feats = {}
    with torch.no_grad():
        c, r = S.forward(sd, x, None, collect=feats)
    assert rel(c.numpy(), GOLD["eval_c"]) < 2e-5 and rel(r.numpy(), GOLD["eval_r"]) < 2e-5
    for n in ("norm1", "norm2", "pool5"):
        assert rel(S.checksum(feats[n]), GOLD["eval_%s_sum" % n]) < 1e-5
    assert rel(feats["norm1"][:, :6, :5, :5].numpy(), GOLD["eval_norm1_crop"]) < 1e-5
    assert abs(float(S.softmax_cross_entropy_with_logits(c, yc, "mean")) - float(GOLD["loss_mean"])) < 1e-5
    assert abs(float(S.softmax_cross_entropy_with_logits(r, yw, "sum")) - float(GOLD["loss_sum"])) < 1e-4
    assert rel(S.softmax_cross_entropy_with_logits(c, yw, "none").numpy(), GOLD["loss_none"]) < 1e-5


def test_gradients_match_reference():
    sd = S.init_state(NC)
    x, yc, yw = S.synth_batch(2, NC, seed=11)
    g = S.train_grads(sd, x, yc, yw, masks=None)
    assert abs(g["loss"] - float(GOLD["grad_loss"])) < 1e-4 * abs(float(GOLD["grad_loss"]))
    for k, v in g["grads"].items():
        assert rel(S.checksum(v), GOLD["grad_sum/" + k]) < 2e-4, k
    assert rel(g["grads"]["conv1.weight"][:4, :, :3, :3].numpy(), GOLD["grad_conv1_crop"]) < 1e-4
    assert rel(g["grads"]["fc8.weight"][:8, :16].numpy(), GOLD["grad_fc8_crop"]) < 1e-4


def test_dropout_masks_are_explicit():
    sd = S.init_state(NC)
    x, yc, yw = S.synth_batch(2, NC, seed=11)
    m = S.synth_masks(2)
    with torch.no_grad():
        c1, _ = S.forward(sd, x, m)
        c2, _ = S.forward(sd, x, m)
        c0, _ = S.forward(sd, x, None)
    assert torch.equal(c1, c2) and not torch.allclose(c1, c0)
