"""CPU: the Ghiasi oracle (oracle/ghiasi_oracle.py) against golden vectors produced by the reference's own ghiasi.py
(tests/golden/make_golden_ghiasi.py).  No GPU, no reference at run time."""
import os

import numpy as np
import torch

from oracle import ghiasi_oracle as G

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ghiasi_golden.npz"))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_state_dict_layout_matches_reference():
    sd = G.init_state()
    assert list(sd.keys()) == list(GOLD["keys"])
    assert sum(v.numel() for v in sd.values()) == int(GOLD["n_params"]) == 1954593
    assert int(GOLD["n_params_attr"]) == 5 * 512 + 128 + 64 + 6   # Ghiasi.n_params: style-dependent (gamma, beta) outputs


def test_decoder_forward_matches_reference():
    sd = G.init_state()
    for tag, B, hw in (("a", 2, 64), ("b", 1, 96), ("c", 1, 224)):       # c: the training resolution
        x, s = G.synth_inputs(B, hw, seed=2021 + B)
        feats = {}
        with torch.no_grad():
            y = G.forward(sd, x, s, collect=feats)
        assert rel(y[:, :, :16, :16].numpy(), GOLD[tag + "_out_crop"]) < 2e-5
        assert rel(G.checksum(y), GOLD[tag + "_out_sum"]) < 1e-5
        for i in (0, 2, 3, 7, 8, 10):
            assert rel(G.checksum(feats["l%d" % i]), GOLD["%s_l%d_sum" % (tag, i)]) < 2e-4, i
        assert rel(feats["l7"][:, :8, :6, :6].numpy(), GOLD[tag + "_l7_crop"]) < 1e-4
        assert float(y.min()) > 0.0 and float(y.max()) < 1.0


def test_blocks_match_reference():
    sd = G.init_state()
    x, s = G.synth_inputs(2, 24, seed=7)
    with torch.no_grad():
        f = {}
        G.forward(sd, torch.nn.functional.pad(x, (0, 0, 0, 0)), s, collect=f) if False else None
        import torch.nn.functional as F
        l0 = F.relu(F.instance_norm(G._conv(sd, "layers.0.conv", x, 9), eps=G.IN_EPS))
        l1 = F.relu(F.instance_norm(G._conv(sd, "layers.1.conv", l0, 3, 2), eps=G.IN_EPS))
        assert rel(l1[:, :4].numpy(), GOLD["blk_convinrelu"]) < 2e-5
        h = torch.from_numpy(G.prng.uniform("blk/h", (2, 128, 12, 12), -1.0, 1.0, 7))
        p = "layers.3."
        y = F.instance_norm(G._conv(sd, p + "conv1", h, 3), eps=G.IN_EPS)
        y = F.relu(G._fc(sd, p + "fc_gamma1", s) * y + G._fc(sd, p + "fc_beta1", s))
        y = F.instance_norm(G._conv(sd, p + "conv2", y, 3), eps=G.IN_EPS)
        y = G._fc(sd, p + "fc_gamma2", s) * y + G._fc(sd, p + "fc_beta2", s)
        assert rel((h + y)[:, :4].numpy(), GOLD["blk_residual"]) < 2e-5
        p = "layers.8."
        u = F.interpolate(h, scale_factor=2, mode="nearest")
        u = F.instance_norm(G._conv(sd, p + "conv", u, 3), eps=G.IN_EPS)
        u = F.relu(G._fc(sd, p + "fc_gamma", s) * u + G._fc(sd, p + "fc_beta", s))
        assert rel(u[:, :4].numpy(), GOLD["blk_upsample"]) < 2e-5


def test_embedding_sampler_algebra():
    A = G.embedding_A(GOLD["emb_cov"])
    z = torch.from_numpy(G.prng.normalish("emb/z", (5, 100), 1.0, 3))
    mean = torch.from_numpy(G.prng.uniform("emb/mean", (1, 100), -0.5, 0.5, 3))
    base = torch.from_numpy(G.prng.uniform("emb/base", (100,), -0.5, 0.5, 3))
    assert rel(G.sample_embedding(z, A, mean).numpy(), GOLD["emb_sample"]) < 1e-5
    assert rel(G.restyle_embedding(z, A, mean, base, 0.5).numpy(), GOLD["emb_restyle"]) < 1e-5
    # A A^T reproduces the covariance
    assert rel((A.double() @ A.double().t()).numpy(), GOLD["emb_cov"]) < 1e-5


def test_upsample_conv_phase_weights_identity():
    """nearest x2 upsampling + ReflectionPad2d(1) + 3x3 convolution == four 2x2 convolutions on the replicate-padded low-resolution
    input with the summed weights of styleaug._phase_weights (what csrc/ghiasi.hip gconv_up2_kernel computes)"""
    import torch
    import torch.nn.functional as F
    from speedplusbaseline_amd.styleaug import _phase_weights
    torch.manual_seed(3)
    w = torch.randn(6, 8, 3, 3)
    x = torch.randn(2, 8, 5, 7, dtype=torch.float64)
    wp = _phase_weights(w)
    assert tuple(wp.shape) == (4, 6, 4, 8) and wp.dtype == torch.float32     # (the caller rounds to its storage format: bf16 or IEEE half)
    wp = wp.to(torch.bfloat16)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="reflect"), w.double())
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    out = torch.zeros_like(ref)
    for py in (0, 1):
        for px in (0, 1):
            k = wp[py * 2 + px].double().permute(0, 2, 1).reshape(6, 8, 2, 2)      # [Cout][tap][Cin] -> [Cout][Cin][ty][tx]
            out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + 6, px:px + 8], k)
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-2          # one bf16 rounding of the summed weights
