"""Generates tests/golden/ghiasi_golden.npz by IMPORTING the reference's own src/styleaug/ghiasi.py (read-only at
/root/reference; it needs no stubs).  Only data travels: inputs and weights are regenerated from the portable RNG recipe
(oracle/portable_rng.py), outputs are stored as arrays.

Run:  python tests/golden/make_golden_ghiasi.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ghiasi_oracle as G  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_ghiasi", "/root/reference/src/styleaug/ghiasi.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def main():
    torch.manual_seed(0)
    net = ref.Ghiasi().eval()
    sd = G.init_state()
    assert list(net.state_dict().keys()) == list(sd.keys())
    assert all(tuple(v.shape) == tuple(sd[k].shape) for k, v in net.state_dict().items())
    net.load_state_dict(sd, strict=True)
    out = {"n_params": np.int64(sum(v.numel() for v in net.state_dict().values())), "keys": np.array(list(sd.keys())),
           "n_params_attr": np.int64(net.n_params)}
    with torch.no_grad():
        for tag, B, hw in (("a", 2, 64), ("b", 1, 96), ("c", 1, 224)):      # c: the training resolution
            x, s = G.synth_inputs(B, hw, seed=2021 + B)
            feats = {}
            hooks = [net.layers[i].register_forward_hook(lambda m, a, o, i=i: feats.__setitem__(i, o.detach().clone()))
                     for i in range(11)]
            y = net(x, s)
            for h in hooks:
                h.remove()
            out[tag + "_out_crop"] = y[:, :, :16, :16].numpy()
            out[tag + "_out_sum"] = np.array(G.checksum(y))
            for i in (0, 2, 3, 7, 8, 10):
                out["%s_l%d_sum" % (tag, i)] = np.array(G.checksum(feats[i]))
            out[tag + "_l7_crop"] = feats[7][:, :8, :6, :6].numpy()
    # per-block pins: ConvInRelu, ResidualBlock, UpsampleConvInRelu on small tensors
    x, s = G.synth_inputs(2, 24, seed=7)
    with torch.no_grad():
        out["blk_convinrelu"] = net.layers[1](net.layers[0](x)).numpy()[:, :4]
        h = torch.from_numpy(G.prng.uniform("blk/h", (2, 128, 12, 12), -1.0, 1.0, 7))
        out["blk_residual"] = net.layers[3](h, s).numpy()[:, :4]
        out["blk_upsample"] = net.layers[8](h, s).numpy()[:, :4]
    # embedding sampler algebra (styleAugmentor.py:36-46) on a synthetic SPD covariance
    Q = G.prng.uniform("emb/q", (100, 100), -1.0, 1.0, 3).astype(np.float64)
    cov = Q @ Q.T / 100.0 + 0.05 * np.eye(100)
    u, sv, _ = np.linalg.svd(cov)
    A = torch.tensor(np.matmul(u, np.diag(sv ** 0.5))).float()
    z = torch.from_numpy(G.prng.normalish("emb/z", (5, 100), 1.0, 3))
    mean = torch.from_numpy(G.prng.uniform("emb/mean", (1, 100), -0.5, 0.5, 3))
    base = torch.from_numpy(G.prng.uniform("emb/base", (100,), -0.5, 0.5, 3))
    emb = torch.mm(z, A.transpose(1, 0)) + mean
    out["emb_sample"] = emb.numpy()
    out["emb_restyle"] = (0.5 * emb + (1 - 0.5) * base).numpy()
    out["emb_cov"] = cov
    np.savez_compressed(os.path.join(HERE, "ghiasi_golden.npz"), **out)
    print("wrote ghiasi_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
