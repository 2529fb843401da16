"""Generates tests/golden/spn_golden.npz by IMPORTING the reference's own src/nets/spn.py (read-only at /root/reference;
no stubs needed) with pretrain=False.  Only data travels: weights/inputs come from the portable RNG recipe, outputs are
stored as arrays.  num_classes=64 keeps the fixture small (the class count only sizes fc8/fc11).

Run:  python tests/golden/make_golden_spn.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import spn_oracle as S  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_spn", "/root/reference/src/nets/spn.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def caffe_weights(out):
    """spn.py:104-123 run on a synthetic AlexNet file: per-tensor digests + a crop of what the reference's loader leaves"""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "bvlc_alexnet.npy")
        S.caffe_file(path)
        net = ref.SpacecraftPoseNet(64, keep_prob=0.5, pretrain=False)
        net.load_state_dict(S.init_state(64), strict=True)
        net.load_weights(path)
    for k, v in net.state_dict().items():
        out["caffe_sum/" + k] = np.array(S.checksum(v))
    out["caffe_conv2_crop"] = net.conv2.weight[:3, :4].detach().numpy().copy()


def full_size(out):
    """BASELINE configs[5] sizes: 5000 classes, batch 32 at 227x227.  Logit digests, a crop, the three reductions of the
    soft-target cross-entropy and the trainer's loss; digests of the gradients of the reference loss (eval mode: dropout
    is the identity) for every parameter."""
    NC, B = 5000, 32
    net = ref.SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False)
    net.load_state_dict(S.init_state(NC), strict=True)
    x, yc, yw = S.synth_batch(B, NC, seed=23)
    net.eval()
    c, r = net(x)
    lc = ref.softmax_cross_entropy_with_logits(c, yc, "mean"); lr = ref.softmax_cross_entropy_with_logits(r, yw, "mean")
    loss = lc + 10.0 * lr
    loss.backward()
    out["full_c_sum"] = np.array(S.checksum(c)); out["full_r_sum"] = np.array(S.checksum(r))
    out["full_c_crop"] = c[:4, :8].detach().numpy().copy(); out["full_r_crop"] = r[-4:, -8:].detach().numpy().copy()
    out["full_losses"] = np.array([float(loss), float(lc), float(lr)])
    out["full_loss_none"] = ref.softmax_cross_entropy_with_logits(r, yw, "none").detach().numpy().copy()
    for k, p in net.named_parameters():
        out["full_grad_sum/" + k] = np.array(S.checksum(p.grad))
    out["full_grad_fc11_crop"] = net.fc11.weight.grad[:6, :6].numpy().copy()
    for k, p in net.named_parameters():      # 64 scattered elements of every gradient (index (j * 7919) % n): layout check
        g = p.grad.flatten()
        idx = (torch.arange(64, dtype=torch.int64) * 7919) % g.numel()
        out["full_grad_samples/" + k] = g[idx].numpy().copy()
    print("full size: loss %.6f (class %.6f, regress %.6f)" % (float(loss), float(lc), float(lr)))


def main():
    NC = 64
    net = ref.SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False)
    sd = S.init_state(NC)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    out = {"keys": np.array(list(sd.keys())), "n_params_5000": np.int64(sum(int(np.prod(s)) for s in S.param_shapes(5000).values()))}
    x, yc, yw = S.synth_batch(2, NC, seed=11)
    net.eval()
    feats = {}
    hooks = [getattr(net, n).register_forward_hook(lambda m, a, o, n=n: feats.__setitem__(n, o.detach().clone()))
             for n in ("norm1", "norm2", "pool5", "conv3")]
    with torch.no_grad():
        c, r = net(x)
    for h in hooks:
        h.remove()
    out["eval_c"] = c.numpy(); out["eval_r"] = r.numpy()
    for n in ("norm1", "norm2", "pool5"):
        out["eval_%s_sum" % n] = np.array(S.checksum(feats[n]))
    out["eval_norm1_crop"] = feats["norm1"][:, :6, :5, :5].numpy()
    out["loss_mean"] = np.float64(ref.softmax_cross_entropy_with_logits(c, yc, "mean"))
    out["loss_sum"] = np.float64(ref.softmax_cross_entropy_with_logits(r, yw, "sum"))
    out["loss_none"] = ref.softmax_cross_entropy_with_logits(c, yw, "none").numpy()
    # training-mode gradient with dropout disabled through p=0 modules is not the reference's path; instead pin the
    # eval-mode gradient (dropout identity) of the reference loss assembly (trainer.py:160-165)
    net.zero_grad()
    c, r = net(x)
    loss = ref.softmax_cross_entropy_with_logits(c, yc, "mean") + 10.0 * ref.softmax_cross_entropy_with_logits(r, yw, "mean")
    loss.backward()
    out["grad_loss"] = np.float64(loss.detach())
    for k, p in net.named_parameters():
        out["grad_sum/" + k] = np.array(S.checksum(p.grad))
    out["grad_conv1_crop"] = net.conv1.weight.grad[:4, :, :3, :3].numpy()
    out["grad_fc8_crop"] = net.fc8.weight.grad[:8, :16].numpy()
    full_size(out)
    caffe_weights(out)
    np.savez_compressed(os.path.join(HERE, "spn_golden.npz"), **out)
    print("wrote spn_golden.npz", out["n_params_5000"], float(out["grad_loss"]))


if __name__ == "__main__":
    main()
