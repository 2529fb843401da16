"""SPN surface pieces that run without a GPU: the AlexNet weight import (spn.py:104-123) against what the reference's own
loader produced from the same file, and checkpoint resume in the reference's order -- get_optimizer, load_checkpoint,
THEN model.to(device) (train.py:84-97; utils.py:121-135) -- for both fused optimizers while the model is still on the CPU."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import spn_oracle as S
from speedplusbaseline_amd.nets import get_optimizer
from speedplusbaseline_amd.nets.park2019 import KeypointRegressionNet
from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet
from speedplusbaseline_amd.utils import load_checkpoint, save_checkpoint

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "spn_golden.npz"))


def test_load_weights_matches_the_references_loader(tmp_path):
    path = str(tmp_path / "bvlc_alexnet.npy")
    S.caffe_file(path)
    net = SpacecraftPoseNet(64, pretrain=False)
    init = S.init_state(64)
    net.load_state_dict(init, strict=True)
    v0 = net._version
    net.load_weights(path)
    assert net._version > v0                                   # compute copies / bf16 shadow are marked stale
    for k, v in net.state_dict().items():
        got, want = np.array(S.checksum(v)), GOLD["caffe_sum/" + k]
        assert np.allclose(got, want, rtol=1e-6, atol=1e-6), k
    assert np.array_equal(net.conv2.weight[:3, :4].detach().numpy(), GOLD["caffe_conv2_crop"])
    blob = np.load(path, allow_pickle=True, encoding="bytes").item()
    assert np.array_equal(net.conv4.weight.detach().numpy(), np.transpose(blob["conv4"][0], (3, 2, 0, 1)))   # HWIO -> OIHW
    assert np.array_equal(net.fc6.weight.detach().numpy(), init["fc6.weight"].numpy())                        # fc layers untouched
    bad = dict(blob); bad["conv3"] = [blob["conv3"][0][:, :, :100], blob["conv3"][1]]
    np.save(path, bad, allow_pickle=True)
    with pytest.raises(ValueError):
        net.load_weights(path)


def test_pretrain_true_reads_the_cwd_relative_file(tmp_path, monkeypatch):
    """get_model hard-wires pretrain=True and the path is relative to the working directory (build.py:48, spn.py:102)"""
    os.makedirs(tmp_path / "checkpoints" / "pretrained")
    S.caffe_file(str(tmp_path / "checkpoints" / "pretrained" / "bvlc_alexnet.npy"))
    monkeypatch.chdir(tmp_path)
    net = SpacecraftPoseNet(64)                                  # pretrain=True
    assert np.allclose(np.array(S.checksum(net.conv1.weight)), GOLD["caffe_sum/conv1.weight"], rtol=1e-6)
    monkeypatch.chdir(tmp_path / "checkpoints")
    with pytest.raises(FileNotFoundError):
        SpacecraftPoseNet(64)


def _cfg(**kw):
    c = types.SimpleNamespace(model_name="spn", num_keypoints=11, num_classes=64, dann=False, optimizer="adamw", lr=1e-3, momentum=0.9,
                              weight_decay=0.01, fp16=False, precision="bf16", synthetic_batches=1)
    c.__dict__.update(kw)
    return c


def test_spn_checkpoint_resume_with_the_model_still_on_the_cpu(tmp_path):
    cfg = _cfg()
    model = SpacecraftPoseNet(64, pretrain=False)
    opt = get_optimizer(cfg, model)
    n = sum((p.numel() + 7) // 8 * 8 for p in model.parameters())
    opt._t, opt._m, opt._v = 7, torch.arange(n, dtype=torch.float32) * 1e-6, torch.ones(n) * 0.25       # as after 7 steps
    save_checkpoint({"epoch": 3, "model": "spn", "state_dict": model.state_dict(), "best_score": 3, "optimizer": opt.state_dict()},
                    True, str(tmp_path))
    assert os.path.exists(tmp_path / "checkpoint.pth.tar") and os.path.exists(tmp_path / "model_best.pth.tar")
    model2 = SpacecraftPoseNet(64, pretrain=False)
    opt2 = get_optimizer(cfg, model2)
    epoch, best = load_checkpoint(str(tmp_path / "checkpoint.pth.tar"), model2, opt2, torch.device("cpu"))   # no GPU touched
    assert (epoch, best) == (3, 3) and opt2._t == 7
    assert torch.equal(opt2._m, opt._m) and torch.equal(opt2._v, opt._v)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    bare = torch.load(tmp_path / "model_best.pth.tar")
    assert list(bare.keys()) == list(model.state_dict().keys())
    # a checkpoint of a run that never stepped
    fresh = get_optimizer(cfg, model2).state_dict()
    opt2.load_state_dict(fresh)
    assert opt2._m is None and opt2._t == 0


def test_krn_checkpoint_resume_with_the_model_still_on_the_cpu(tmp_path):
    cfg = _cfg(model_name="krn")
    model = KeypointRegressionNet(11)
    opt = get_optimizer(cfg, model)
    sd = opt.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["lr"] == 1e-3
    g = torch.Generator().manual_seed(1)
    state = {i: {"step": torch.tensor(5.0), "exp_avg": torch.randn(p.shape, generator=g), "exp_avg_sq": torch.rand(p.shape, generator=g)}
             for i, p in enumerate(opt.param_groups[0]["params"])}
    sd = {"state": state, "param_groups": sd["param_groups"]}
    sd["param_groups"][0]["lr"] = 5e-4
    save_checkpoint({"epoch": 2, "model": "krn", "state_dict": model.state_dict(), "best_score": 2, "optimizer": sd}, False, str(tmp_path))
    model2 = KeypointRegressionNet(11)
    opt2 = get_optimizer(cfg, model2)
    epoch, _ = load_checkpoint(str(tmp_path / "checkpoint.pth.tar"), model2, opt2, torch.device("cpu"))
    assert epoch == 2 and opt2.param_groups[0]["lr"] == 5e-4
    assert opt2._pending_state is not None and len(opt2._pending_state) == len(state)      # applied when the step binds to the GPU arena
    assert not os.path.exists(tmp_path / "model_best.pth.tar")
