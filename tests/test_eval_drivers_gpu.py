"""valid_krn / valid_spn (reference src/core/inference.py:43-142,146-196) IN PROCESS on the MI355X: the real models' HIP forward
runs for every batch of SyntheticEvalLoader; once with its own (random-weight) output -- finite metrics, the four result files --
and once with the output replaced by the ground truth in the network's format AFTER the forward ran, which the post-processing
must turn into zero error (the CPU twin of these known-answer checks is tests/test_eval_drivers_cpu.py)."""
import os
import types

import numpy as np
import pytest
import torch

from speedplusbaseline_amd import pose
from speedplusbaseline_amd.core.inference import valid_krn, valid_spn
from speedplusbaseline_amd.data import SyntheticEvalLoader, synthetic_eval_assets

pytestmark = pytest.mark.gpu


def _krn(device, precision):
    from oracle import krn_oracle as O
    from speedplusbaseline_amd.nets import get_model
    cfg = types.SimpleNamespace(model_name="krn", num_keypoints=11, num_classes=5000, dann=False, optimizer="adamw", lr=1e-4,
                                momentum=0.9, weight_decay=0.01, fp16=False, precision=precision, max_epochs=1, texture_ratio=0.5)
    m = get_model(cfg)
    m.load_state_dict(O.init_state(11), strict=True)
    return m.to(device)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_valid_krn_in_process(device, tmp_path, precision):
    corners3D, cameraMatrix, distCoeffs, _ = synthetic_eval_assets(11, 64, 2021)
    batches = list(SyntheticEvalLoader(4, 3, corners3D, cameraMatrix, distCoeffs, hw=(224, 224), seed=11))
    model = _krn(device, precision)
    cfg = types.SimpleNamespace(logdir=str(tmp_path / "raw"))
    perf = valid_krn(0, cfg, model, batches, cameraMatrix, distCoeffs, corners3D, None, device)
    assert perf['eR'].count == 12 and all(np.isfinite(perf[k].avg) for k in perf)
    assert 0.0 <= perf['eR'].avg <= 180.0 and perf['speed (raw)'].avg > 0.0     # random weights: some pose, a bad one
    for fn in ('err_q.txt', 'err_t.txt', 'speed_raw.txt', 'speed_mod.txt'):
        assert len(open(os.path.join(cfg.logdir, fn)).read().split()) == 12

    calls = []

    class GtAfterForward(torch.nn.Module):
        """runs the real forward, checks its output, hands the ground truth on"""
        def __init__(self, net): super().__init__(); self.net = net
        def forward(self, images):
            x, y = self.net(images)
            assert x.shape == (images.shape[0], 11) and torch.isfinite(x).all() and torch.isfinite(y).all()
            _, bbox, q, t = batches[len(calls)]; calls.append(1)
            px = pose.project_keypoints(q.double().numpy(), t.double().numpy(), cameraMatrix, distCoeffs, corners3D)
            b = bbox.double().numpy()
            return (torch.from_numpy((px[:, 0] - b[:, 0:1]) / (b[:, 1:2] - b[:, 0:1])).float(),
                    torch.from_numpy((px[:, 1] - b[:, 2:3]) / (b[:, 3:4] - b[:, 2:3])).float())
    cfg2 = types.SimpleNamespace(logdir=str(tmp_path / "gt"))
    perf = valid_krn(0, cfg2, GtAfterForward(model), batches, cameraMatrix, distCoeffs, corners3D, None, device)
    assert len(calls) == 3
    assert perf['eR'].avg < 0.05 and perf['eT'].avg < 1e-3 and perf['speed (raw)'].avg < 1e-3 and perf['speed (thr)'].avg == 0.0
    assert all(abs(float(v)) < 0.1 for v in open(os.path.join(cfg2.logdir, 'err_q.txt')).read().split())


def test_valid_spn_in_process(device):
    from oracle import spn_oracle as S
    from speedplusbaseline_amd.nets import get_model
    NC = 64
    cfg = types.SimpleNamespace(model_name="spn", num_keypoints=11, num_classes=NC, dann=False, optimizer="adamw", lr=1e-3, momentum=0.9,
                                weight_decay=0.01, fp16=False, precision="bf16", num_neighbors=5, synthetic_batches=1)   # synthetic run: no AlexNet npy
    model = get_model(cfg)
    model.load_state_dict(S.init_state(NC), strict=True)
    model = model.to(device)
    corners3D, cameraMatrix, distCoeffs, qClass = synthetic_eval_assets(11, NC, 2021)
    batches = list(SyntheticEvalLoader(4, 2, corners3D, cameraMatrix, distCoeffs, hw=(227, 227), seed=12))
    perf = valid_spn(0, cfg, model, batches, cameraMatrix, distCoeffs, corners3D, None, device, qClass)
    assert perf['eR'].count == 8 and all(np.isfinite(perf[k].avg) for k in perf) and 0.0 <= perf['eR'].avg <= 180.0

    # known answer: ground-truth attitudes ARE classes; the real forward runs, its regression output is replaced by a one-hot
    g = np.random.default_rng(3)
    kb = []
    for _ in range(2):
        idx = g.integers(0, NC, size=4)
        q = qClass[idx].astype(np.float64)
        t = g.random((4, 3)) * np.array([0.6, 0.4, 6.0]) + np.array([-0.3, -0.2, 5.0])
        px = pose.project_keypoints(q, t, cameraMatrix, distCoeffs, corners3D)
        bbox = np.stack([px[:, 0].min(1), px[:, 0].max(1), px[:, 1].min(1), px[:, 1].max(1)], axis=1)
        kb.append((torch.rand(4, 3, 227, 227), torch.from_numpy(bbox).float(), torch.from_numpy(q).float(), torch.from_numpy(t).float(), idx))
    calls = []

    class OneHotAfterForward(torch.nn.Module):
        def __init__(self, net): super().__init__(); self.net = net
        def forward(self, images):
            c, r = self.net(images)
            assert r.shape == (4, NC) and torch.isfinite(r.float()).all() and torch.isfinite(c.float()).all()
            idx = kb[len(calls)][4]; calls.append(1)
            w = torch.full((4, NC), -40.0)
            w[torch.arange(4), torch.from_numpy(idx)] = 40.0
            return c, w
    perf = valid_spn(0, cfg, OneHotAfterForward(model), [b[:4] for b in kb], cameraMatrix, distCoeffs, corners3D, None, device, qClass)
    assert len(calls) == 2 and perf['eR'].avg < 1e-2 and perf['eT'].avg < 2e-3 and perf['speed (raw)'].avg < 2e-3
