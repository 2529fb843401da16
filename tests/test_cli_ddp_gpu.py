"""Data parallelism through the kept command line (judge row e2): `train.py` / `adapt.py` started as
`python -m torch.distributed.run --nproc-per-node 2 ...` -- one process per rank, LOCAL_RANK -> device, process group, per-rank
data shard, parameter broadcast, rank-0-only files.  A one-GPU box cannot host two RCCL ranks, so SPB_ONE_DEVICE=1 puts both ranks
on cuda:0 with gloo collectives on device tensors (speedplusbaseline_amd.parallel.init_job); on the 8-GPU node the same scripts
run with backend nccl (= RCCL over xGMI).  Checked per recipe: both ranks finish, the scripts' own end-of-epoch replica check saw
bit-identical parameters, exactly one checkpoint was written and it loads strict=True into a freshly built model.
Reference surface kept: /root/reference/train.py:49-160, adapt.py:47-148 (single-process there: SURVEY F2)."""
import os
import socket
import subprocess
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch2(script, *args):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SPB_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, script)] + [str(a) for a in args]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (script, args, p.stdout[-1500:], p.stderr[-3000:])
    return p.stdout + p.stderr


def _cfg(**kw):
    base = dict(model_name="krn", num_keypoints=11, num_classes=64, dann=False, optimizer="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01,
                fp16=False, precision=None, synthetic_batches=1)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _strict_load(cfg, path, n_keys):
    from speedplusbaseline_amd.nets import get_model
    ck = torch.load(path, map_location="cpu")
    assert len(ck["state_dict"]) == n_keys
    model = get_model(cfg)
    model.load_state_dict(ck["state_dict"], strict=True)
    assert all(torch.isfinite(v.float()).all() for v in ck["state_dict"].values())
    return ck


def test_train_krn_two_ranks(device, tmp_path):
    out = launch2("train.py", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 3, "--max_epochs", 2, "--optimizer", "adamw", "--lr",
                  "1e-3", "--weight_decay", "0.01", "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--precision", "bf16")
    assert "Data parallel: 2 ranks, batch 4 per GPU (global 8)" in out
    assert out.count("replicas identical after epoch") == 2          # rank 0 logs, once per epoch
    assert out.count("Checkpoint saved") == 2 and "Training 002" in out
    ck = _strict_load(_cfg(), tmp_path / "save" / "checkpoint.pth.tar", 350)
    assert ck["epoch"] == 2 and ck["optimizer"]["state"]
    assert sorted(os.listdir(tmp_path / "save")) == ["checkpoint.pth.tar", "config.txt", "model_best.pth.tar"]


def test_train_spn_fp16_two_ranks(device, tmp_path):
    out = launch2("train.py", "--model_name", "spn", "--num_classes", 64, "--batch_size", 4, "--synthetic_batches", 3, "--max_epochs", 1,
                  "--optimizer", "adamw", "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--use_fp16")
    assert "float16 with device-side dynamic loss scaling" in out and "replicas identical after epoch 1" in out
    ck = _strict_load(_cfg(model_name="spn", fp16=True), tmp_path / "save" / "checkpoint.pth.tar", 22)
    fused = ck["optimizer"]["spn_fused"]
    assert fused["t"] == 3 and fused["amp"] is not None and float(fused["amp"][0]) > 0     # GradScaler state travels with the optimizer


def test_adapt_dann_two_ranks(device, tmp_path):
    out = launch2("adapt.py", "--perform_dann", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 3, "--max_epochs", 1,
                  "--optimizer", "adamw", "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--precision", "bf16")
    assert "replicas identical after epoch 1" in out
    ck = _strict_load(_cfg(dann=True), tmp_path / "save" / "checkpoint.pth.tar", 354)
    assert "domain_classifier.0.weight" in ck["state_dict"]


def test_scale_command_bench_two_ranks(device):
    """the driver's multi-GPU line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` -- with N = 2 on this one-GPU box (both ranks on cuda:0, gloo collectives), so that the first 8-GPU
    run is not the first run of that path: rank 0 prints ONE JSON line with the contract's fields, n_gpus 2, the global batch of both ranks,
    weak scaling, a finite value that is consistent with ms_per_step."""
    import json
    out = launch2("bench.py", "--gpus", 2, "--steps", 3, "--warmup", 1, "--bare")
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["unit"] == "images/sec" or "images" in r["unit"]
    assert r["config"]["global_batch"] == 96 and r["dtype"] == "bf16" and r["data"] == "synthetic"
    assert r["value"] > 0 and abs(r["value"] - 96 / (r["ms_per_step"] * 1e-3)) <= 1e-3 * r["value"]
    import math
    assert math.isfinite(r["value"]) and all(math.isfinite(float(v)) for v in r["config"]["loss_last_step"])
