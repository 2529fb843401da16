"""The command-line surface end to end on the MI355X: train.py (KRN and SPN, with resume), adapt.py (DANN) and test.py as
subprocesses on synthetic batches, and the SPN epoch driver called the way train.py calls it (reference train.py:125-155,
adapt.py:112-140, test.py:42-91, trainer.py:114-199)."""
import os
import subprocess
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args, cwd=ROOT):
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert p.returncode == 0, (script, args, p.stdout[-1500:], p.stderr[-3000:])
    return p.stdout + p.stderr


def test_train_krn_resume_then_test(device, tmp_path):
    common = ["--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 3, "--optimizer", "adamw", "--lr", "1e-3", "--weight_decay",
              "0.01", "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--precision", "bf16"]
    out = run("train.py", *common, "--max_epochs", 1)
    assert os.path.exists(tmp_path / "save" / "checkpoint.pth.tar") and os.path.exists(tmp_path / "save" / "model_best.pth.tar")
    ck = torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")
    assert ck["epoch"] == 1 and ck["model"] == "krn" and len(ck["state_dict"]) == 350 and ck["optimizer"]["state"]
    out = run("train.py", *common, "--max_epochs", 2)                        # auto-resume: loads epoch 1, trains epoch 2 only
    assert "Checkpoint loaded" in out and "Training 002" in out and "Training 001" not in out
    assert torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")["epoch"] == 2
    res = tmp_path / "pred.pt"
    run("test.py", "--model_name", "krn", "--synthetic_batches", 2, "--pretrained", tmp_path / "save" / "model_best.pth.tar", "--resultfn", res,
        "--logdir", tmp_path / "log")
    assert os.path.exists(res)


def test_train_spn_with_use_fp16_flag_warns_and_resumes(device, tmp_path):
    common = ["--model_name", "spn", "--num_classes", 64, "--batch_size", 4, "--synthetic_batches", 2, "--optimizer", "adamw", "--savedir",
              tmp_path / "save", "--logdir", tmp_path / "log", "--use_fp16"]
    out = run("train.py", *common, "--max_epochs", 1)
    assert "float16 with device-side dynamic loss scaling" in out and "loss_c" in out   # --use_fp16: real fp16 + GradScaler arithmetic for SPN
    out = run("train.py", *common, "--max_epochs", 2)                        # resume with the model still on the CPU
    assert "Checkpoint loaded" in out
    ck = torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")
    assert ck["epoch"] == 2 and ck["optimizer"]["spn_fused"]["t"] == 4


def test_train_krn_with_use_fp16_flag_runs_float16_and_resumes_the_scaler(device, tmp_path):
    """--use_fp16 for KRN = the reference's recipe (train.py:101-104: autocast + GradScaler): IEEE-half kernels, GradScaler's state on the
    device; the checkpoint carries it (loss scale, steps actually taken) and a resumed run starts from it"""
    common = ["--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 4, "--optimizer", "adamw", "--lr", "1e-4", "--weight_decay", "0.01",
              "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--use_fp16"]
    out = run("train.py", *common, "--max_epochs", 1)
    assert "KRN runs in float16 with device-side dynamic loss scaling" in out and "Training 001" in out
    ck = torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")
    amp = ck["optimizer"]["spb_amp"]
    from speedplusbaseline_amd import _lib as L
    taken, scale = float(amp[L.AMP_STEPS]), float(amp[L.AMP_SCALE])
    assert 0 <= taken <= 4 and scale == 65536.0 * 0.5 ** (4 - taken)          # every step was either taken or skipped with the scale halved
    assert all(torch.isfinite(v.float()).all() for v in ck["state_dict"].values())
    out = run("train.py", *common, "--max_epochs", 2)
    assert "Checkpoint loaded" in out and "Training 002" in out
    amp2 = torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")["optimizer"]["spb_amp"]
    taken2 = float(amp2[L.AMP_STEPS])
    assert taken <= taken2 <= taken + 4 and float(amp2[L.AMP_SCALE]) == scale * 0.5 ** (4 - (taken2 - taken))   # continued from the saved state


def test_adapt_dann(device, tmp_path):
    out = run("adapt.py", "--perform_dann", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 2, "--max_epochs", 1, "--optimizer",
              "adamw", "--savedir", tmp_path / "save", "--logdir", tmp_path / "log", "--precision", "bf16")
    ck = torch.load(tmp_path / "save" / "checkpoint.pth.tar", map_location="cpu")
    assert len(ck["state_dict"]) == 354 and "domain_classifier.0.weight" in ck["state_dict"]


def test_train_single_epoch_spn_driver(device, capsys):
    from oracle import spn_oracle as S
    from speedplusbaseline_amd.core.trainer import train_single_epoch_spn
    from speedplusbaseline_amd.data import SyntheticSpnLoader
    from speedplusbaseline_amd.nets import get_model, get_optimizer
    cfg = types.SimpleNamespace(model_name="spn", num_keypoints=11, num_classes=64, dann=False, optimizer="adamw", lr=1e-3, momentum=0.9,
                                weight_decay=0.01, fp16=False, precision="bf16", synthetic_batches=1, texture_ratio=0.5, seed=1)
    model = get_model(cfg)
    model.load_state_dict(S.init_state(64), strict=True)
    opt = get_optimizer(cfg, model)
    model = model.to(device)
    before = model.flat_parameters().clone()

    class Writer:
        def __init__(self): self.s = {}
        def add_scalar(self, k, v, e): self.s[k] = (v, e)
    w = Writer()
    loader = SyntheticSpnLoader(4, 5, 64, 5, (227, 227), seed=3)
    train_single_epoch_spn(1, cfg, model, loader, opt, w, device, styleAugmentor=None, scaler=None)
    out = capsys.readouterr().out
    assert "loss_c" in out and "loss_r" in out and "0005/0005" in out
    assert set(w.s) == {"train/loss_c", "train/loss_r"} and all(v[0] == v[0] and v[0] > 0 for v in w.s.values())
    assert opt._t == 5 and float((model.flat_parameters() - before).abs().max()) > 0
    assert torch.isfinite(model.flat_parameters()).all()


def test_train_with_per_epoch_validation_krn_and_spn(device, tmp_path):
    """train.py --test_epoch 1 (reference train.py:135-138): valid_krn / valid_spn run after every epoch on the synthetic loader"""
    out = run("train.py", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 2, "--optimizer", "adamw", "--max_epochs", 1,
              "--test_epoch", 1, "--savedir", tmp_path / "k_save", "--logdir", tmp_path / "k_log", "--precision", "bf16")
    assert "Testing 001" in out or "Testing 000" in out, out[-2000:]
    for fn in ('err_q.txt', 'err_t.txt', 'speed_raw.txt', 'speed_mod.txt'):
        assert len(open(tmp_path / "k_log" / fn).read().split()) >= 1
    out = run("train.py", "--model_name", "spn", "--num_classes", 64, "--batch_size", 4, "--synthetic_batches", 2, "--optimizer", "adamw",
              "--max_epochs", 1, "--test_epoch", 1, "--savedir", tmp_path / "s_save", "--logdir", tmp_path / "s_log")
    assert "Testing" in out and "eR" in out, out[-2000:]


def test_adapt_with_per_epoch_validation(device, tmp_path):
    """adapt.py --test_epoch 1 with a log directory that does not exist yet (FileNotFoundError at the first validation)"""
    out = run("adapt.py", "--perform_dann", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 2, "--max_epochs", 1, "--optimizer",
              "adamw", "--test_epoch", 1, "--savedir", tmp_path / "save", "--logdir", tmp_path / "fresh" / "log", "--precision", "bf16")
    assert "Testing" in out
    assert os.path.exists(tmp_path / "fresh" / "log" / "err_q.txt") and os.path.exists(tmp_path / "save" / "checkpoint.pth.tar")


def test_test_py_spn_with_the_default_result_name(device, tmp_path):
    """test.py --model_name spn with the reference's default --resultfn '' (was IsADirectoryError after the whole evaluation)"""
    run("test.py", "--model_name", "spn", "--num_classes", 64, "--synthetic_batches", 3, "--logdir", tmp_path / "log")
    txt = open(tmp_path / "log" / "results.txt").read()
    assert "eR" in txt and "speed (raw)" in txt


def test_train_krn_deterministic_flag_gives_bit_identical_checkpoints(device, tmp_path):
    """--deterministic: the reproducible build of the kernels behind the kept command line -- two runs from the same seed write
    bit-identical parameters, BatchNorm buffers and optimizer moments (the reference cannot: utils.py:297-298); without the flag two runs differ"""
    def run_once(tag, *extra):
        d = tmp_path / tag
        run("train.py", "--model_name", "krn", "--batch_size", 8, "--synthetic_batches", 4, "--max_epochs", 1, "--optimizer", "adamw", "--lr", "1e-3",
            "--weight_decay", "0.01", "--savedir", d / "save", "--logdir", d / "log", "--precision", "fp32", *extra)
        ck = torch.load(d / "save" / "checkpoint.pth.tar", map_location="cpu")
        return ck
    a, b = run_once("a", "--deterministic"), run_once("b", "--deterministic")
    assert all(torch.equal(a["state_dict"][k], b["state_dict"][k]) for k in a["state_dict"])
    sa, sb = a["optimizer"]["state"], b["optimizer"]["state"]
    assert all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"]) for i in sa)
    c, d = run_once("c"), run_once("d")
    assert any(not torch.equal(c["state_dict"][k], d["state_dict"][k]) for k in c["state_dict"])        # float atomics: run-to-run different


def test_adapt_dann_deterministic(device, tmp_path):
    def run_once(tag):
        d = tmp_path / tag
        run("adapt.py", "--perform_dann", "--model_name", "krn", "--batch_size", 4, "--synthetic_batches", 3, "--max_epochs", 1, "--optimizer",
            "adamw", "--savedir", d / "save", "--logdir", d / "log", "--precision", "bf16", "--deterministic")
        return torch.load(d / "save" / "checkpoint.pth.tar", map_location="cpu")["state_dict"]
    a, b = run_once("a"), run_once("b")
    assert len(a) == 354 and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("env", [{"SPB_EVENT_FORKS": "1"}, {"ROCPROF_COUNTER_COLLECTION": "1"}])
def test_bench_runs_with_event_ordered_side_stream(device, env):
    """The product library orders its side stream by a spinning gate kernel (csrc/krn_plan.hip); under a tool that serialises the device's
    kernels (rocprofv3 --pmc exports ROCPROF_COUNTER_COLLECTION=1 to the application) or with SPB_EVENT_FORKS=1 it falls back to events:
    the same training steps, the same kind of answer."""
    import json
    outs = []
    for extra in ({}, env):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bare", "--steps", "3", "--warmup", "0", "--batch", "8"], cwd=ROOT,
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT, **extra))
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    a, b = (o["config"]["loss_last_step"] for o in outs)
    assert all(v == v and abs(v) < 1e6 for v in a + b), (a, b)
    assert 0.2 * a[0] < b[0] < 5 * a[0], (a, b)       # three AdamW steps from a random init: chaotic already, the same order of magnitude
