#!/usr/bin/env python
"""Throughput of the KRN training step on MI355X (BASELINE.json metric: images/sec, KRN 224x224, bs=48/GPU).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = forward + zero_grad + backward + [gradient all-reduce over RCCL] + global-norm clip + AdamW on one batch of
synthetic images already resident in HBM (README recipe: bs 48, AdamW lr 1e-3 wd 0.01; trainer.py:72-98 order).  Prints
ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the dominant kernel family's algorithmic bytes / its launch time, timed live with HIP events on the launch
                stream in an instrumented pass of the same step (speedplusbaseline_amd engine profiler)
  kernels       the same for every kernel family of the step
  cpu_baseline  the CPU oracle (oracle/krn_oracle.py, the reference's --no_cuda fp32 path restated) on this box's cores
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: one hardware queue per stream (see the package's __init__)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}


def _host():
    """CPU model and core count of this box (stated beside the CPU baseline)"""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(cpu_model=model, host_cores=os.cpu_count())


def _pmc_traffic(family, files=("r6_pmc_traffic.json",), alg_bytes=None):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs,
    corrected as MI355X_MICROARCH.md prescribes; scratch/pmc_summary.py) -- rocprofv3 cannot run inside bench.py.  Only the passes of the
    CURRENT round's kernels are consulted (an older file describes other launches), and a figure below 0.7 x the algorithmic bytes of the
    same family is a mapping error of the summary script (round 5: a new kernel landed under `other:`), not a measurement: it is refused."""
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)["families"][family]["hbm_bytes_per_launch"]
        except Exception:
            continue
        if alg_bytes and t < 0.7 * alg_bytes:
            sys.stderr.write("bench.py: PMC traffic of %s in %s (%.1f MB per launch) is below 0.7 x its algorithmic bytes (%.1f MB): refused\n"
                             % (family, name, t / 1e6, alg_bytes / 1e6))
            return None, name + " (refused: below 0.7 x algorithmic)"
        return t, name
    return None, None


def _device_index(local_rank):
    # SPB_ONE_DEVICE=1 (test rigs with one GPU): every rank on device 0, with SPB_DIST_BACKEND=gloo for the collectives
    return 0 if os.environ.get("SPB_ONE_DEVICE") == "1" else local_rank


def _init_dist(dev):
    # "nccl" is RCCL on ROCm; a one-GPU test rig (SPB_ONE_DEVICE=1: every rank on device 0) cannot host two RCCL ranks and takes gloo
    backend = os.environ.get("SPB_DIST_BACKEND") or ("gloo" if os.environ.get("SPB_ONE_DEVICE") == "1" else "nccl")
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", device_id=dev)
    else:
        torch.distributed.init_process_group(backend)

# ---- step-level algorithmic bytes: what `roofline` of the non-headline workloads is computed from
KRN_BYTES_PER_IMAGE = 83.9e6        # bf16 activations, each conv reads its input once and writes its output once, x3 (fwd, dgrad, wgrad)
GHIASI_FLOPS_PER_IMAGE = 15.43e9    # decoder forward at 224x224
GHIASI_BYTES_PER_IMAGE = 45.95e6


def _timed(fn, steps, warmup):
    """ms per call of fn(): HIP-synchronised wall clock around `steps` calls after `warmup` untimed ones"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _hbm_roof(alg_bytes, ms):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                alg_bytes_per_step=round(alg_bytes))


def spn_step_bytes(net, B, precision):
    """algorithmic HBM bytes of one SPN train step: the optimizer's pass over the f32 arenas (p, g, m, v read; p, m, v written:
    28 B per parameter, + the 16-bit shadow written, + fp16's inf / nan check reading g once more), the fully connected weights
    streamed as 16-bit shadows by forward and input gradient (2 x 2 B) and their f32 gradient written once (4 B), and the
    convolution trunk's activations (2.2 MB per image and pass)"""
    n = net.flat_parameters().numel()
    n_fc = n - net._conv_end
    half = precision in ("bf16", "fp16")
    opt = n * (28 + (2 if half else 0) + (4 if precision == "fp16" else 0))
    fc = n_fc * ((2 + 2 if half else 4 + 4) + 4)
    return opt + fc + 3 * 2.2e6 * B * (1 if half else 2)


def bench_others(steps=40, warmup=10):
    """The other BASELINE.json workloads on this GPU, each for a short, fixed number of steps AFTER the headline region, so that they
    sit under the driver's clock too: DANN (configs[3]) at the README's bs=16 and at bs=48, SPN (configs[5]) in bf16 and fp16,
    KRN + style augmentation (configs[4]) and the style decoder alone.  Each runs `bench.py --bare` of its own in a fresh process --
    exactly the command a reader would run by hand: measured in THIS process after the headline region they were 1.5-3x slower
    (DANN bs=16 7.4 instead of 2.6 ms), because the HIP runtime hands hardware queues to streams round robin and the streams of the
    finished workloads keep theirs: two concurrently active streams of a later workload end up sharing a queue."""
    import subprocess
    runs = (("dann_bs16", ["--model", "dann", "--batch", "16"]), ("dann_bs48", ["--model", "dann", "--batch", "48"]),
            ("spn_bf16", ["--model", "spn", "--precision", "bf16"]), ("spn_fp16", ["--model", "spn", "--precision", "fp16"]),
            ("krn_fp16", ["--precision", "fp16"]),     # the reference's own AMP recipe for KRN (float16 + GradScaler), beside the bf16 headline
            ("styleaug", ["--styleaug"]), ("decoder", ["--model", "decoder"]),
            ("decoder_fp16", ["--model", "decoder", "--precision", "fp16"]))   # the same kernels in IEEE half: 8x closer to the reference's float32 decoder
    out = {}
    t_start = time.perf_counter()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    for name, extra in runs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--bare", "--steps", str(steps), "--warmup", str(warmup)] + extra
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
        except Exception as e:      # never lose the headline line to a side measurement
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            continue
        ms, bsz = d["ms_per_step"], d["config"].get("per_gpu_batch", 48)
        roof = d.get("roofline")
        if name.startswith("dann"):
            roof = _hbm_roof(2 * KRN_BYTES_PER_IMAGE * bsz, ms)          # two passes (source, target) of the KRN traffic
        elif name == "styleaug":
            roof = _hbm_roof(KRN_BYTES_PER_IMAGE * bsz + 0.5 * GHIASI_BYTES_PER_IMAGE * bsz, ms)   # the KRN step + half a restyle (coin p = 0.5)
        elif name == "krn_fp16":
            roof = _hbm_roof(KRN_BYTES_PER_IMAGE * bsz, ms)
        out[name] = dict(ms_per_step=ms, value=d["value"], unit=d["unit"], dtype=d["dtype"], steps=d["steps"], roofline=roof,
                         workload=d["config"]["workload"], command="python bench.py " + " ".join(cmd[2:]))
    out["_note"] = ("each entry: its own `python bench.py --bare ...` process started by this one after the headline region (%d steps after %d "
                    "warm-up steps); roofline = SURVEY 8d algorithmic bytes (decoder: flops) per step / step time; %.0f s in total"
                    % (steps, warmup, time.perf_counter() - t_start))
    return out


def bench_decoder(args):
    """the Ghiasi style decoder alone (15.43 GFLOP per image at 224x224): one restyle of a resident batch per step"""
    from speedplusbaseline_amd.styleaug import Ghiasi, StyleAugmentor
    dev = torch.device("cuda", 0)
    B = args.batch
    torch.manual_seed(2021)
    prec = "fp16" if args.precision == "fp16" else "bf16"   # (--precision fp32 is the default of nothing here: the headline default is bf16)
    aug = StyleAugmentor.synthetic(0.5, dev, Ghiasi().state_dict(), seed=2021, precision=prec)   # random decoder weights (none offline)
    gen = torch.Generator(device="cpu"); gen.manual_seed(2021)
    x = torch.rand(B, 3, 224, 224, generator=gen).to(dev)
    ms = _timed(lambda: aug(x), args.steps, args.warmup)
    tf = GHIASI_FLOPS_PER_IMAGE * B / (ms * 1e-3) / 1e12
    traffic = src = None
    if B == 48:
        for nm in ("r6_ghiasi_pmc_traffic.json", "r5_ghiasi_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", nm)) as f:
                    traffic, src = json.load(f)["restyle_hbm_bytes"], nm
                break
            except Exception:
                continue
    print(json.dumps({
        "metric": "images/sec Ghiasi style decoder 224x224 forward", "value": round(B / (ms * 1e-3), 1), "unit": "images/sec", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": prec, "data": "synthetic",
        "config": {"workload": "Ghiasi style decoder forward (StyleAugmentor, alpha 0.5), %d images 224x224" % B, "per_gpu_batch": B, "weights": "random init"},
        "roofline": dict(bound="mfma", kernel="all launches of one restyle", achieved=round(tf, 1), peak=MFMA_PEAK_TFLOPS["bf16"], unit="TFLOP/s",
                         frac=round(tf / MFMA_PEAK_TFLOPS["bf16"], 4), traffic=traffic, traffic_source=src,
                         alg_flops_per_step=GHIASI_FLOPS_PER_IMAGE * B, alg_bytes_per_step=GHIASI_BYTES_PER_IMAGE * B),
        "cpu_baseline": None}))


def main():
    from speedplusbaseline_amd import _lib as L
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=48, help="images per GPU (README recipe: 48)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp16"],
                    help="bf16: the headline dtype (BASELINE configs[1]); fp16: the IEEE-half build of the same kernels with GradScaler's dynamic loss "
                         "scaling on the device (--model krn: the reference's own --use_fp16 recipe; --model spn: BASELINE configs[5]); fp32: parity mode")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph instead of enqueuing the launches "
                    "eagerly (eager + side-stream weight gradients is the faster, default mode)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)  # old spelling of the default
    ap.add_argument("--model", default="krn", choices=["krn", "spn", "dann", "preproc", "decoder"], help="krn: the headline benchmark (BASELINE configs[1]); "
                    "spn: Spacecraft Pose Network train step, 227x227, bs=32, 5000 classes (configs[5] flavour); "
                    "dann: RevGrad domain-adversarial step, source + target batch (configs[3] flavour; --batch 16 is the README recipe)")
    ap.add_argument("--styleaug", action="store_true", help="BASELINE configs[4] flavour: restyle the batch with the Ghiasi "
                    "decoder on a rank-synchronous coin (p=0.5, alpha=0.5) before the train step (trainer.py:68-69)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the `others` block of the KRN line (the other BASELINE workloads, "
                    "timed briefly in the same process after the headline region)")
    ap.add_argument("--bare", action="store_true", help="timed region only (no instrumented pass, no others, no CPU baseline): what the "
                    "rocprofv3 PMC passes run, so that a pass holds exactly warmup + steps steps")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=32)
    args = ap.parse_args()

    if args.bare:
        args.no_cpu_baseline = args.no_others = True
    if args.precision == "fp16" and args.model not in ("spn", "krn", "decoder"):
        raise SystemExit("--precision fp16 exists for --model spn, krn and decoder (RevGrad / DANN: bf16 with f32 accumulation, DESIGN.md a13)")
    if args.model == "spn":
        return bench_spn(args)
    if args.model == "dann":
        return bench_dann(args)
    if args.model == "preproc":
        return bench_preproc(args)
    if args.model == "decoder":
        return bench_decoder(args)
    from speedplusbaseline_amd.engine import KrnEngine
    from speedplusbaseline_amd.step import FusedTrainStep
    from oracle import krn_oracle as O  # checker / cpu_baseline leg only

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    dev = torch.device("cuda", _device_index(local))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _init_dist(dev)
        group = torch.distributed.group.WORLD

    # A/B experiments: SPB_DEBUG="spb_debug_set_x:1,2;spb_debug_set_y:3" makes speedplusbaseline_amd._lib load the TUNING build
    # (libspb_hip_tune.so, include/spb_hip_tuning.h) and apply the knobs; unset, the product library runs, which has none.
    B = args.batch
    eng = KrnEngine(11).attach(dev, args.precision)
    sd = O.init_state(11)  # random-init weights of the KRN architecture (no checkpoints offline)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(dev))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(dev))
    if world > 1:  # identical replicas: rank 0's parameters everywhere
        torch.distributed.broadcast(eng.params, 0); torch.distributed.broadcast(eng.buffers, 0)
    gen = torch.Generator(device="cpu"); gen.manual_seed(2021 + rank)  # seed 2021 (config.py:13) + rank offset
    x = torch.rand(B, 3, 224, 224, generator=gen).to(dev)   # U[0,1) like transforms.py:192-196
    y = torch.rand(B, 2, 11, generator=gen).to(dev)
    step = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0,
                          dist_group=group, world_size=world, use_graph=args.graph)

    aug = None
    if args.styleaug:
        from speedplusbaseline_amd.styleaug import Ghiasi, StyleAugmentor
        from speedplusbaseline_amd.parallel import shared_coin
        torch.manual_seed(2021)
        aug = StyleAugmentor.synthetic(0.5, dev, Ghiasi().state_dict(), seed=2021)   # random decoder weights (none offline)

    # style augmentation one batch ahead on a side stream, as core/trainer.py's AugLookahead does for a real loader: the
    # decoder of step i+1 is enqueued before train step i.  Every timed step still pays for exactly one coin / restyle.
    aug_stream = torch.cuda.Stream(device=dev) if aug is not None else None
    staged = {}

    def stage(i, xs):
        if not shared_coin(i, 2021, 0.5):
            return xs, None
        aug_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aug_stream):
            out = aug(xs)
            ev = torch.cuda.Event(); ev.record(aug_stream)
        return out, ev

    def one_step(i, xs, ys):
        if aug is None:
            return step(xs, ys)
        cur = staged.pop(i, None) or stage(i, xs)
        if os.environ.get("SPB_AUG_LOOKAHEAD", "1") != "0":
            staged[i + 1] = stage(i + 1, xs)
        xin, ev = cur
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            xin.record_stream(torch.cuda.current_stream())
        return step(xin, ys)

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i, x, y)
    if step.static_inputs() is not None:  # replay straight from the graph's input buffers
        x, y = step.static_inputs()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        scal = one_step(args.warmup + i, x, y)
    t_host = time.perf_counter() - t0          # the host's share: all K steps enqueued (it runs ahead of the GPU until the queues fill)
    sync_all()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss_last = [float(v) for v in scal.cpu()]
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    # ---- instrumented pass: per-kernel-family time (HIP events on the launch stream) and algorithmic bytes
    kernels = {}
    roofline = None
    if rank == 0 and not args.bare:
        es = 2 if args.precision in ("bf16", "fp16") else 4
        n_prof = 5
        eng.prof_enable(B, 0, True)
        import ctypes as C
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        t_prep = t_zero = t_opt = 0.0
        agg = {}
        for _ in range(n_prof):
            step.t += 1; step._refresh_hyper()
            _, _, _ = eng.forward(x, y, training=True, slot=0)
            ev[0].record(); eng.grads.zero_(); ev[1].record()
            eng.backward(B, slot=0)
            ev[2].record(); step._update(); ev[3].record()
            ev[4].record()
            L.check(eng.lib.spb_krn_prepare_weights(eng.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "prepare_weights")
            ev[5].record()
            torch.cuda.synchronize()
            t_zero += ev[0].elapsed_time(ev[1]); t_opt += ev[2].elapsed_time(ev[3]); t_prep += ev[4].elapsed_time(ev[5])
            for k, v in eng.prof_read(B, 0).items():
                a = agg.setdefault(k, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
                for f in a:
                    a[f] += v[f]
        eng.prof_enable(B, 0, False)
        agg["weight_prep"] = dict(launches=n_prof, ms=t_prep, bytes=float(eng.weight_prep_bytes()) * n_prof, flops=0.0)
        agg["grad_zero"] = dict(launches=n_prof, ms=t_zero, bytes=4.0 * eng.n_params * n_prof, flops=0.0)
        agg["clip_adamw"] = dict(launches=2 * n_prof, ms=t_opt, bytes=(4.0 + 28.0) * eng.n_params * n_prof, flops=0.0)
        tot_ms = sum(v["ms"] for v in agg.values())
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            ms = v["ms"] / n_prof
            kernels[k] = dict(launches_per_step=v["launches"] // n_prof, ms_per_step=round(ms, 4),
                              share=round(v["ms"] / tot_ms, 4),
                              GBps=round(v["bytes"] / n_prof / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                              TFLOPs=round(v["flops"] / n_prof / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                              alg_MB_per_step=round(v["bytes"] / n_prof / 1e6, 2))
        dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
        dk, dv = dom
        achieved = dv["bytes"] / (dv["ms"] * 1e-3) / 1e9
        # HBM bytes per launch of that family from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs, corrected as MI355X_MICROARCH.md prescribes; scratch/pmc_summary.py) -- rocprofv3 cannot run inside bench.py
        traffic, traffic_src = _pmc_traffic(dk, alg_bytes=dv["bytes"] / dv["launches"])
        roofline = dict(bound="hbm", kernel=dk, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                        launches_per_step=dv["launches"] // n_prof,
                        avg_launch_us=round(dv["ms"] / dv["launches"] * 1e3, 2),
                        alg_bytes_per_launch=round(dv["bytes"] / dv["launches"]),
                        traffic_over_alg=round(traffic / (dv["bytes"] / dv["launches"]), 3) if traffic else None,
                        step_sum_of_kernels_ms=round(tot_ms / n_prof, 3),
                        step_alg_GBps=round(sum(v["bytes"] for v in agg.values()) / n_prof / (ms_per_step * 1e-3) / 1e9, 1),
                        step_alg_TFLOPs=round(sum(v["flops"] for v in agg.values()) / n_prof / (ms_per_step * 1e-3) / 1e12, 2),
                        mfma_peak_TFLOPs=MFMA_PEAK_TFLOPS[args.precision],
                        # the same figure for every family above 5 % of the step (the depthwise backward runs as two kernels since
                        # round 3 -- input gradient on the launch stream, weight gradient on the side stream -- and is measured as such)
                        families={k: dict(frac=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          achieved=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                          avg_launch_us=round(v["ms"] / v["launches"] * 1e3, 2),
                                          alg_bytes_per_launch=round(v["bytes"] / v["launches"]),
                                          **(lambda t: dict(traffic=t, traffic_over_alg=round(t / (v["bytes"] / v["launches"]), 3) if t else None))(
                                              _pmc_traffic(k, alg_bytes=v["bytes"] / v["launches"])[0]))
                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0.05 * tot_ms})

    # ---- style augmentation: the decoder is the matrix-core-bound kernel family of this workload (15.43 GFLOP per
    # image, 0.74 TFLOP per restyled batch); its achieved rate, timed alone with HIP events on the launch stream
    if rank == 0 and aug is not None and not args.bare:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            aug(x)
        torch.cuda.synchronize()
        n_dec = 10
        e0.record()
        for _ in range(n_dec):
            aug(x)
        e1.record()
        torch.cuda.synchronize()
        dec_ms = e0.elapsed_time(e1) / n_dec
        dec_tf = 15.43e9 * B / (dec_ms * 1e-3) / 1e12
        krn_roofline = roofline
        dec_traffic, dec_src = (None, None)
        if B == 48:
            try:   # HBM bytes of one restyle (all decoder launches), from the committed FETCH_SIZE / WRITE_SIZE passes of scratch/bench_ghiasi.py
                for nm in ("r6_ghiasi_pmc_traffic.json", "r5_ghiasi_pmc_traffic.json"):
                    if os.path.exists(os.path.join(ROOT, "profiles", nm)):
                        with open(os.path.join(ROOT, "profiles", nm)) as f:
                            dec_traffic, dec_src = json.load(f)["restyle_hbm_bytes"], nm
                        break
            except Exception:
                pass
        roofline = dict(bound="mfma", kernel="Ghiasi decoder (all launches of one restyle, %d images)" % B, achieved=round(dec_tf, 1),
                        peak=MFMA_PEAK_TFLOPS["bf16"], unit="TFLOP/s", frac=round(dec_tf / MFMA_PEAK_TFLOPS["bf16"], 4),
                        traffic=dec_traffic, traffic_source=dec_src,
                        decoder_ms_per_batch=round(dec_ms, 3), alg_flops_per_batch=15.43e9 * B,
                        train_step_dominant_kernel=krn_roofline)

    # ---- CPU baseline: the oracle's train step (reference --no_cuda fp32 path) on this box's host cores, rank 0 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # N=1 only: the other ranks would sit in the process group meanwhile
        # PyTorch's CPU conv/BN kernels stop scaling (and then collapse: 504 s for 3 steps on 256 threads, measured) well
        # before a 256-core host is full; 32 threads is the best of {8,16,32,64,256} on this class of box
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(ncores)
        tr = O.KrnTrainer(O.init_state(11), "adamw", lr=1e-3, momentum=0.9, weight_decay=0.01)
        xc, yc = x.cpu(), y.cpu()
        t1 = time.perf_counter()
        tr.step(xc, yc)  # warm-up
        warm = time.perf_counter() - t1
        n_cpu = args.cpu_steps if warm < 8.0 else 1  # keep the whole leg within ~30 s
        t1 = time.perf_counter()
        for _ in range(n_cpu):
            tr.step(xc, yc)
        cdt = time.perf_counter() - t1
        args.cpu_steps = n_cpu
        cpu = dict(value=round(B * args.cpu_steps / cdt, 2), unit="images/sec", cores=torch.get_num_threads(), kind="port", **_host(),
                   sample="%d train steps of the same bs=%d 224x224 batch, fp32, PyTorch CPU oracle (%.1f s)" % (args.cpu_steps, B, cdt))

    others = None
    if rank == 0 and world == 1 and not args.no_others and aug is None:
        try:
            others = bench_others()
        except Exception as e:      # never lose the headline line to a side measurement
            others = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        out = {
            "metric": "images/sec KRN 224x224 bs=48/GPU train step",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "KRN (MobileNetV2 features + ConvDw extras + 7x7 keypoint head) train step, 224x224, "
                                   "bs=%d/GPU, AdamW lr 1e-3 wd 0.01 + clip_grad_norm 1.0" % B +
                                   (" + style augmentation (Ghiasi decoder, p=0.5, alpha=0.5)" if args.styleaug else ""),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay (fwd+bwd | all-reduce | clip+AdamW)" if args.graph else
                                 "eager enqueue, weight-gradient GEMMs on a side stream",
                       "weights": "random init (no checkpoints offline)", "loss_last_step": loss_last,
                       "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 4),
                       # the stream forks of the step rest on a runtime property that is tested at start-up (spb_fork_selftest,
                       # include/spb_hip.h): 1 = device-word forks, 0 = the test failed -> events, -1 = events forced by the environment
                       "hip_runtime_version": int(L.lib().spb_hip_runtime_version()), "torch_hip": torch.version.hip,
                       "stream_fork_selftest": int(L.lib().spb_fork_selftest()),
                       "virtual_expanded_tensors": eng.virtual_activations(B, 0)},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "others": others,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_preproc(args):
    """GPU input pipeline: bs=48 regions of interest cut from 1920x1200 grey frames -> Pillow-exact bilinear
    resize to 224x224, ToTensor, rotate / flip / brightness-contrast / noise (p = 0.5 each), crops resident in HBM."""
    from speedplusbaseline_amd.transforms import build_transforms
    from oracle import preproc_oracle as P   # synthetic frames; cpu_baseline leg
    dev = torch.device("cuda", 0)
    B, S = args.batch, 224
    frames = [P.synth_frame(1200, 1920, 40 + i)[:, :, 0] for i in range(4)]          # grey, like SPEED+
    rng = np.random.default_rng(2021)
    fr, boxes, kps = [], [], []
    for i in range(B):
        cx, cy, half = rng.uniform(600, 1300), rng.uniform(400, 800), rng.uniform(120, 380)
        boxes.append(np.array([cx - half, cx + half, cy - half, cy + half], dtype=np.float32))
        kps.append(rng.uniform(0, 1, (2, 11)).astype(np.float32))
        fr.append(frames[i % 4])
    t = build_transforms("krn", (S, S), p_aug=0.5, is_train=True, device=dev, device_noise=True)
    torch.manual_seed(2021)
    a, out, _, _ = t.stage(fr, boxes, kps)
    for _ in range(args.warmup):
        t.launch(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        t.launch(a)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    us = e0.elapsed_time(e1) / args.steps * 1e3
    # algorithmic bytes per batch: crops read once, uint8 intermediate written + read once, float32 output written, noise read
    tab_h = [int(round(float(b[3] - b[2]) * 1.25)) for b in boxes]
    alg = a.src_bytes + 2 * sum(min(h, 1200) * S for h in tab_h) + B * 3 * S * S * 4 + (B // 2) * 3 * S * S * 4
    roofline = dict(bound="hbm", kernel="preproc (coeffs + horizontal pass + vertical pass/ToTensor/augment)", achieved=round(alg / (us * 1e-6) / 1e9, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=None, launches_per_step=3,
                    avg_launch_us=round(us / 3, 1), alg_bytes_per_step=alg)
    # host half (decisions, crop slices, packing, upload) for the same batch, one core
    t1 = time.perf_counter()
    for _ in range(5):
        torch.manual_seed(2021)
        t.stage(fr, boxes, kps)
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t1) / 5 * 1e3
    cpu = None
    if not args.no_cpu_baseline:
        n = 0
        t1 = time.perf_counter()
        torch.manual_seed(2021)
        while time.perf_counter() - t1 < 10.0:
            i = n % B
            P.krn_sample(np.repeat(fr[i][:, :, None], 3, axis=2), boxes[i], kps[i].copy(), S, 0.5, True)
            n += 1
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(n / cdt, 1), unit="images/sec", cores=1, kind="port", **_host(),
                   sample="%d samples through the per-sample CPU pipeline (Pillow resize + torch ops, as the reference's DataLoader "
                          "workers run it), one process (%.1f s)" % (n, cdt))
    print(json.dumps({
        "metric": "images/sec input pipeline: RoI crop -> 224x224 + augmentations, bs=%d" % B, "value": round(B * args.steps / dt, 1),
        "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 -> f32 (22-bit fixed-point filter)", "data": "synthetic",
        "config": {"workload": "GPU half of the input pipeline on %d regions of interest (240..760 px squares, enlarged x1..1.5) of 1920x1200 grey "
                               "frames, crops resident in HBM" % B, "per_gpu_batch": B, "host_half_ms_per_batch": round(host_ms, 2),
                   "host_half_note": "crop decisions + slicing + packing + one upload, single Python process, not in `value`"},
        "roofline": roofline, "cpu_baseline": cpu}))


def bench_dann(args):
    """RevGrad / DANN step (dann.py:68-100): source forward (pose loss + domain BCE vs 1), target forward (domain BCE vs 0),
    one backward through both passes with the gradient reversed at the 7x7x320 feature, clip_grad_norm_(1.0), AdamW.
    One step = B source + B target images; `value` counts source images (the reference's epoch unit)."""
    from speedplusbaseline_amd.engine import KrnEngine
    from speedplusbaseline_amd.step import FusedTrainStep
    from oracle import krn_oracle as O  # checker / cpu_baseline leg only
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d): launch with python -m torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    dev = torch.device("cuda", _device_index(local))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _init_dist(dev)
        group = torch.distributed.group.WORLD
    B = args.batch
    eng = KrnEngine(11, dann=True).attach(dev, args.precision)
    sd = O.init_state(11, dann=True)
    for info in eng.param_infos:
        eng.param_view(info).copy_(sd[info[0]].to(dev))
    for name, shape, off, numel in eng.buffer_infos:
        eng.buffers[off: off + numel].copy_(sd[name].flatten().to(dev))
    if world > 1:
        torch.distributed.broadcast(eng.params, 0); torch.distributed.broadcast(eng.buffers, 0)
    gen = torch.Generator(device="cpu"); gen.manual_seed(2021 + rank)
    xs = torch.rand(B, 3, 224, 224, generator=gen).to(dev)
    ys = torch.rand(B, 2, 11, generator=gen).to(dev)
    xt = torch.rand(B, 3, 224, 224, generator=gen).to(dev)
    step = FusedTrainStep(eng, B, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01, max_norm=1.0, dist_group=group,
                          world_size=world, use_graph=False, dann=True)
    alpha = O.dann_alpha(5, 1, 100, 10)

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(xs, ys, xt, alpha)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scal = step(xs, ys, xt, alpha)
    sync_all()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    # the host's share: one step enqueued into EMPTY queues (nothing throttles the host), best of five
    t_host = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(xs, ys, xt, alpha)
        t_host = min(t_host, time.perf_counter() - t1)
    torch.cuda.synchronize()
    # ---- instrumented pass (both passes back to back on the launch stream, HIP events around every launch): the dominant
    # kernel family of the step and its achieved HBM rate, as for the KRN line
    roofline = None
    if rank == 0 and not args.bare:
        n_prof = 3
        eng.prof_enable(B, 0, True); eng.prof_enable(B, 1, True)
        agg = {}
        from speedplusbaseline_amd import ops as _ops
        for _ in range(n_prof):
            _ops.arena_zero(eng.grads)
            _, _, dom_s = eng.forward(xs, ys, training=True, slot=0, domain=True)
            _, dl_s = eng.bce_logits(dom_s, 1.0)
            _, _, dom_t = eng.forward(xt, None, training=True, slot=1, domain=True)
            _, dl_t = eng.bce_logits(dom_t, 0.0)
            eng.backward(B, slot=0, with_pose=True, dlogit=dl_s, alpha=alpha)
            eng.backward(B, slot=1, with_pose=False, dlogit=dl_t, alpha=alpha)
            torch.cuda.synchronize()
            for slot in (0, 1):
                for k, v in eng.prof_read(B, slot).items():
                    a = agg.setdefault(k, dict(launches=0, ms=0.0, bytes=0.0, flops=0.0))
                    for f in a:
                        a[f] += v[f]
        eng.prof_enable(B, 0, False); eng.prof_enable(B, 1, False)
        dk, dv = max(agg.items(), key=lambda kv: kv[1]["ms"])
        ach = dv["bytes"] / (dv["ms"] * 1e-3) / 1e9
        traffic, traffic_src = _pmc_traffic(dk, ("r6_dann_pmc_traffic.json",), alg_bytes=dv["bytes"] / dv["launches"]) if B == 48 else (None, None)   # passes taken at bs=48+48
        roofline = dict(bound="hbm", kernel=dk, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=traffic_src,
                        launches_per_step=dv["launches"] // n_prof, avg_launch_us=round(dv["ms"] / dv["launches"] * 1e3, 2),
                        alg_bytes_per_launch=round(dv["bytes"] / dv["launches"]),
                        traffic_over_alg=round(traffic / (dv["bytes"] / dv["launches"]), 3) if traffic else None,
                        step_sum_of_kernels_ms=round(sum(v["ms"] for v in agg.values()) / n_prof, 3))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # N=1 only: the other ranks would sit in the process group meanwhile
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(ncores)
        tr = O.DannTrainer(O.init_state(11, dann=True), "adamw", lr=1e-3, momentum=0.9, weight_decay=0.01)
        a, b, c = xs.cpu(), ys.cpu(), xt.cpu()
        t1 = time.perf_counter(); tr.step(a, b, c, alpha); warm = time.perf_counter() - t1
        n_cpu = max(1, min(10, int(12.0 / max(warm, 1e-3))))
        t1 = time.perf_counter()
        for _ in range(n_cpu):
            tr.step(a, b, c, alpha)
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(B * n_cpu / cdt, 2), unit="images/sec", cores=torch.get_num_threads(), kind="port", **_host(),
                   sample="%d DANN steps of the same bs=%d source + bs=%d target batch, fp32, PyTorch CPU oracle (%.1f s)" % (n_cpu, B, B, cdt))
    if rank == 0:
        print(json.dumps({
            "metric": "source images/sec RevGrad (DANN) 224x224 train step", "value": round(world * B * args.steps / dt, 1),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "RevGrad (KRN + gradient-reversal domain classifier) step: bs=%d source + bs=%d target images/GPU, "
                                   "AdamW lr 1e-3 wd 0.01 + clip_grad_norm 1.0" % (B, B), "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world, "weights": "random init", "loss_last_step": [float(v) for v in scal.cpu()],
                       "host_enqueue_ms_per_step": round(t_host * 1e3, 4)},
            "roofline": roofline, "cpu_baseline": cpu}))
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_spn(args):
    """SPN train step (trainer.py:114-199): forward, soft-target CE x2, backward, clip_grad_value_(1.0), AdamW.
    N > 1: one process per GPU, every rank its own bs=32 batch, ONE all-reduce of the flat f32 gradient arena (609 MB)
    before the fused clip+update (weak scaling)."""
    from speedplusbaseline_amd.nets.spn import SpacecraftPoseNet
    from speedplusbaseline_amd.optim import SpnOptimizer
    from speedplusbaseline_amd.data import SyntheticSpnLoader
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d): launch with python -m torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    dev = torch.device("cuda", _device_index(local))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _init_dist(dev)
        group = torch.distributed.group.WORLD
    B = 32 if args.batch == 48 else args.batch
    NC = 5000
    torch.manual_seed(2021)
    net = SpacecraftPoseNet(NC, keep_prob=0.5, pretrain=False, precision=args.precision).to(dev).train()
    opt = SpnOptimizer(list(net.parameters()), kind="adamw", lr=1e-4, momentum=0.9, weight_decay=0.0, model=net)
    if world > 1:
        torch.distributed.broadcast(net.flat_parameters(), 0)
        net.invalidate()
    x, yc, yw = next(iter(SyntheticSpnLoader(B, 1, NC, 5, seed=2021 + rank)))
    x, yc, yw = x.to(dev), yc.to(dev), yw.to(dev)

    def one():
        out = net.loss_and_grads(x, yc, yw, world_size=world, group=group, compress_bf16={"1": True, "0": False}.get(os.environ.get("SPB_SPN_BF16_GRADS")),   # default: the library's (on in bf16)
                                 optimizer=None if os.environ.get("SPB_SPN_EARLY_UPDATE") == "0" else opt,
                                 sharded=world > 1 and os.environ.get("SPB_SPN_SHARDED", "1") != "0")   # data parallel: rank-sharded fc update
        opt.step(world_size=world, group=group)
        return out

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    t_host = time.perf_counter() - t0           # the host's share: all K steps enqueued
    sync_all()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    roofline = cpu = None
    ms_step = dt / args.steps * 1e3
    if rank == 0:
        # STEP-level roofline: the step is HBM-bound by the optimizer's pass over the 152 M-element f32 arenas (28 B per parameter + the
        # 16-bit shadow; fp16 adds the inf / nan check's read of g), the fully connected weights streamed by forward and input gradient
        # and their f32 gradient written once: algorithmic bytes per step / step time (spn_step_bytes)
        by_step = spn_step_bytes(net, B, args.precision)
        roofline = _hbm_roof(by_step, ms_step)
        roofline["kernel"] = "whole step (optimizer pass over the arenas + fully connected weight streams + trunk activations)"
        traffic = traffic_src = None
        if B == 32 and NC == 5000:
            for nm in ("r6_spn_%s_pmc_traffic.json" % args.precision, "r5_spn_%s_pmc_traffic.json" % args.precision):
                try:
                    with open(os.path.join(ROOT, "profiles", nm)) as f:
                        j = json.load(f)
                    if "step_hbm_bytes" in j:
                        traffic, traffic_src = j["step_hbm_bytes"], nm
                        break
                except Exception:
                    continue
        roofline["traffic"], roofline["traffic_source"] = traffic, traffic_src
        if not args.bare:
            # the dominant kernel alone: the fused clip + AdamW pass as ONE arena-wide launch, HIP events on the launch stream
            n = net.flat_parameters().numel()
            evs = []
            for _ in range(5):
                net.loss_and_grads(x, yc, yw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); opt.step(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3
            by = n * (28 + (2 if args.precision in ("bf16", "fp16") else 0)) + (4 * n if args.precision == "fp16" else 0)
            ach = by / (us * 1e-6) / 1e9
            roofline["optimizer_alone"] = dict(kernel="optim_step (clip_grad_value + AdamW + 16-bit shadow)" if args.precision != "fp16" else
                                               "amp_check + amp_step + optim_step", achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                                               avg_launch_us=round(us, 1), alg_bytes_per_launch=by,
                                               traffic=_pmc_traffic("optim_step_full", ("r3_spn_pmc_traffic.json", "r2_spn_pmc_traffic.json"))[0])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # N=1 only: the other ranks would sit in the process group meanwhile
        from oracle import spn_oracle as S
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(ncores)
        Bc = 8
        sd = S.init_state(NC)
        xc, ycc, ywc = S.synth_batch(Bc, NC, seed=11)
        masks = S.synth_masks(Bc, seed=5)
        # one reference step (trainer.py:146-185): forward, the two soft cross-entropies, backward, clip_grad_value_(1.0), AdamW
        leaves = [v.clone().requires_grad_(True) for v in sd.values()]
        names = list(sd.keys())
        cpu_opt = torch.optim.AdamW(leaves, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0)

        def cpu_step():
            cpu_opt.zero_grad(set_to_none=True)
            c_, r_ = S.forward(dict(zip(names, leaves)), xc, masks, 0.5)
            loss_, _, _ = S.loss_fn(c_, r_, ycc, ywc)
            loss_.backward()
            torch.nn.utils.clip_grad_value_(leaves, 1.0)
            cpu_opt.step()
        t1 = time.perf_counter()
        cpu_step()      # warm-up
        warm = time.perf_counter() - t1
        nst = max(1, min(40, int(12.0 / max(warm, 1e-3))))     # ~10-15 s of CPU work
        t1 = time.perf_counter()
        for _ in range(nst):
            cpu_step()
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(Bc * nst / cdt, 2), unit="images/sec", cores=torch.get_num_threads(), kind="port", **_host(),
                   sample="%d train steps (forward, backward, clip_grad_value_, AdamW over all 152 M parameters) of a bs=%d 227x227 batch, "
                          "fp32, PyTorch CPU oracle (%.1f s)" % (nst, Bc, cdt))
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec SPN 227x227 bs=32/GPU train step", "value": round(world * B * args.steps / dt, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "SPN (AlexNet trunk + two 5000-class attitude heads) train step, 227x227, bs=%d/GPU, AdamW + "
                                   "clip_grad_value 1.0, dropout 0.5" % B, "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world + (" (fc gradients reduce-scattered, optimizer state of the fully connected layers "
                                                          "sharded by rank, bf16 shadows all-gathered)" if world > 1 and os.environ.get("SPB_SPN_SHARDED", "1") != "0" else ""),
                       "weights": "random init", "loss_last_step": [float(v) for v in out.cpu()],
                       "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 4)},
            "roofline": roofline, "cpu_baseline": cpu}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
