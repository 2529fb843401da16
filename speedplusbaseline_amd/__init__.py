"""speedplusbaseline_amd: the KRN / SPN / DANN training hot path of tpark94/speedplusbaseline on MI355X (gfx950) HIP kernels."""
import os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue serialise.  A data-parallel training step here uses the launch stream, the weight-gradient side stream, the
# communication stream, the update / augmentation stream and RCCL's own: give them a queue each.  Read by the runtime when it
# initialises (the first HIP call), so it has to be in the environment before that; an explicit setting wins.
# The CLI entry points (train.py / adapt.py / test.py / bench.py) export it before `import torch`; a host that imported torch
# and initialised HIP first keeps the runtime's 4 queues -- say so instead of silently serialising streams.
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("speedplusbaseline_amd: HIP was initialised before GPU_MAX_HW_QUEUES=8 could be set; streams may share hardware "
                "queues (export GPU_MAX_HW_QUEUES=8 before starting Python, as train.py / adapt.py / test.py do)")
