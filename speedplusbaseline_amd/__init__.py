"""speedplusbaseline_amd: the KRN / SPN / DANN training hot path of tpark94/speedplusbaseline on MI355X (gfx950) HIP kernels."""
import os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue serialise.  A data-parallel training step here uses the launch stream, the weight-gradient side stream, the
# communication stream, the update / augmentation stream and RCCL's own: give them a queue each.  Read by the runtime when it
# initialises (the first HIP call), so it has to be in the environment before that; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
