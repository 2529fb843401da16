from speedplusbaseline_amd.nets import get_model, get_optimizer  # noqa: F401
