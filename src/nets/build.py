from speedplusbaseline_amd.nets.build import *  # noqa: F401,F403
from speedplusbaseline_amd.nets.build import get_model, get_optimizer  # noqa: F401
