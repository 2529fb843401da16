from speedplusbaseline_amd.utils import *  # noqa: F401,F403
from speedplusbaseline_amd.utils import AverageMeter, report_progress, save_checkpoint, load_checkpoint, set_all_seeds, setup_logger  # noqa: F401
