"""Model / optimizer factory -- mirror of reference src/nets/build.py:39-78 (same names, same cfg fields)."""
import logging

from .park2019 import KeypointRegressionNet
from .revgrad import RevGrad
from .spn import SpacecraftPoseNet
from ..optim import FusedOptimizer

logger = logging.getLogger(__name__)


def _precision(cfg):
    p = getattr(cfg, "precision", None)
    if p:
        return p
    if getattr(cfg, "fp16", False):
        # reference: torch.cuda.amp autocast (float16) + GradScaler (train.py:101-104, trainer.py:73-94).  The MI355X kernels
        # compute in bfloat16 with f32 accumulation / master weights / statistics instead: same 16-bit operand width and
        # matrix-core rate, f32's exponent range, so no loss scaling (and no inf-check host sync) is needed.
        global _warned_fp16
        if not _warned_fp16:
            logger.warning("--use_fp16: float16 autocast + GradScaler is replaced by bfloat16 compute (no loss scaling); "
                           "pass --precision bf16 to select it explicitly")
            _warned_fp16 = True
        return "bf16"
    return None


_warned_fp16 = False


def get_model(cfg):
    assert cfg.model_name == 'krn' or cfg.model_name == 'spn', 'Model name must be either krn or spn'
    if not cfg.dann:
        if cfg.model_name == 'krn':
            model = KeypointRegressionNet(cfg.num_keypoints, precision=_precision(cfg))
            logger.info('KRN created')
        else:
            try:
                model = SpacecraftPoseNet(cfg.num_classes, pretrain=True, precision=_precision(cfg))   # build.py:48
            except FileNotFoundError:
                if not getattr(cfg, 'synthetic_batches', 0):
                    raise
                # synthetic run: the AlexNet npy (checkpoints/pretrained/bvlc_alexnet.npy) is not available offline
                model = SpacecraftPoseNet(cfg.num_classes, pretrain=False, precision=_precision(cfg))
            logger.info('SPN created')
    else:
        model = RevGrad(cfg.num_keypoints, precision=_precision(cfg))
        logger.info('RevGrad created with {}'.format(cfg.model_name))
    logger.info('   - Number of total parameters:     {:,}'.format(sum(p.numel() for p in model.parameters())))
    logger.info('   - Number of trainable parameters: {:,}'.format(sum(p.numel() for p in model.parameters() if p.requires_grad)))
    return model


def get_optimizer(cfg, model):
    """sgd / rmsprop / adam / adamw with cfg.momentum doubling as RMSprop alpha and Adam beta1 (build.py:60-78).
    Returns a torch.optim.Optimizer subclass whose update (and the trainers' clip_grad_norm_) is one fused HIP pass
    over the model's flat parameter arena."""
    if cfg.optimizer not in ('sgd', 'rmsprop', 'adam', 'adamw'):
        raise ValueError('unknown optimizer %r' % cfg.optimizer)
    params = [p for p in model.parameters() if p.requires_grad]
    if isinstance(model, SpacecraftPoseNet):
        from ..optim import SpnOptimizer
        optimizer = SpnOptimizer(params, kind=cfg.optimizer, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                                 model=model, clip_value=1.0)
        logger.info('Optimizer created: {}'.format(cfg.optimizer))
        return optimizer
    optimizer = FusedOptimizer(params, kind=cfg.optimizer, lr=cfg.lr, momentum=cfg.momentum,
                               weight_decay=cfg.weight_decay, model=model)
    logger.info('Optimizer created: {}'.format(cfg.optimizer))
    return optimizer
