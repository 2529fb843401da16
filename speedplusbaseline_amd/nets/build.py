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
        if p == "fp16" and getattr(cfg, "dann", False):
            raise ValueError("--precision fp16 exists for SPN and KRN (IEEE-half kernels + device-side loss scaling); RevGrad / DANN runs "
                             "--precision bf16 or fp32 (the reference's adapt.py has no mixed precision at all: adapt.py:99-101)")
        return p
    if getattr(cfg, "fp16", False):
        # reference: torch.cuda.amp autocast (float16) + GradScaler (train.py:101-104, trainer.py:73-94, 146-181).
        #   SPN: real float16 -- IEEE-half activations / weight shadows on v_mfma_f32_16x16x32_f16 (libspb_hip_f16.so), f32
        #        master weights and accumulation, GradScaler's dynamic loss scale kept on the device (spb_amp_*).
        #   KRN (round 5): the same -- the IEEE-half build of the KRN kernels, GradScaler's state on the device (FusedTrainStep._update_fp16).
        #        --precision bf16 selects bfloat16 compute instead (BASELINE configs[1]: same operand width and matrix-core rate, f32's
        #        exponent range, no loss scaling; 8x coarser mantissa).
        #   RevGrad / DANN: bfloat16 (the reference's adapt.py runs without mixed precision, adapt.py:99-101).
        global _warned_fp16
        if not cfg.dann:
            if not _warned_fp16:
                logger.info("--use_fp16: %s runs in float16 with device-side dynamic loss scaling (GradScaler defaults: "
                            "init 65536, x2 after 2000 clean steps, x0.5 and a skipped step on overflow)", cfg.model_name.upper())
                _warned_fp16 = True
            return "fp16"
        if not _warned_fp16:
            logger.warning("--use_fp16: for RevGrad / DANN float16 autocast + GradScaler is replaced by bfloat16 compute (no loss "
                           "scaling); pass --precision bf16 to select it explicitly")
            _warned_fp16 = True
        return "bf16"
    return None


_warned_fp16 = False


def get_model(cfg):
    assert cfg.model_name == 'krn' or cfg.model_name == 'spn', 'Model name must be either krn or spn'
    if not cfg.dann:
        if cfg.model_name == 'krn':
            model = KeypointRegressionNet(cfg.num_keypoints, precision=_precision(cfg), deterministic=getattr(cfg, "deterministic", False))
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
        model = RevGrad(cfg.num_keypoints, precision=_precision(cfg), deterministic=getattr(cfg, "deterministic", False))
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
