"""Training-loop utilities the hot path touches: meters, progress line, checkpoint I/O, seeding.
Re-authored counterparts of reference src/utils/utils.py:43-135,289-312 (same names and file formats); the pose /
camera / quaternion helpers of that file belong to the evaluation post-processing (out of scope)."""
import logging
import os
import random
import sys

import numpy as np
import torch

logger = logging.getLogger(__name__)


class AverageMeter(object):
    """running value / sum / count / average with a unit label"""

    def __init__(self, unit='-'):
        self.unit = unit
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count else 0


def setup_logger(phase):
    logging.basicConfig(format='%(asctime)-15s %(message)s', datefmt='%Y/%m/%d %H:%M:%S')
    root = logging.getLogger()
    root.setLevel(logging.INFO)
    return root


def report_progress(epoch, lr, epoch_iter, epoch_size, time, is_train=True, **kwargs):
    width = 30
    frac = float(epoch_iter / epoch_size)
    bar = '█' * int(round(frac * width))
    msg = ("\rTraining " if is_train else "\rTesting ")
    msg += "{:03d} (lr: {:.5f}): {:04d}/{:04d} [{}{:03d}%] [{:.0f} ({:.0f}) ms] ".format(
        epoch, lr, epoch_iter, epoch_size, bar + ' ' * (width - len(bar)), round(frac * 100), time.val, time.avg)
    for key, item in kwargs.items():
        if item is not None:
            msg += '{}: {:.2f} ({:.2f}) [{}] '.format(key, item.val, item.avg, item.unit)
    sys.stdout.write(msg)
    if epoch_iter == epoch_size:
        sys.stdout.write('\n')
    sys.stdout.flush()


CHECKPOINT_KEYS = ('epoch', 'model', 'state_dict', 'best_score', 'optimizer')     # the file format of utils.py:109-119
BEST_MODEL_FILE = 'model_best.pth.tar'


def _write_atomically(obj, path):
    """torch.save to a temporary name in the same directory, then rename: an interrupted run never leaves a truncated
    checkpoint where auto-resume (train.py:86-94) would pick it up"""
    tmp = '%s.tmp%d' % (path, os.getpid())
    torch.save(obj, tmp)
    os.replace(tmp, path)
    return path


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth.tar'):
    """Reference contract (utils.py:109-119): `output_dir/filename` holds the whole `states` dict
    ({epoch, model, state_dict, best_score, optimizer}); when `is_best`, the bare model state_dict additionally goes to
    `output_dir/model_best.pth.tar`, which is what test.py --pretrained and adapt.py --pretrained read."""
    written = [_write_atomically(states, os.path.join(output_dir, filename))]
    weights = states.get('state_dict') if is_best else None
    if weights is not None:
        written.append(_write_atomically(weights, os.path.join(output_dir, BEST_MODEL_FILE)))
    for path, what in zip(written, ('Checkpoint', 'Best model')):
        logger.info('%s saved to %s', what, path)


def _optimizer_state_to(optimizer, device):
    for per_param in optimizer.state.values():
        moved = {k: v.to(device) for k, v in per_param.items() if torch.is_tensor(v)}
        per_param.update(moved)


def load_checkpoint(checkpoint_file, model, optimizer, device):
    """Reference contract (utils.py:121-135): strict load of the model weights from a checkpoint read onto the CPU, optimizer
    state restored and placed on `device`; returns (epoch, best_score).  The fused optimizers of this build keep their moments in
    flat arenas of their own (FusedOptimizer / SpnOptimizer.load_state_dict), so the per-parameter move only concerns plain torch
    optimizers."""
    ckpt = torch.load(checkpoint_file, map_location='cpu')
    missing = [k for k in ('state_dict', 'epoch', 'best_score') if k not in ckpt]
    if missing:
        raise KeyError('%s is not a training checkpoint (no %s)' % (checkpoint_file, ', '.join(missing)))
    model.load_state_dict(ckpt['state_dict'], strict=True)
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optimizer'])
        _optimizer_state_to(optimizer, device)
    logger.info('Checkpoint loaded from %s at epoch %s', checkpoint_file, ckpt['epoch'])
    return ckpt['epoch'], ckpt['best_score']


def set_all_seeds(seed, cfg, use_cuda):
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if use_cuda:
        torch.cuda.manual_seed(seed)


def num_total_parameters(model):
    return sum(p.numel() for p in model.parameters())


def num_trainable_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
