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


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth.tar'):
    """savedir/checkpoint.pth.tar = {epoch, model, state_dict, best_score, optimizer}; bare state_dict to
    model_best.pth.tar when is_best (same files and keys as the reference, utils.py:109-119)"""
    torch.save(states, os.path.join(output_dir, filename))
    logger.info('Checkpoint saved to {}'.format(os.path.join(output_dir, filename)))
    if is_best and 'state_dict' in states:
        torch.save(states['state_dict'], os.path.join(output_dir, 'model_best.pth.tar'))
        logger.info('Best model saved to {}'.format(os.path.join(output_dir, 'model_best.pth.tar')))


def load_checkpoint(checkpoint_file, model, optimizer, device):
    load_dict = torch.load(checkpoint_file, map_location='cpu')
    model.load_state_dict(load_dict['state_dict'], strict=True)
    if optimizer is not None:
        optimizer.load_state_dict(load_dict['optimizer'])
        for state in optimizer.state.values():
            for k, v in state.items():
                if isinstance(v, torch.Tensor):
                    state[k] = v.to(device)
    logger.info('Checkpoint loaded from {} at epoch {}'.format(checkpoint_file, load_dict['epoch']))
    return load_dict['epoch'], load_dict['best_score']


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
