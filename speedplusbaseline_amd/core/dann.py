"""DANN epoch driver -- mirror of reference src/core/dann.py:38-117."""
import math
import time

import torch
import torch.nn as nn
from torch.nn.utils import clip_grad_norm_

from ..optim import FusedOptimizer
from ..utils import AverageMeter, report_progress
from .trainer import _world


def dann_alpha(idx, epoch, n_batches, max_epochs):
    """domain-loss factor schedule 2/(1+exp(-10p)) - 1, p = progress in [0,1] (dann.py:77-78)"""
    p = float(idx + epoch * n_batches) / max_epochs / n_batches
    return 2. / (1. + math.exp(-10 * p)) - 1


def train_dann_single_epoch_krn(epoch, cfg, model, dataloader_source, dataloader_target, optimizer, writer, device, scaler=None):
    training_time_meter = AverageMeter('ms')
    loss_pose_meter = AverageMeter('-')
    loss_source_meter = AverageMeter('-')
    loss_target_meter = AverageMeter('-')
    model.train()
    lr = optimizer.param_groups[-1]['lr']
    n_batches = min(len(dataloader_source), len(dataloader_target))
    fused = isinstance(optimizer, FusedOptimizer)
    world, group = _world()     # data parallel (one process per GPU): the summed gradient arena is averaged before the clip
    for idx, ((source, label), target) in enumerate(zip(dataloader_source, dataloader_target)):
        B = source.size(0)
        ts = time.time()
        source = source.to(device, non_blocking=True)
        label = label.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        alpha = dann_alpha(idx, epoch, n_batches, cfg.max_epochs)
        if fused:
            s = optimizer.train_step(source, label, target_images=target, alpha=alpha, world_size=world, group=group).tolist()
            l_pose, l_src, l_tgt = s[0], s[3], s[4]
        else:
            optimizer.zero_grad(set_to_none=True)
            (loss_pose_source, sm), domain_source_pred = model(source, y=label, alpha=alpha)
            loss_domain_source = nn.functional.binary_cross_entropy_with_logits(
                domain_source_pred, torch.ones(B, device=device), reduction='mean')
            _, domain_target_pred = model(target, alpha=alpha)
            loss_domain_target = nn.functional.binary_cross_entropy_with_logits(
                domain_target_pred, torch.zeros(B, device=device), reduction='mean')
            (loss_pose_source + loss_domain_source + loss_domain_target).backward()
            clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
            l_pose, l_src, l_tgt = float(loss_pose_source), float(loss_domain_source), float(loss_domain_target)
        training_time_meter.update((time.time() - ts) * 1000, B)
        loss_pose_meter.update(l_pose, B)
        loss_source_meter.update(l_src, B)
        loss_target_meter.update(l_tgt, B)
        report_progress(epoch=epoch, lr=lr, epoch_iter=idx + 1, epoch_size=n_batches, time=training_time_meter, is_train=True,
                        loss_pose=loss_pose_meter, loss_source=loss_source_meter, loss_target=loss_target_meter)
    if writer is not None:
        writer.add_scalar('train/loss_pose', loss_pose_meter.avg, epoch)
        writer.add_scalar('train/loss_source', loss_source_meter.avg, epoch)
        writer.add_scalar('train/loss_target', loss_target_meter.avg, epoch)
