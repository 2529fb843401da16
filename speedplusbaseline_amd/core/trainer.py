"""Epoch drivers -- mirror of reference src/core/trainer.py:41-199 (same names and positional signatures; train.py
looks them up by name).  The per-batch sequence is the reference's: to(device), optional style augmentation, forward,
zero_grad, backward, clip_grad_norm_(1.0), optimizer step, meters.  With the FusedOptimizer returned by get_optimizer the
whole sequence is one train_step() of HIP launches; any other torch optimizer goes through the generic autograd path."""
import logging
import random
import time

import torch

from torch.nn.utils import clip_grad_norm_

from ..optim import FusedOptimizer
from ..utils import AverageMeter, report_progress

logger = logging.getLogger("Training")


def _texture_coin(cfg, step):
    """the reference's per-batch coin (random.random() < cfg.texture_ratio).  Under data parallelism every rank must take
    the same branch (the decoder adds ~6 ms to the step; a rank that restyles alone stalls the all-reduce of the
    others), so the draw is a function of (seed, step) there."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        from ..parallel import shared_coin
        return shared_coin(step, getattr(cfg, "seed", 2021), cfg.texture_ratio)
    return random.random() < cfg.texture_ratio


class AugLookahead:
    """One-batch lookahead for the style augmentation (trainer.py:63-69 does it inline): the host-to-device copy and the
    Ghiasi decoder of batch i+1 run on a side stream while batch i trains.  The decoder is matrix-core bound (4.7 ms per 48
    images) and the train step a chain of latency-bound launches, so the two overlap almost for free; the augmentation
    does not depend on the weights, so results are unchanged.  The per-batch coin is drawn in batch order."""

    def __init__(self, loader, device, augmentor, coin):
        self.loader, self.device, self.aug, self.coin = loader, device, augmentor, coin
        self.stream = torch.cuda.Stream(device=device)

    def _stage(self, idx, batch):
        images, rest = batch[0], batch[1:]
        restyle = self.coin(idx)                       # drawn here, in batch order
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            images = images.to(self.device, non_blocking=True)
            rest = tuple(t.to(self.device, non_blocking=True) for t in rest)
            if restyle:
                images = self.aug(images)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return images, rest, ev

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        idx = 0
        try:
            nxt = self._stage(idx, next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                batch = next(it)
            except StopIteration:
                batch = None
            idx += 1
            nxt = self._stage(idx, batch) if batch is not None else None     # enqueued BEFORE the current batch trains
            images, rest, ev = cur
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            images.record_stream(main)
            for t in rest:
                t.record_stream(main)
            yield (images,) + rest


def _world():
    """(world_size, group) of the data-parallel job this process belongs to (one process per GPU); (1, None) when single"""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return torch.distributed.get_world_size(), torch.distributed.group.WORLD
    return 1, None


def train_single_epoch_krn(epoch, cfg, model, data_loader, optimizer, writer, device, styleAugmentor=None, scaler=None):
    training_time_meter = AverageMeter('ms')
    loss_x_meter = AverageMeter('-')
    loss_y_meter = AverageMeter('-')
    model.train()
    lr = optimizer.param_groups[-1]['lr']
    fused = isinstance(optimizer, FusedOptimizer) and scaler is None
    world, group = _world()
    n_iter = len(data_loader)
    lookahead = styleAugmentor is not None and torch.device(device).type == "cuda"
    if lookahead:   # trainer.py:68-69, one batch ahead on a side stream
        data_loader = AugLookahead(data_loader, device, styleAugmentor, lambda i: _texture_coin(cfg, epoch * n_iter + i))
    for idx, (images, target) in enumerate(data_loader):
        start = time.time()
        B = images.shape[0]
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        if styleAugmentor is not None and not lookahead and _texture_coin(cfg, epoch * n_iter + idx):   # trainer.py:68-69
            images = styleAugmentor(images)
        if fused:
            lx, ly = optimizer.train_step(images, target, world_size=world, group=group)[1:3].tolist()  # host floats per step, as the reference reports
        else:
            loss, summary = model(images, target)
            optimizer.zero_grad(set_to_none=True)
            if scaler is not None:
                scaler.scale(loss).backward()
                scaler.unscale_(optimizer)
                clip_grad_norm_(model.parameters(), 1.0)
                scaler.step(optimizer)
                scaler.update()
            else:
                loss.backward()
                clip_grad_norm_(model.parameters(), 1.0)
                optimizer.step()
            lx, ly = summary['loss_x'], summary['loss_y']
        training_time_meter.update((time.time() - start) * 1000, B)
        loss_x_meter.update(lx, B)
        loss_y_meter.update(ly, B)
        report_progress(epoch=epoch, lr=lr, epoch_iter=idx + 1, epoch_size=n_iter, time=training_time_meter, is_train=True,
                        loss_x=loss_x_meter, loss_y=loss_y_meter)
    if writer is not None:
        writer.add_scalar('train/loss_x', loss_x_meter.avg, epoch)
        writer.add_scalar('train/loss_y', loss_y_meter.avg, epoch)


def train_single_epoch_spn(epoch, cfg, model, data_loader, optimizer, writer, device, styleAugmentor=None, scaler=None):
    """trainer.py:114-199.  One step = forward, loss = softCE(c, yClasses) + 10 softCE(r, yWeights), backward,
    clip_grad_value_(1.0), optimizer step -- all HIP launches (SpacecraftPoseNet.loss_and_grads + SpnOptimizer.step);
    `scaler`: in fp16 mode (--use_fp16) the GradScaler's arithmetic runs on the device inside loss_and_grads / optimizer.step
    (SpnOptimizer._step_fp16); a torch GradScaler passed here only contributes its hyper-parameters.  bf16 / fp32 need none."""
    training_time_meter = AverageMeter('ms')
    loss_class_meter = AverageMeter('-')
    loss_weight_meter = AverageMeter('-')
    model.train()
    lr = optimizer.param_groups[-1]['lr']
    world, group = _world()
    if scaler is not None and getattr(model, "precision", None) == "fp16" and hasattr(optimizer, "amp_interval"):
        optimizer.amp_growth, optimizer.amp_backoff = float(scaler.get_growth_factor()), float(scaler.get_backoff_factor())
        optimizer.amp_interval = int(scaler.get_growth_interval())
    n_iter = len(data_loader)
    for idx, (images, yClasses, yWeights) in enumerate(data_loader):
        start = time.time()
        B = images.shape[0]
        images = images.to(device, non_blocking=True)
        yClasses = yClasses.to(device, non_blocking=True)
        yWeights = yWeights.to(device, non_blocking=True)
        if styleAugmentor is not None and _texture_coin(cfg, epoch * n_iter + idx):
            images = styleAugmentor(images)
        out = model.loss_and_grads(images, yClasses, yWeights, world_size=world, group=group,   # gradients land in p.grad (no autograd);
                                   optimizer=optimizer)       # the heads' share of the update runs beside the rest of backward
        optimizer.step(world_size=world, group=group)            # [all-reduce,] clip_grad_value_(1.0) + update: one launch
        lc, lr_ = out[1:3].tolist()                              # host floats per step, as the reference reports
        training_time_meter.update((time.time() - start) * 1000, B)
        loss_class_meter.update(lc, B)
        loss_weight_meter.update(lr_, B)
        report_progress(epoch=epoch, lr=lr, epoch_iter=idx + 1, epoch_size=n_iter, time=training_time_meter, is_train=True,
                        loss_c=loss_class_meter, loss_r=loss_weight_meter)
    model.join_updates()     # the last step's update of the heads (it ran on beside this loop's tail) before anyone reads parameters
    if writer is not None:
        writer.add_scalar('train/loss_c', loss_class_meter.avg, epoch)
        writer.add_scalar('train/loss_r', loss_weight_meter.avg, epoch)
