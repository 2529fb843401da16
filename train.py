"""python train.py --flags...  -- same command line as the reference's train.py:49-160 (config.py), MI355X backend."""
import json
import logging
import os
import os.path as osp

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (speedplusbaseline_amd/__init__.py)

import torch

from config import cfg
from speedplusbaseline_amd.core.trainer import train_single_epoch_krn, train_single_epoch_spn  # noqa: F401 (looked up by name)
from speedplusbaseline_amd.core.inference import valid_krn, valid_spn  # noqa: F401
from speedplusbaseline_amd.data import SyntheticEvalLoader, SyntheticKeypointLoader, SyntheticSpnLoader, synthetic_eval_assets
from speedplusbaseline_amd.nets import get_model, get_optimizer
from speedplusbaseline_amd.parallel import check_replicas, init_job, sync_replicas
from speedplusbaseline_amd.utils import load_checkpoint, save_checkpoint, set_all_seeds, setup_logger

logger = logging.getLogger(__name__)


def _writer(logdir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(logdir)
    except Exception:
        logger.info('tensorboard is not installed: scalar logging disabled')
        return None


def main():
    if not (torch.cuda.is_available() and cfg.use_cuda):
        raise SystemExit("This build runs the training step on an AMD MI355X only (HIP kernels); --no_cuda / CPU execution "
                         "is what the reference implementation is for.")
    # one process per GPU: under `python -m torch.distributed.run --nproc-per-node N train.py ...` this binds cuda:LOCAL_RANK and
    # opens the RCCL process group; a plain `python train.py` is the reference's single cuda:0 process (train.py:50)
    job = init_job()
    device = job.device
    setup_logger('train')
    if not job.is_main:
        logging.getLogger().setLevel(logging.WARNING)      # one progress log per job
    logger.info('Random seed value: {}'.format(cfg.seed))
    if job.world > 1:
        logger.info('Data parallel: %d ranks, batch %d per GPU (global %d), gradient mean over ranks', job.world, cfg.batch_size,
                    cfg.batch_size * job.world)
    set_all_seeds(cfg.seed, cfg, True)          # the same model initialisation on every rank
    os.makedirs(cfg.savedir, exist_ok=True)
    os.makedirs(cfg.logdir, exist_ok=True)
    writer = _writer(cfg.logdir) if job.is_main else None
    if job.is_main:
        with open(osp.join(cfg.savedir, 'config.txt'), 'w') as f:
            json.dump(cfg.__dict__, f, indent=2)
    model = get_model(cfg)
    styleAugmentor = None
    if cfg.randomize_texture:
        from speedplusbaseline_amd.styleaug import StyleAugmentor
        try:
            styleAugmentor = StyleAugmentor(cfg.texture_alpha, device, precision=cfg.styleaug_precision)
        except FileNotFoundError:
            if not getattr(cfg, 'synthetic_batches', 0):
                raise
            # synthetic run: random decoder weights and a synthetic embedding distribution (no checkpoints offline)
            from speedplusbaseline_amd.styleaug import Ghiasi
            styleAugmentor = StyleAugmentor.synthetic(cfg.texture_alpha, device, Ghiasi().state_dict(), precision=cfg.styleaug_precision)
    optimizer = get_optimizer(cfg, model)
    checkpoint_file = osp.join(cfg.savedir, 'checkpoint.pth.tar')
    if cfg.auto_resume and osp.exists(checkpoint_file):
        last_epoch, _ = load_checkpoint(checkpoint_file, model, optimizer, device)
        begin_epoch = best_perf = last_epoch
    else:
        begin_epoch = best_perf = 0
    model = model.to(device)
    sync_replicas(model, job)                   # rank 0's parameters / BatchNorm buffers everywhere
    if job.world > 1:
        # after the (identical) model initialisation the stochastic ops -- SPN dropout, GPU augmentation, style sampling -- draw from a
        # per-rank stream: the same seed on every rank would put correlated noise on the 8 shards of one global batch
        set_all_seeds(job.seed(cfg.seed), cfg, True)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=cfg.lr_decay_step, gamma=cfg.lr_decay_alpha)
    if cfg.synthetic_batches <= 0:
        if cfg.model_name != 'krn':
            raise SystemExit("Only the KRN dataset loader is built (speedplusbaseline_amd.datasets); pass --synthetic_batches N "
                             "to train SPN on synthetic batches.")
        # SPEED+ on disk (train.py:112 of the reference): frames decoded by DataLoader workers, crop + resize + ToTensor +
        # augmentation of the batch on the GPU (speedplusbaseline_amd.transforms)
        from speedplusbaseline_amd.datasets import make_dataloader
        train_loader = make_dataloader(cfg, is_train=True, is_source=True, device=device, rank=job.rank, world=job.world)
    elif cfg.model_name == 'spn':
        train_loader = SyntheticSpnLoader(cfg.batch_size, cfg.synthetic_batches, cfg.num_classes, cfg.num_neighbors, (227, 227), seed=job.seed(cfg.seed))
    else:
        train_loader = SyntheticKeypointLoader(cfg.batch_size, cfg.synthetic_batches, cfg.num_keypoints, cfg.input_shape, seed=job.seed(cfg.seed))
    # per-epoch validation (train.py:135-138 of the reference: every cfg.test_epoch epochs, default -1 = never)
    test_loader = None
    if cfg.test_epoch > 0:
        if cfg.synthetic_batches <= 0:
            raise SystemExit("--test_epoch %d: the SPEED+ test loaders are not part of this build; validate with test.py, or pass "
                             "--synthetic_batches N (synthetic frames, camera, keypoint model and attitude classes)." % cfg.test_epoch)
        corners3D, cameraMatrix, distCoeffs, attClasses = synthetic_eval_assets(cfg.num_keypoints, cfg.num_classes, cfg.seed)
        hw = (227, 227) if cfg.model_name == 'spn' else tuple(cfg.input_shape)
        test_loader = SyntheticEvalLoader(1, cfg.synthetic_batches, corners3D, cameraMatrix, distCoeffs, hw, seed=cfg.seed)
    for epoch in range(begin_epoch, cfg.max_epochs):
        if hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)       # data parallel: a fresh permutation per epoch, the same on every rank
        eval('train_single_epoch_' + cfg.model_name)(epoch + 1, cfg, model, train_loader, optimizer, writer, device,
                                                     styleAugmentor=styleAugmentor, scaler=None)
        lr_scheduler.step()
        if test_loader is not None and (epoch + 1) % cfg.test_epoch == 0 and job.is_main:
            eval('valid_' + cfg.model_name)(epoch + 1, cfg, model, test_loader, cameraMatrix, distCoeffs, corners3D, writer,
                                            device, attClasses)
        if test_loader is not None and (epoch + 1) % cfg.test_epoch == 0:
            job.barrier()       # the other ranks wait HERE (not inside the state-dict / replica-check collectives below) while rank 0 validates
        perf = epoch + 1
        is_best = perf > best_perf
        best_perf = max(best_perf, perf)
        # state_dict() may gather rank-sharded optimizer state (a collective): every rank builds it, rank 0 writes the files
        states = {'epoch': epoch + 1, 'model': cfg.model_name, 'state_dict': model.state_dict(),
                  'best_score': best_perf, 'optimizer': optimizer.state_dict()}
        if job.world > 1:      # collective: the replicas must still be bit-identical after an epoch of exchanged gradients
            check_replicas(model, job)
            logger.info('Data parallel: %d replicas identical after epoch %d', job.world, epoch + 1)
        if job.is_main:
            save_checkpoint(states, is_best, cfg.savedir)
        job.barrier()
    if writer is not None:
        writer.close()
    job.close()


if __name__ == '__main__':
    main()
