"""python adapt.py --perform_dann ...  -- same command line as the reference's adapt.py:47-148, MI355X backend."""
import logging
import os
import os.path as osp

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (speedplusbaseline_amd/__init__.py)

import torch

from config import cfg
from speedplusbaseline_amd.core.dann import train_dann_single_epoch_krn
from speedplusbaseline_amd.core.inference import valid_krn
from speedplusbaseline_amd.data import SyntheticEvalLoader, SyntheticKeypointLoader, synthetic_eval_assets
from speedplusbaseline_amd.nets import get_model, get_optimizer
from speedplusbaseline_amd.parallel import check_replicas, init_job, sync_replicas
from speedplusbaseline_amd.utils import load_checkpoint, save_checkpoint, set_all_seeds, setup_logger

logger = logging.getLogger(__name__)


def main():
    assert cfg.dann and cfg.model_name == 'krn', 'DANN (--perform_dann) is only for KRN'
    if not (torch.cuda.is_available() and cfg.use_cuda):
        raise SystemExit("This build runs on an AMD MI355X only (HIP kernels).")
    job = init_job()       # one process per GPU under torch.distributed.run (RCCL); a plain `python adapt.py` is the reference's cuda:0 process
    device = job.device
    setup_logger('adapt')
    if not job.is_main:
        logging.getLogger().setLevel(logging.WARNING)
    set_all_seeds(2021, cfg, True)  # the reference pins 2021 here (adapt.py:55)
    os.makedirs(cfg.savedir, exist_ok=True)
    os.makedirs(cfg.logdir, exist_ok=True)   # valid_krn (--test_epoch) writes its result files there
    model = get_model(cfg)
    optimizer = get_optimizer(cfg, model)
    checkpoint_file = osp.join(cfg.savedir, 'checkpoint.pth.tar')
    begin_epoch = 0
    if cfg.auto_resume and osp.exists(checkpoint_file):
        begin_epoch, _ = load_checkpoint(checkpoint_file, model, optimizer, device)
    elif cfg.pretrained and osp.exists(cfg.pretrained):
        model.net.load_state_dict(torch.load(cfg.pretrained, map_location='cpu'), strict=True)
    model = model.to(device)
    sync_replicas(model, job)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=cfg.lr_decay_step, gamma=cfg.lr_decay_alpha)
    if cfg.synthetic_batches <= 0:
        raise SystemExit("The SPEED+ dataset pipeline is not part of this build; pass --synthetic_batches N.")
    src = SyntheticKeypointLoader(cfg.batch_size, cfg.synthetic_batches, cfg.num_keypoints, cfg.input_shape, seed=job.seed(cfg.seed))
    tgt = SyntheticKeypointLoader(cfg.batch_size, cfg.synthetic_batches, cfg.num_keypoints, cfg.input_shape, labels=False,
                                  seed=job.seed(cfg.seed) + 1)
    target_test_loader = None
    if cfg.test_epoch > 0:      # adapt.py:124-126 of the reference: the target-domain test set every cfg.test_epoch epochs
        corners3D, cameraMatrix, distCoeffs, _ = synthetic_eval_assets(cfg.num_keypoints, cfg.num_classes, cfg.seed)
        target_test_loader = SyntheticEvalLoader(1, cfg.synthetic_batches, corners3D, cameraMatrix, distCoeffs,
                                                 tuple(cfg.input_shape), seed=cfg.seed)
    for epoch in range(begin_epoch, cfg.max_epochs):
        train_dann_single_epoch_krn(epoch, cfg, model, src, tgt, optimizer, None, device)
        lr_scheduler.step()
        if target_test_loader is not None and (epoch + 1) % cfg.test_epoch == 0 and job.is_main:
            valid_krn(epoch, cfg, model, target_test_loader, cameraMatrix, distCoeffs, corners3D, None, device, None)
        if target_test_loader is not None and (epoch + 1) % cfg.test_epoch == 0:
            job.barrier()       # the other ranks wait here while rank 0 validates (not inside the collectives below)
        states = {'epoch': epoch + 1, 'model': cfg.model_name, 'state_dict': model.state_dict(),
                  'best_score': epoch + 1, 'optimizer': optimizer.state_dict()}
        if job.world > 1:      # collective: the replicas must still be bit-identical after an epoch of exchanged gradients
            check_replicas(model, job)
            logger.info('Data parallel: %d replicas identical after epoch %d', job.world, epoch + 1)
        if job.is_main:
            save_checkpoint(states, True, cfg.savedir)
        job.barrier()
    job.close()


if __name__ == '__main__':
    main()
