"""python test.py --pretrained model_best.pth.tar ...  -- evaluation CLI (reference test.py:42-91): forward pass on the
MI355X, pose post-processing and SPEED metrics of every batch in speedplusbaseline_amd.pose, averages written to
logdir/resultfn in the reference's format."""
import logging
import os
import os.path as osp

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (speedplusbaseline_amd/__init__.py)

import torch

from config import cfg
from speedplusbaseline_amd.core.inference import valid_krn, valid_spn  # noqa: F401 (looked up by name)
from speedplusbaseline_amd.data import SyntheticEvalLoader, synthetic_eval_assets
from speedplusbaseline_amd.nets import get_model
from speedplusbaseline_amd.utils import set_all_seeds, setup_logger

logger = logging.getLogger(__name__)


def main():
    if not (torch.cuda.is_available() and cfg.use_cuda):
        raise SystemExit("This build runs on an AMD MI355X only (HIP kernels); --no_cuda / CPU execution is what the reference "
                         "implementation is for.")
    device = torch.device('cuda:0')
    setup_logger('test')
    os.makedirs(cfg.logdir, exist_ok=True)
    logger.info('Random seed value: {}'.format(cfg.seed))
    set_all_seeds(cfg.seed, cfg, True)
    model = get_model(cfg)
    if cfg.pretrained and osp.exists(cfg.pretrained):
        model.load_state_dict(torch.load(cfg.pretrained, map_location='cpu'), strict=True)
        logger.info('Model loaded from {}'.format(cfg.pretrained))
    model = model.to(device)
    if cfg.synthetic_batches <= 0:
        raise SystemExit("The SPEED+ test loaders are not part of this build; pass --synthetic_batches N (random frames and poses, "
                         "synthetic camera / keypoint model / attitude classes).")
    corners3D, cameraMatrix, distCoeffs, attClasses = synthetic_eval_assets(cfg.num_keypoints, cfg.num_classes, cfg.seed)
    hw = (227, 227) if cfg.model_name == 'spn' else tuple(cfg.input_shape)
    test_loader = SyntheticEvalLoader(1, cfg.synthetic_batches, corners3D, cameraMatrix, distCoeffs, hw, seed=cfg.seed)   # batch 1: datasets/build.py:51
    assert attClasses.shape[0] == cfg.num_classes, 'Number of classes not matching.'
    performances = eval('valid_' + cfg.model_name)(0, cfg, model, test_loader, cameraMatrix, distCoeffs, corners3D, None, device, attClasses)
    writefn = osp.join(cfg.logdir, cfg.resultfn or 'results.txt')   # the reference's default '' names the directory itself (IsADirectoryError)
    with open(writefn, 'w') as f:
        for metric in performances:
            msg = metric + ': {:.5f} [' + performances[metric].unit + ']\n'
            f.write(msg.format(performances[metric].avg))
    logger.info('Test results written to {}'.format(writefn))


if __name__ == '__main__':
    main()
