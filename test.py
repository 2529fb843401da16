"""python test.py --pretrained model_best.pth.tar ...  -- evaluation CLI (reference test.py:42-91): forward pass on the
MI355X; the OpenCV pose solve + SPEED metrics that follow in the reference are CPU post-processing outside this build."""
import logging
import os.path as osp

import torch

from config import cfg
from speedplusbaseline_amd.core.inference import predict_keypoints
from speedplusbaseline_amd.data import SyntheticKeypointLoader
from speedplusbaseline_amd.nets import get_model
from speedplusbaseline_amd.utils import setup_logger

logger = logging.getLogger(__name__)


def main():
    if not (torch.cuda.is_available() and cfg.use_cuda):
        raise SystemExit("This build runs on an AMD MI355X only (HIP kernels).")
    device = torch.device('cuda:0')
    setup_logger('test')
    model = get_model(cfg)
    if cfg.pretrained and osp.exists(cfg.pretrained):
        model.load_state_dict(torch.load(cfg.pretrained, map_location='cpu'), strict=True)
        logger.info('   - Pretrained model loaded from {}'.format(cfg.pretrained))
    model = model.to(device)
    if cfg.synthetic_batches <= 0:
        raise SystemExit("The SPEED+ dataset pipeline is not part of this build; pass --synthetic_batches N.")
    loader = SyntheticKeypointLoader(1, cfg.synthetic_batches, cfg.num_keypoints, cfg.input_shape, seed=cfg.seed)
    preds = predict_keypoints(model, loader, device)
    logger.info('predicted keypoints for %d images; first: x=%s', len(preds), preds[0][0][0, :3].tolist())
    if cfg.resultfn:
        torch.save(preds, cfg.resultfn)


if __name__ == '__main__':
    main()
