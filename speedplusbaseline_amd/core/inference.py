"""Evaluation drivers -- the FORWARD part of reference src/core/inference.py:43-142 (model.eval(), no_grad forward,
keypoints back on the host).  The per-image pose solve (EPnP via OpenCV) and SPEED metrics that follow in the
reference are CPU post-processing outside the hot path ("next"); they run here only when cv2 and the
reference's utils are importable, otherwise the function returns the raw keypoints."""
import logging
import time

import torch

from ..utils import AverageMeter, report_progress

logger = logging.getLogger("Testing")


def predict_keypoints(model, data_loader, device, max_batches=None):
    """list of (xc [B,K], yc [B,K]) CPU tensors, normalised to the network input frame"""
    model.eval()
    out = []
    with torch.no_grad():
        for idx, batch in enumerate(data_loader):
            images = batch[0] if isinstance(batch, (tuple, list)) else batch
            out.append(model(images.to(device)))
            if max_batches is not None and idx + 1 >= max_batches:
                break
    return out


def valid_krn(epoch, cfg, model, data_loader, cameraMatrix, distCoeffs, corners3D, writer, device, qClass=None):
    time_meter = AverageMeter('ms')
    model.eval()
    preds = []
    n = len(data_loader)
    with torch.no_grad():
        for idx, batch in enumerate(data_loader):
            start = time.time()
            images = batch[0]
            xc, yc = model(images.to(device))
            preds.append((xc, yc) + tuple(batch[1:]))
            time_meter.update((time.time() - start) * 1000, images.shape[0])
            report_progress(epoch=epoch, lr=float('nan'), epoch_iter=idx + 1, epoch_size=n, time=time_meter, is_train=False)
    try:
        import cv2  # noqa: F401
    except ImportError:
        logger.warning("cv2 is not installed: keypoints->pose (EPnP) and SPEED metrics are skipped; returning keypoints")
        return preds
    raise NotImplementedError("pose post-processing (EPnP + SPEED score) is the next row after the hot path (DESIGN.md)")


def valid_spn(*args, **kwargs):
    raise NotImplementedError("SPN evaluation has no HIP path yet (DESIGN.md: scope / next rows)")
