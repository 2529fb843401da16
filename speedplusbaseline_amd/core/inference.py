"""Evaluation drivers -- mirror of reference src/core/inference.py:43-249 (same names, positional signatures, meters,
TensorBoard tags, result files and returned dict).  The forward runs on the MI355X (model.eval(), BatchNorm from the running
statistics); the pose post-processing of a whole batch is speedplusbaseline_amd.pose (EPnP without OpenCV, SPEED metrics
with the reference's `speed_q` defect fixed -- SURVEY.md F9).

Differences from the reference, all invisible at its evaluation batch size of 1 (datasets/build.py:51): every sample of a
batch is scored (the reference updates its meters and result lists with the LAST sample of a batch only, inference.py:95-106),
and `speed_score(applyThresh=False)` returns a value instead of raising UnboundLocalError (metrics.py:62)."""
import logging
import os
import os.path as osp
import time

import numpy as np
import torch

from .. import pose
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


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


class _Trackers:
    def __init__(self):
        self.time = AverageMeter('ms')
        self.err_q, self.err_t = AverageMeter('deg'), AverageMeter('m')
        self.speed, self.speed_th, self.acc = AverageMeter('-'), AverageMeter('-'), AverageMeter('%')
        self.all = {"err_q": [], "err_t": [], "speed_raw": [], "speed_mod": []}

    def score(self, q_pr, t_pr, q_gt, t_gt):
        err_q, err_t = pose.error_orientation(q_pr, q_gt), pose.error_translation(t_pr, t_gt)
        raw, acc = pose.speed_score(t_pr, q_pr, t_gt, q_gt, applyThresh=False)
        mod, _ = pose.speed_score(t_pr, q_pr, t_gt, q_gt, applyThresh=True, rotThresh=0.169, posThresh=0.002173)    # inference.py:91-92
        n = len(err_q)
        self.err_q.update(float(err_q.mean()), n); self.err_t.update(float(err_t.mean()), n)
        self.speed.update(float(raw.mean()), n); self.speed_th.update(float(mod.mean()), n); self.acc.update(float(acc.mean()) * 100, n)
        for k, v in (("err_q", err_q), ("err_t", err_t), ("speed_raw", raw), ("speed_mod", mod)):
            self.all[k].extend(float(x) for x in v)

    def finish(self, epoch, writer):
        if writer is not None:
            writer.add_scalar('Valid/err_q [deg]', self.err_q.avg, epoch)
            writer.add_scalar('Valid/err_t [m]', self.err_t.avg, epoch)
            writer.add_scalar('Valid/speed (raw) [-]', self.speed.avg, epoch)
            writer.add_scalar('Valid/speed (thr) [-]', self.speed_th.avg, epoch)
        return {'eR': self.err_q, 'eT': self.err_t, 'speed (raw)': self.speed, 'speed (thr)': self.speed_th}


def _keypts_to_pose(x_pr, y_pr, bbox, corners3D, cameraMatrix, distCoeffs=np.zeros((1, 5))):
    """inference.py:227-249 for a batch ([B,K] keypoints, [B,4] RoIs) or one sample ([K], [4]): (q [.,4], t [.,3])"""
    x, y, b = _np(x_pr), _np(y_pr), _np(bbox)
    single = x.ndim == 1
    if single:
        x, y, b = x[None], y[None], b[None]
    q, t = pose.epnp(corners3D, pose.keypoints_to_pixels(x, y, b), cameraMatrix, distCoeffs)
    return (q[0], t[0]) if single else (q, t)


def valid_krn(epoch, cfg, model, data_loader, cameraMatrix, distCoeffs, corners3D, writer, device, qClass=None):
    ''' Validate KRN model '''
    tr = _Trackers()
    model.eval()
    model3d = pose._Model(corners3D)            # control points / barycentric coordinates of the 3-D model: once per run
    n = len(data_loader)
    for idx, (images, bbox, q_gt, t_gt) in enumerate(data_loader):
        start = time.time()
        B = images.shape[0]
        with torch.no_grad():
            x_pr, y_pr = model(images.to(device))
        q_pr, t_pr = _keypts_to_pose(x_pr, y_pr, bbox, model3d, cameraMatrix, distCoeffs)
        tr.score(q_pr, t_pr, _np(q_gt), _np(t_gt))
        tr.time.update((time.time() - start) * 1000, B)
        report_progress(epoch=epoch, lr=float('nan'), epoch_iter=idx + 1, epoch_size=n, time=tr.time, is_train=False,
                        eT=tr.err_t, eR=tr.err_q, speed=tr.speed, acc=tr.acc)
    performances = tr.finish(epoch, writer)
    logdir = getattr(cfg, 'logdir', None)
    if logdir:                                    # inference.py:128-142
        os.makedirs(logdir, exist_ok=True)
        for fn, key in (('err_q.txt', 'err_q'), ('err_t.txt', 'err_t'), ('speed_raw.txt', 'speed_raw'), ('speed_mod.txt', 'speed_mod')):
            with open(osp.join(logdir, fn), 'w') as f:
                for v in tr.all[key]:
                    f.write('{:.5f}\n'.format(v))
    return performances


def valid_spn(epoch, cfg, model, data_loader, cameraMatrix, distCoeffs, corners3D, writer, device, qClass):
    ''' Valid SPN model '''
    tr = _Trackers()
    model.eval()
    qClass = _np(qClass)
    n = len(data_loader)
    for idx, (images, bbox, q_gt, t_gt) in enumerate(data_loader):
        start = time.time()
        B = images.shape[0]
        with torch.no_grad():
            _, weights = model(images.to(device))
        q_pr, _, _ = pose.spn_attitude(_np(weights.float()), qClass, cfg.num_neighbors)         # inference.py:174-181
        t_pr = pose.compute_position_spn(q_pr, _np(bbox), corners3D, cameraMatrix, distCoeffs)  # inference.py:184
        tr.score(q_pr, t_pr, _np(q_gt), _np(t_gt))
        tr.time.update((time.time() - start) * 1000, B)
        report_progress(epoch=epoch, lr=float('nan'), epoch_iter=idx + 1, epoch_size=n, time=tr.time, is_train=False,
                        eT=tr.err_t, eR=tr.err_q, speed=tr.speed, acc=tr.acc)
    return tr.finish(epoch, writer)
