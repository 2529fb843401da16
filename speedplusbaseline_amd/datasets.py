"""Dataset side of the GPU input pipeline: the reference's CSV-driven datasets (src/datasets/Park2019KRNDataset.py:45-121,
src/datasets/SPNDataset.py, src/datasets/build.py:35-66) split where the work moves to the GPU.

DataLoader workers only read the CSV row and decode the frame (`Image.open`, kept single-channel when the file is grey: the
reference's convert('RGB') of a grey frame is three equal bands, which the resize kernel reproduces from one); the main
process hands the batch of frames to speedplusbaseline_amd.transforms.GpuBatchTransform, which crops on the host and resizes /
converts / augments on the GPU.  `make_dataloader(cfg, ...)` has the reference's signature and yields the same tuples
(images are already on the GPU)."""
import os.path as osp

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .transforms import build_transforms


class Park2019KRNFrames(Dataset):
    """CSV columns (no header): image path, xmin, xmax, ymin, ymax [pix], q0..q3, t1..t3, kx1, ky1, ..., kx11, ky11 [pix]
    (Park2019KRNDataset.py:39-44).  Returns the decoded frame and the raw labels; no transform."""

    def __init__(self, cfg, is_train=True, is_source=True, load_labels=True):
        import pandas as pd
        self.is_train, self.load_labels = is_train, load_labels
        self.root = osp.join(cfg.dataroot, cfg.dataname)
        self.num_keypts = cfg.num_keypoints
        if is_train and is_source:
            if not load_labels:
                raise AssertionError("source-domain training needs labels")
            csvfile = osp.join(self.root, cfg.train_domain, 'splits_' + cfg.model_name, cfg.train_csv)
        else:
            if is_train and load_labels:
                raise AssertionError("target-domain training images carry no labels")
            csvfile = osp.join(self.root, cfg.test_domain, 'splits_' + cfg.model_name, cfg.test_csv)
        self.csv = pd.read_csv(csvfile, header=None)

    def __len__(self):
        return len(self.csv)

    def __getitem__(self, index):
        from PIL import Image
        row = self.csv.iloc[index]
        img = Image.open(osp.join(self.root, row[0]))
        frame = np.asarray(img if img.mode in ("L", "RGB") else img.convert("RGB"))
        bbox = np.array(row[1:5], dtype=np.float32)
        if self.is_train and self.load_labels:
            keypts = np.transpose(np.reshape(np.array(row[12:12 + 2 * self.num_keypts], dtype=np.float32), (self.num_keypts, 2)))
        else:
            keypts = np.zeros((2, self.num_keypts), dtype=np.float32)
        q = np.array(row[5:9], dtype=np.float32)
        t = np.array(row[9:12], dtype=np.float32)
        return frame, bbox, keypts, q, t


class GpuTransformLoader:
    """iterates a DataLoader of raw frames and applies the batched GPU transform; yields what the reference's loaders yield:
    train + labels: (images, keypts); train without labels: images; test: (images, bbox, q_gt, t_gt)"""

    def __init__(self, loader, transform, is_train, load_labels):
        self.loader, self.transform, self.is_train, self.load_labels = loader, transform, is_train, load_labels

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for samples in self.loader:
            frames, bboxes, keypts, qs, ts = zip(*samples)
            images, boxes, kps = self.transform(list(frames), list(bboxes), [k.copy() for k in keypts])
            if self.is_train:
                yield (images, kps) if self.load_labels else images
            else:
                yield images, boxes, torch.from_numpy(np.stack(qs)), torch.from_numpy(np.stack(ts))


def make_dataloader(cfg, is_train=True, is_source=True, load_labels=True, device="cuda", rank=0, world=1):
    """src/datasets/build.py:48-66 (batch size / shuffle / workers / drop_last as there), KRN datasets.  world > 1 (data parallel,
    one process per GPU): every rank draws cfg.batch_size frames per step from its own 1/world shard of an epoch's permutation
    (DistributedSampler seeded with cfg.seed; call loader.set_epoch(e) to reshuffle)."""
    if cfg.model_name != 'krn':
        raise NotImplementedError("the SPN dataset needs the attitude-class files of the reference checkout; only the KRN loader is built")
    dataset = Park2019KRNFrames(cfg, is_train, is_source, load_labels)
    sampler = None
    if world > 1 and is_train:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=int(getattr(cfg, "seed", 0)), drop_last=True)
    loader = DataLoader(dataset, batch_size=cfg.batch_size if is_train else 1, shuffle=is_train and sampler is None, sampler=sampler,
                        num_workers=cfg.num_workers if is_train else 1, collate_fn=list, drop_last=True)
    transform = build_transforms(cfg.model_name, cfg.input_shape, is_train=is_train, device=device)
    out = GpuTransformLoader(loader, transform, is_train, load_labels)
    out.set_epoch = sampler.set_epoch if sampler is not None else (lambda epoch: None)
    return out
