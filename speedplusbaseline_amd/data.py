"""Synthetic loaders (the SPEED+ dataset and the torchvision/PIL pipeline of reference src/datasets are out of scope;
BASELINE.json measures on synthetic 224x224 batches)."""
import torch


class SyntheticKeypointLoader:
    """yields (images [B,3,H,W] U[0,1), keypoints [B,2,K] U[0,1)) -- the value ranges of transforms.py:157-159,192-196"""

    def __init__(self, batch_size, n_batches, num_keypoints=11, hw=(224, 224), labels=True, seed=2021):
        self.B, self.n, self.K, self.hw, self.labels, self.seed = batch_size, n_batches, num_keypoints, tuple(hw), labels, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            x = torch.rand(self.B, 3, self.hw[0], self.hw[1], generator=g)
            if self.labels:
                yield x, torch.rand(self.B, 2, self.K, generator=g)
            else:
                yield x
