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


class SyntheticSpnLoader:
    """yields (images [B,3,227,227] U[0,1), yClasses, yWeights [B,num_classes]): num_neighbors random attitude classes per
    image with probability 1/num_neighbors each and normalised random weights (the layout of SPNDataset.py:83-92)"""

    def __init__(self, batch_size, n_batches, num_classes=5000, num_neighbors=5, hw=(227, 227), seed=2021):
        self.B, self.n, self.C, self.nn, self.hw, self.seed = batch_size, n_batches, num_classes, num_neighbors, tuple(hw), seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            x = torch.rand(self.B, 3, self.hw[0], self.hw[1], generator=g)
            idx = torch.randint(0, self.C, (self.B, self.nn), generator=g)
            w = torch.rand(self.B, self.nn, generator=g) + 0.1
            yc = torch.zeros(self.B, self.C).scatter_add_(1, idx, torch.full((self.B, self.nn), 1.0 / self.nn))
            yw = torch.zeros(self.B, self.C).scatter_add_(1, idx, w / w.sum(1, keepdim=True))
            yield x, yc, yw
