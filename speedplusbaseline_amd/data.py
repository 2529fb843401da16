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


def synthetic_eval_assets(num_keypoints=11, num_classes=5000, seed=2021):
    """stand-ins for the evaluation assets that are not redistributed (tangoPoints.mat, camera.json, attitudeClasses.mat):
    a ~1.2 m `num_keypoints`-point model [K,3], a 1920x1200 pinhole camera with mild lens distortion, unit class quaternions"""
    import numpy as np
    g = np.random.RandomState(seed)
    corners3D = g.uniform(-0.6, 0.6, size=(num_keypoints, 3)).astype(np.float32)
    cameraMatrix = np.array([[2988.58, 0.0, 960.0], [0.0, 2988.34, 600.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    distCoeffs = np.array([-0.2238, 0.5141, -0.0006, -0.0002, -0.1313], dtype=np.float32)
    q = g.normal(size=(num_classes, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return corners3D, cameraMatrix, distCoeffs, q.astype(np.float32)


class SyntheticEvalLoader:
    """yields (images, bbox [B,4] = xmin, xmax, ymin, ymax, q_gt [B,4], t_gt [B,3]) like the reference's test loaders
    (Park2019KRNDataset.py:101-109 / SPNDataset.py): random frames, random poses in front of the camera, the box of the
    projected model.  Random-weight networks score badly on it; it exists to run the evaluation path end to end."""

    def __init__(self, batch_size, n_batches, corners3D, cameraMatrix, distCoeffs, hw=(224, 224), seed=2021):
        self.B, self.n, self.hw, self.seed = batch_size, n_batches, tuple(hw), seed
        self.assets = (corners3D, cameraMatrix, distCoeffs)

    def __len__(self):
        return self.n

    def __iter__(self):
        import numpy as np
        from . import pose
        pts, K, dist = self.assets
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            x = torch.rand(self.B, 3, self.hw[0], self.hw[1], generator=g)
            q = torch.randn(self.B, 4, generator=g).double().numpy(); q /= np.linalg.norm(q, axis=1, keepdims=True)
            t = torch.rand(self.B, 3, generator=g).double().numpy() * np.array([1.2, 0.8, 8.0]) + np.array([-0.6, -0.4, 4.0])
            px = pose.project_keypoints(q, t, K, dist, pts)
            bbox = np.stack([px[:, 0].min(1), px[:, 0].max(1), px[:, 1].min(1), px[:, 1].max(1)], axis=1)
            yield x, torch.from_numpy(bbox).float(), torch.from_numpy(q).float(), torch.from_numpy(t).float()
