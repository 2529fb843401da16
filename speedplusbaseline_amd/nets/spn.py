"""Spacecraft Pose Network surface (reference src/nets/spn.py:37-143).

The SPN (AlexNet trunk + two attitude heads) HIP path is the next row of the hot-path table and is not built yet;
softmax_cross_entropy_with_logits and the class are kept importable so callers fail loudly and precisely instead of
at import time.  No CPU/PyTorch fallback is provided on purpose.
"""
import torch.nn as nn


def softmax_cross_entropy_with_logits(logits, target, reduction="mean"):
    raise NotImplementedError("SPN soft-target cross-entropy has no HIP kernel yet (DESIGN.md: scope, next rows)")


class SpacecraftPoseNet(nn.Module):
    def __init__(self, num_classes, keep_prob=0.5, pretrain=True):
        super().__init__()
        raise NotImplementedError("SpacecraftPoseNet (model_name='spn') is not built yet in this MI355X implementation; "
                                  "see DESIGN.md (scope / next rows). KRN and KRN+DANN are available.")
