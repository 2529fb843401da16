"""Generates tests/golden/preproc_golden.npz by running the REFERENCE's own input pipeline
(/root/reference/src/datasets/transforms.py: build_transforms('krn' | 'spn', ...)) in this container.

torchvision==0.9.0 (requirements.txt:4) is absent from the image; the reference uses five of its functional ops, for which this
script installs a stand-in module BEFORE importing the reference (nothing of the reference is copied):
    resized_crop(PIL, top, left, h, w, size) -> img.crop(...).resize(size[::-1], Image.BILINEAR)      (what 0.9.0 does for PIL images)
    to_tensor(PIL)                            -> uint8 HWC -> CHW float32 / 255
    rotate(tensor, 90k) / hflip / vflip       -> torch.rot90(k) counter-clockwise / torch.flip
The arithmetic that matters (Pillow's resample, torch's float32 ops) is therefore the real third-party code that is present;
crop boxes, keypoint updates, coin order, brightness/contrast and noise are the reference's own lines.

Inputs are seeded synthetic frames (oracle.preproc_oracle.synth_frame); stored: the random state recipe (seed), the inputs'
bounding boxes / keypoints, and the reference's outputs (image tensor, bbox, keypoints) at a small network input (32x32).
Run:  python tests/golden/make_golden_preproc.py
"""
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import preproc_oracle as P  # noqa: E402

tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); F = types.ModuleType("torchvision.transforms.functional")


def resized_crop(img, top, left, height, width, size, interpolation=Image.BILINEAR):
    return img.crop((left, top, left + width, top + height)).resize(tuple(size[::-1]), interpolation)


def to_tensor(pic):
    return torch.from_numpy(np.array(pic, np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def rotate(img, angle):
    assert float(angle) in (90.0, 180.0, 270.0)
    return torch.rot90(img, int(round(float(angle) / 90)), dims=(1, 2))


F.resized_crop, F.to_tensor, F.rotate = resized_crop, to_tensor, rotate
F.hflip = lambda img: torch.flip(img, dims=(2,))
F.vflip = lambda img: torch.flip(img, dims=(1,))
tv.transforms = tvt; tvt.functional = F
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": F})

import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_transforms", "/root/reference/src/datasets/transforms.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

S = 32
CASES = []   # (model, is_train, p_aug, seed, frame (h, w, seed), bbox, n samples)
for seed in range(8):
    CASES.append(("krn", True, 0.5, 100 + seed, (240, 320, seed), (60.0 + 5 * seed, 200.0 + 7 * seed, 40.0 + 3 * seed, 170.0 + 4 * seed)))
CASES.append(("krn", True, 1.0, 7, (240, 320, 3), (10.0, 150.0, 100.0, 235.0)))         # every augmentation, box at the frame edge
CASES.append(("krn", False, 0.5, 8, (240, 320, 4), (100.0, 220.0, 60.0, 120.0)))        # evaluation: fixed 1.2x enlargement
CASES.append(("spn", True, 0.5, 9, (240, 320, 5), (33.0, 300.0, 20.0, 200.0)))          # SPN: ResizeCrop, no augmentation
CASES.append(("krn", True, 0.5, 10, (120, 90, 6), (30.0, 50.0, 40.0, 60.0)))            # small box: the resize is an up-scale

out = {"S": np.int32(S), "n": np.int32(len(CASES))}
for i, (model, is_train, p_aug, seed, (fh, fw, fseed), bbox) in enumerate(CASES):
    frame = P.synth_frame(fh, fw, fseed)
    kp = np.random.default_rng(1000 + i).uniform([[bbox[0]], [bbox[2]]], [[bbox[1]], [bbox[3]]], (2, 11)).astype(np.float32)
    t = ref.build_transforms(model, (S, S), p_aug=p_aug, is_train=is_train)
    torch.manual_seed(seed)
    img, bb, kk = t(Image.fromarray(frame, "RGB"), np.array(bbox, dtype=np.float32), kp.copy())
    out["case%d_meta" % i] = np.array([{"krn": 0, "spn": 1}[model], int(is_train), seed, fh, fw, fseed], dtype=np.int64)
    out["case%d_p" % i] = np.float64(p_aug)
    out["case%d_bbox" % i] = np.array(bbox, dtype=np.float32)
    out["case%d_kp" % i] = kp
    out["case%d_image" % i] = img.numpy()
    out["case%d_obox" % i] = np.asarray(bb, dtype=np.float32)
    out["case%d_okp" % i] = np.asarray(kk, dtype=np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preproc_golden.npz"), **out)
print("wrote", len(CASES), "cases")
