"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference's per-sample input pipeline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(speedplusbaseline_amd.transforms + csrc/preproc.hip) never does.

Follows /root/reference/src/datasets/transforms.py:
  RandomCrop            :107-160   (bbox -> square RoI, random enlargement / shift in training, keypoint normalisation)
  ResizeCrop            :163-186
  ToTensor              :192-196
  Rotate / Flip         :38-68     (image + keypoints)
  BrightnessContrast    :70-92
  GaussianNoise         :94-105
  RandomApply / Compose :198-215, build_transforms :217-244
and the call site Park2019KRNDataset.__getitem__ (src/datasets/Park2019KRNDataset.py:81-109).

Third-party arithmetic not under /root/reference: torchvision==0.9.0 (requirements.txt:4) `transforms.functional`, absent from
this image.  Its published behaviour for the calls above, restated here:
  resized_crop(PIL, top, left, h, w, size)  = img.crop((left, top, left+w, top+h)).resize(size[::-1], Image.BILINEAR)
  to_tensor(PIL 'RGB')                      = uint8 HWC -> CHW float32 / 255
  rotate(tensor, 90k), hflip, vflip         = exact quarter turns counter-clockwise / mirror images (index permutations; the
                                              keypoint updates at transforms.py:46-52,62-66 assume exactly that)
Pillow (12.2 here and on the GPU box) IS present, so `resize_pil` calls it directly and `resize_restated` -- the integer
arithmetic of Pillow's Resample.c that the HIP kernels implement -- is pinned against it (tests/test_preproc_oracle.py).
Pinned as a whole by tests/golden/preproc_golden.npz: outputs of the reference's own build_transforms run in this container
(tests/golden/make_golden_preproc.py, with the torchvision stand-in described above).
"""
import math

import numpy as np
import torch

PRECISION_BITS = 22   # 32 - 8 - 2 (Pillow Resample.c)


# ---------------------------------------------------------------------------------------------------------------- resize
def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter over the whole input"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int64)
    return bounds, ik


def resize_restated(img, out_w, out_h):
    """Image.resize((out_w, out_h), BILINEAR) of a uint8 [H, W, C] array: horizontal pass into uint8, then vertical pass"""
    H, W, C = img.shape
    bh, kh = resample_coeffs(W, out_w)
    bv, kv = resample_coeffs(H, out_h)
    tmp = np.zeros((H, out_w, C), dtype=np.uint8)
    for xx in range(out_w):
        xmin, n = bh[xx]
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += img[:, xmin + x, :].astype(np.int64) * kh[xx, x]
        tmp[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    out = np.zeros((out_h, out_w, C), dtype=np.uint8)
    for yy in range(out_h):
        ymin, n = bv[yy]
        acc = np.full((out_w, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for y in range(n):
            acc += tmp[ymin + y].astype(np.int64) * kv[yy, y]
        out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_pil(img, out_w, out_h):
    from PIL import Image
    return np.asarray(Image.fromarray(img, "RGB").resize((out_w, out_h), Image.BILINEAR))


# ------------------------------------------------------------------------------------------------------ the transforms
def random_crop_box(bbox, org_w, org_h, is_train, draw):
    """transforms.py:117-146.  draw() returns the next torch.rand(1) tensor.  -> (xmin, xmax, ymin, ymax) ints"""
    xmin, xmax, ymin, ymax = bbox
    w, h = xmax - xmin, ymax - ymin
    x, y = xmin + w / 2.0, ymin + h / 2.0
    roi_size = max((w, h))
    if is_train:
        roi_size = (1 + 0.5 * draw()) * roi_size
        fx = 0.2 * (draw() * 2 - 1) * roi_size
        fy = 0.2 * (draw() * 2 - 1) * roi_size
    else:
        roi_size = (1 + 0.2) * roi_size
        fx = fy = 0
    xmin = max(0, int(x - roi_size / 2.0 + fx))
    xmax = min(org_w, int(x + roi_size / 2.0 + fx))
    ymin = max(0, int(y - roi_size / 2.0 + fy))
    ymax = min(org_h, int(y + roi_size / 2.0 + fy))
    return xmin, xmax, ymin, ymax


def resize_crop_box(bbox, org_w, org_h):
    """transforms.py:169-178"""
    xmin, xmax, ymin, ymax = bbox
    return max(0, int(xmin)), min(org_w, int(xmax)), max(0, int(ymin)), min(org_h, int(ymax))


def normalise_keypoints(keypts, box):
    """transforms.py:151-154"""
    xmin, xmax, ymin, ymax = box
    k = torch.tensor(keypts, dtype=torch.float32)
    k[0] = (k[0] - xmin) / (xmax - xmin)
    k[1] = (k[1] - ymin) / (ymax - ymin)
    return k


def to_tensor(u8_hwc):
    return torch.from_numpy(np.ascontiguousarray(u8_hwc)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def rotate(image, keypts, k):
    """transforms.py:38-54 for angle = 90 k (counter-clockwise)"""
    image = torch.rot90(image, k, dims=(1, 2))
    x, y = keypts[0].clone(), keypts[1].clone()
    if k == 1:
        keypts[0], keypts[1] = y, 1.0 - x
    elif k == 2:
        keypts[0], keypts[1] = 1.0 - x, 1.0 - y
    elif k == 3:
        keypts[0], keypts[1] = 1.0 - y, x
    return image, keypts


def flip(image, keypts, horizontal):
    """transforms.py:56-68"""
    if horizontal:
        image = torch.flip(image, dims=(2,))
        keypts[0] = 1.0 - keypts[0]
    else:
        image = torch.flip(image, dims=(1,))
        keypts[1] = 1.0 - keypts[1]
    return image, keypts


def brightness_contrast(image, a, b):
    """transforms.py:70-92: clamp(a * image + b, 0, 1) with float32 one-element tensors a, b"""
    return torch.clamp(a * image + b, 0, 1)


def gaussian_noise(image, noise, std=25 / 255):
    """transforms.py:94-105: noise = randn(image.shape) * std"""
    return torch.clamp(image + noise * std, 0, 1)


def krn_sample(frame_u8, bbox, keypts, size, p_aug, is_train, resize=resize_pil):
    """build_transforms('krn', (size, size), p_aug, is_train) (transforms.py:217-244) on one uint8 [H, W, 3] frame, drawing
    from torch's global generator in the reference's order.  Returns (image [3,S,S] f32, bbox [4] f32, keypts [2,K] f32) and
    the draws as a dict (what the GPU path is handed in the parity tests)."""
    org_h, org_w = frame_u8.shape[:2]
    rec = {}
    box = random_crop_box(bbox, org_w, org_h, is_train, lambda: torch.rand(1))
    rec["box"] = box
    k = normalise_keypoints(keypts, box)
    xmin, xmax, ymin, ymax = box
    crop = frame_u8[ymin:ymax, xmin:xmax]
    image = to_tensor(resize(np.ascontiguousarray(crop), size, size))
    rec.update(rot=0, flip=0, bc=None, noise=None)
    if is_train:
        alpha = torch.tensor((0.5, 2.0)).log()
        beta = torch.tensor((-25, 25)) / 255
        if torch.rand(1) < p_aug:                                  # Rotate
            kq = int(float(torch.randint(1, 4, (1,))))
            image, k = rotate(image, k, kq)
            rec["rot"] = kq
        if torch.rand(1) < p_aug:                                  # Flip
            hz = bool(torch.rand(1) < 0.5)
            image, k = flip(image, k, hz)
            rec["flip"] = 1 if hz else 2
        if torch.rand(1) < p_aug:                                  # BrightnessContrast
            loga = torch.rand(1) * (alpha[1] - alpha[0]) + alpha[0]
            a = loga.exp()
            b = torch.rand(1) * (beta[1] - beta[0]) + beta[0]
            image = brightness_contrast(image, a, b)
            rec["bc"] = (float(a), float(b))
        if torch.rand(1) < p_aug:                                  # GaussianNoise
            nz = torch.randn(image.shape, dtype=torch.float32)
            image = gaussian_noise(image, nz)
            rec["noise"] = nz
    return image, torch.tensor(box, dtype=torch.float32), k, rec


def synth_frame(h, w, seed):
    """bit-portable synthetic grey frame replicated to 3 bands (smooth blobs + texture), uint8 [h, w, 3]"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w))
    for _ in range(6):
        cy, cx, r, a = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(20, 120), rng.uniform(60, 200)
        img += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))
    img += rng.integers(0, 40, (h, w))
    g = np.clip(img, 0, 255).astype(np.uint8)
    return np.repeat(g[:, :, None], 3, axis=2)
