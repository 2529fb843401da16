"""Batched input pipeline on the MI355X -- the GPU counterpart of the reference's per-sample src/datasets/transforms.py
(build_transforms, transforms.py:217-244; called from Park2019KRNDataset.__getitem__, Park2019KRNDataset.py:81-109).

The reference resizes, converts and augments ONE image per DataLoader-worker call with PIL / torchvision on the CPU.  At the
step rates of this build (>10^4 images/s per GPU) that is the wall, so here the host only decides (crop box, coins, angles,
a, b -- a few scalars per image, drawn from torch's generator in the reference's order, so a seeded run makes the same
decisions) and cuts the region of interest out of the decoded frame (a slice); the resize to the network input, ToTensor and
the four augmentations of the whole batch are three HIP launches (csrc/preproc.hip), bit-exact against Pillow's bilinear
resample and the reference's float32 tensor arithmetic.

    t = build_transforms('krn', (224, 224), p_aug=0.5, is_train=True, device='cuda')
    images, bboxes, keypts = t(frames, bboxes, keypts)        # lists of B frames (PIL.Image or uint8 [H,W,3] / [H,W]) ...
    # images: float32 [B,3,224,224] on the GPU in [0,1]; bboxes [B,4]; keypts [B,2,K] (CPU, float32)

No CPU fallback: without the HIP library or a GPU this raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

NOISE_STD = 25 / 255          # GaussianNoise(std=25), transforms.py:100,237
_ALPHA = (0.5, 2.0)           # BrightnessContrast(alpha=(0.5, 2.0), beta=(-25, 25)), transforms.py:236
_BETA = (-25, 25)


def _as_u8(frame):
    """PIL.Image (any mode; converted like Image.open(...).convert('RGB'), Park2019KRNDataset.py:84) or ndarray -> uint8 [H,W,C]"""
    if isinstance(frame, np.ndarray):
        a = frame
    else:
        a = np.asarray(frame if frame.mode in ("L", "RGB") else frame.convert("RGB"))
    if a.dtype != np.uint8:
        raise TypeError("frames must be uint8 images")
    if a.ndim == 2:
        a = a[:, :, None]
    if a.shape[2] not in (1, 3):
        raise ValueError("frames must have 1 or 3 channels")
    return a


class GpuBatchTransform:
    """build_transforms(model_name, input_size, p_aug, is_train) of the reference for a whole batch on the GPU.

    device_noise=False draws the Gaussian noise on the CPU with torch.randn(image.shape) exactly where the reference does (a
    seeded run is then bit-identical to the reference pipeline); True draws it on the GPU (same distribution, no 600 KB
    host-to-device copy per noisy image) -- the production setting."""

    def __init__(self, model_name, input_size, p_aug=0.5, is_train=True, device="cuda", device_noise=True):
        if model_name not in ("krn", "spn"):
            raise ValueError(model_name)
        if input_size[0] != input_size[1]:
            raise ValueError("square network inputs only (the quarter-turn augmentation needs them)")
        self.model_name, self.S, self.p, self.is_train = model_name, int(input_size[0]), p_aug, is_train
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GpuBatchTransform runs on the MI355X only (no CPU path)")
        self.device_noise = device_noise
        self._alpha = torch.tensor(_ALPHA).log()
        self._beta = torch.tensor(_BETA) / 255
        self._ws = {}
        self._pin = None
        self._taps = int(L.lib().spb_preproc_max_taps())

    # ---- host decisions, one sample (transforms.py:107-160 / :163-186 and :198-208 in call order)
    def _crop_box(self, bbox, org_w, org_h):
        xmin, xmax, ymin, ymax = bbox
        if self.model_name == "spn":                                   # ResizeCrop
            return max(0, int(xmin)), min(org_w, int(xmax)), max(0, int(ymin)), min(org_h, int(ymax))
        w, h = xmax - xmin, ymax - ymin                                # RandomCrop
        x, y = xmin + w / 2.0, ymin + h / 2.0
        roi_size = max((w, h))
        if self.is_train:
            roi_size = (1 + 0.5 * torch.rand(1)) * roi_size
            fx = 0.2 * (torch.rand(1) * 2 - 1) * roi_size
            fy = 0.2 * (torch.rand(1) * 2 - 1) * roi_size
        else:
            roi_size = (1 + 0.2) * roi_size
            fx = fy = 0
        return (max(0, int(x - roi_size / 2.0 + fx)), min(org_w, int(x + roi_size / 2.0 + fx)),
                max(0, int(y - roi_size / 2.0 + fy)), min(org_h, int(y + roi_size / 2.0 + fy)))

    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._ws[key] = t
        return t[:n].view(*shape) if n else t[:0]

    def __call__(self, frames, bboxes, keypts=None):
        a, out, boxes, kps = self.stage(frames, bboxes, keypts)
        self.launch(a)
        return out, boxes, kps

    def launch(self, a):
        """enqueue the three kernels for a staged batch (its packed crops and tables are resident on the GPU)"""
        with torch.cuda.device(self.device):
            L.check(L.lib().spb_preproc_batch(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "spb_preproc_batch")

    def stage(self, frames, bboxes, keypts=None):
        """host half: decisions, crops, one packed upload.  Returns (kernel arguments, output tensor, bboxes, keypts)"""
        S, B = self.S, len(frames)
        crops, table, ftable, out_boxes, out_k, noises = [], np.zeros((B, 8), dtype=np.int32), np.zeros((B, 2), dtype=np.float32), [], [], {}
        off = 0
        chans = None
        for i, frame in enumerate(frames):
            a = _as_u8(frame)
            org_h, org_w = a.shape[:2]
            chans = a.shape[2] if chans is None else chans
            if a.shape[2] != chans:
                raise ValueError("all frames of a batch must have the same number of channels")
            bbox = np.asarray(bboxes[i], dtype=np.float32)
            xmin, xmax, ymin, ymax = self._crop_box(bbox, org_w, org_h)
            if xmax <= xmin or ymax <= ymin:
                raise ValueError("empty region of interest for sample %d: %s" % (i, (xmin, xmax, ymin, ymax)))
            h, w = ymax - ymin, xmax - xmin
            if int(np.ceil(max(h, w, S) / S)) * 2 + 1 > self._taps:
                raise ValueError("region of interest %dx%d is more than %dx the network input: outside the resize kernels' filter "
                                 "length" % (w, h, (self._taps - 1) // 2))
            if self.model_name == "krn":
                out_boxes.append(torch.tensor([xmin, xmax, ymin, ymax], dtype=torch.float32))
                k = torch.tensor(np.asarray(keypts[i]), dtype=torch.float32) if keypts is not None else torch.zeros(2, 1)
                k[0] = (k[0] - xmin) / (xmax - xmin)
                k[1] = (k[1] - ymin) / (ymax - ymin)
            else:
                out_boxes.append(torch.tensor(bbox, dtype=torch.float32))                      # SPN keeps the original box
                k = torch.tensor(np.asarray(keypts[i]), dtype=torch.float32) if keypts is not None else torch.zeros(2, 1)
            crops.append(a[ymin:ymax, xmin:xmax])                 # a view: copied once, straight into the pinned staging buffer
            rot = flip = flags = 0
            if self.is_train and self.model_name == "krn":
                if torch.rand(1) < self.p:                                                      # Rotate
                    rot = int(float(torch.randint(1, 4, (1,))))
                    x_, y_ = k[0].clone(), k[1].clone()
                    if rot == 1:
                        k[0], k[1] = y_, 1.0 - x_
                    elif rot == 2:
                        k[0], k[1] = 1.0 - x_, 1.0 - y_
                    else:
                        k[0], k[1] = 1.0 - y_, x_
                if torch.rand(1) < self.p:                                                      # Flip
                    if torch.rand(1) < 0.5:
                        flip = 1; k[0] = 1.0 - k[0]
                    else:
                        flip = 2; k[1] = 1.0 - k[1]
                if torch.rand(1) < self.p:                                                      # BrightnessContrast
                    loga = torch.rand(1) * (self._alpha[1] - self._alpha[0]) + self._alpha[0]
                    ftable[i, 0] = float(loga.exp())
                    ftable[i, 1] = float(torch.rand(1) * (self._beta[1] - self._beta[0]) + self._beta[0])
                    flags |= 1
                if torch.rand(1) < self.p:                                                      # GaussianNoise
                    flags |= 2
                    if not self.device_noise:
                        noises[i] = torch.randn((3, S, S), dtype=torch.float32)
            table[i, 0:2] = np.array([off], dtype=np.int64).view(np.int32)          # byte offset, low / high word
            table[i, 2:8] = (h, w, rot, flip, flags, 0)
            off += h * w * chans
            off = (off + 15) // 16 * 16
            out_k.append(k)
        # ---- one packed upload, three launches
        # two pinned staging slots: the upload of batch i may still be in flight when batch i+1 is packed on the host
        # (AugLookahead stages one batch ahead with no synchronisation), so a slot is rewritten only after the event recorded
        # behind its last host-to-device copy has completed
        slot = self._pin_slot = (getattr(self, "_pin_slot", 1) + 1) % 2
        if self._pin is None:
            self._pin, self._pin_ev = [None, None], [None, None]
        if self._pin_ev[slot] is not None:
            self._pin_ev[slot].synchronize()
        pin = self._pin[slot]
        if pin is None or pin.numel() < off:
            pin = self._pin[slot] = torch.empty(max(off, 1), dtype=torch.uint8).pin_memory()
        packed = pin.numpy()
        pos = 0
        for c in crops:
            packed[pos:pos + c.size].reshape(c.shape)[...] = c
            pos = (pos + c.size + 15) // 16 * 16
        dev = self.device
        src = self._buf("src%d" % slot, (off,), torch.uint8); src.copy_(pin[:off], non_blocking=True)
        if self._pin_ev[slot] is None:
            self._pin_ev[slot] = torch.cuda.Event()
        self._pin_ev[slot].record(torch.cuda.current_stream(dev))
        tab = self._buf("tab", (B, 8), torch.int32); tab.copy_(torch.from_numpy(table), non_blocking=True)
        ftab = self._buf("ftab", (B, 2), torch.float32); ftab.copy_(torch.from_numpy(ftable), non_blocking=True)
        flags_any = int(np.bitwise_or.reduce(table[:, 6])) if B else 0
        noise = None
        if flags_any & 2:
            noise = self._buf("noise", (B, 3, S, S), torch.float32)
            if self.device_noise:
                noise.normal_()
            else:
                for i, nz in noises.items():
                    noise[i].copy_(nz, non_blocking=True)
        max_h = int(table[:, 2].max())
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
        a = L.PreprocArgs()
        a.src, a.table, a.ftable, a.noise, a.out = src.data_ptr(), tab.data_ptr(), ftab.data_ptr(), (noise.data_ptr() if noise is not None else None), out.data_ptr()
        a.bounds = self._buf("bounds", (B, 2, S, 2), torch.int32).data_ptr()
        a.coeffs = self._buf("coeffs", (B, 2, S, self._taps), torch.int32).data_ptr()
        a.tmp = self._buf("tmp", (B, max_h, S, chans), torch.uint8).data_ptr()
        a.B, a.S, a.C, a.max_h, a.flags_any, a.noise_std = B, S, chans, max_h, flags_any, NOISE_STD
        a.src_bytes = int(sum(c.size for c in crops))        # python-side attribute (not part of the C struct): for the bench
        return a, out, torch.stack(out_boxes), torch.stack(out_k)


def build_transforms(model_name, input_size, p_aug=0.5, is_train=True, device="cuda", device_noise=True):
    """same arguments as the reference's build_transforms (transforms.py:217) plus the device; returns the batched transform"""
    return GpuBatchTransform(model_name, input_size, p_aug, is_train, device, device_noise)
