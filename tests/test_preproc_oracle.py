"""CPU: the input-pipeline oracle (oracle/preproc_oracle.py) against (1) Pillow itself -- the integer restatement of
Image.resize(BILINEAR) must be bit-exact -- and (2) golden vectors produced by the reference's own build_transforms
(tests/golden/make_golden_preproc.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import preproc_oracle as P

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "preproc_golden.npz"), allow_pickle=False)


@pytest.mark.parametrize("h,w,s", [(97, 131, 32), (300, 260, 64), (40, 50, 64), (224, 224, 224), (225, 223, 224), (900, 700, 224)])
def test_restated_resample_is_pillow_bit_exact(h, w, s):
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_restated(img, s, s), P.resize_pil(img, s, s))


def _case(i):
    model, is_train, seed, fh, fw, fseed = [int(v) for v in GOLD["case%d_meta" % i]]
    return ("krn", "spn")[model], bool(is_train), seed, P.synth_frame(fh, fw, fseed), float(GOLD["case%d_p" % i])


@pytest.mark.parametrize("i", range(int(GOLD["n"])))
def test_pipeline_matches_reference_golden(i):
    model, is_train, seed, frame, p = _case(i)
    S = int(GOLD["S"])
    bbox, kp = GOLD["case%d_bbox" % i], GOLD["case%d_kp" % i]
    torch.manual_seed(seed)
    if model == "krn":
        img, box, k, rec = P.krn_sample(frame, bbox, kp.copy(), S, p, is_train, resize=P.resize_restated)
    else:
        b = P.resize_crop_box(bbox, frame.shape[1], frame.shape[0])
        img = P.to_tensor(P.resize_restated(np.ascontiguousarray(frame[b[2]:b[3], b[0]:b[1]]), S, S))
        box, k = torch.tensor(bbox), torch.tensor(kp)
    assert torch.equal(img, torch.from_numpy(GOLD["case%d_image" % i]))          # bit-exact, float32
    assert np.array_equal(box.numpy(), GOLD["case%d_obox" % i])
    assert np.array_equal(k.numpy(), GOLD["case%d_okp" % i])
