import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import preproc_oracle as P
from speedplusbaseline_amd.transforms import build_transforms
GOLD = np.load("tests/golden/preproc_golden.npz")
dev = torch.device("cuda", 0)
for i in range(int(GOLD["n"])):
    model, is_train, seed, fh, fw, fseed = [int(v) for v in GOLD["case%d_meta" % i]]
    model = ("krn", "spn")[model]; p = float(GOLD["case%d_p" % i])
    frame = P.synth_frame(fh, fw, fseed)
    t = build_transforms(model, (32, 32), p_aug=p, is_train=bool(is_train), device=dev, device_noise=False)
    torch.manual_seed(seed)
    img, box, k = t([frame], [GOLD["case%d_bbox" % i]], [GOLD["case%d_kp" % i].copy()])
    g = torch.from_numpy(GOLD["case%d_image" % i])
    d = (img[0].cpu() - g).abs()
    rec = None
    if model == "krn":
        torch.manual_seed(seed)
        _, _, _, rec = P.krn_sample(frame, GOLD["case%d_bbox" % i], GOLD["case%d_kp" % i].copy(), 32, p, bool(is_train))
        rec = {k_: (v if k_ != "noise" else v is not None) for k_, v in rec.items()}
    print(i, model, "mismatch %d / %d, max %.3e" % (int((d > 0).sum()), d.numel(), float(d.max())), rec)
