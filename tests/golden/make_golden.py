"""Generates tests/golden/*.npz by IMPORTING the reference (read-only at /root/reference) in this container.

Only data (inputs by seed recipe, outputs as arrays) is committed; no reference source travels.  The reference needs
`torchvision` (absent here) for its backbone and `cv2`/`tensorboard`/Tk for unrelated utilities, so those modules are
stubbed in sys.modules before the import.  The torchvision stub below is this repo's own nn.Module statement of
mobilenet_v2 (architecture from the published MobileNetV2 spec, torchvision 0.9 key names); outputs that flow through
it are flagged "backbone=stub" -- the reference pins nothing at that boundary (see oracle/krn_oracle.py header).

Run:  python tests/golden/make_golden.py      (writes next to this file)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import krn_oracle as O  # noqa: E402


# ---------------------------------------------------------------------------------------- stubs for absent packages
class _ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, k=3, stride=1, groups=1):
        super().__init__(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout),
                         nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = cin * t
        self.use_res = stride == 1 and cin == cout
        layers = []
        if t != 1:
            layers.append(_ConvBNReLU(cin, hid, 1))
        layers += [_ConvBNReLU(hid, hid, 3, stride, groups=hid), nn.Conv2d(hid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class _MobileNetV2(nn.Module):
    def __init__(self):
        super().__init__()
        feats = [_ConvBNReLU(3, 32, 3, 2)]
        for k, t, cin, cout, s in O.block_specs():
            feats.append(_InvertedResidual(cin, cout, s, t))
        feats.append(_ConvBNReLU(320, 1280, 1))  # features[18], dropped by the reference's [:-1]
        self.features = nn.Sequential(*feats)


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.mobilenet_v2 = lambda pretrained=False, **kw: _MobileNetV2()
    tvt = types.ModuleType("torchvision.transforms")
    tvtf = types.ModuleType("torchvision.transforms.functional")
    tv.models, tv.transforms = tvm, tvt
    tvt.functional = tvtf
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvtf, "cv2": types.ModuleType("cv2")})
    vis = types.ModuleType("src.utils.visualize")
    vis.imshow = vis.plot_2D_bbox = vis.scatter_keypoints = lambda *a, **k: None
    sys.modules["src.utils.visualize"] = vis


G7S_TENSORS = ("base.0.1.weight", "base.0.1.bias", "base.9.conv.1.1.weight", "extras.2.conv.1.bias",
               "extras.3.conv.4.weight", "head.0.bias", "base.17.conv.3.weight")


def _cfg(**kw):
    c = types.SimpleNamespace(model_name="krn", num_keypoints=11, num_classes=5000, dann=False, optimizer="adamw", lr=1e-4,
                              momentum=0.9, weight_decay=0.01, max_epochs=75, texture_ratio=0.5, use_cuda=False)
    c.__dict__.update(kw)
    return c


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    sys.argv = [sys.argv[0]]
    from src.nets.build import get_model, get_optimizer
    from src.nets.park2019 import ConvDw, RouterV2
    from src.nets.revgrad import GradientReversalFunction
    from src.core.trainer import train_single_epoch_krn
    from src.core.dann import train_dann_single_epoch_krn
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    dev = torch.device("cpu")

    # ---- G1 ConvDw / G2 RouterV2 / G5 GRL (reference code only, no stub involved)
    cd = ConvDw(320, 1024, 1)
    sd = O.init_state(11)
    m = {"conv.0.weight": "extras.0.conv.0.weight", "conv.1": "extras.0.conv.1", "conv.3.weight": "extras.0.conv.3.weight",
         "conv.4": "extras.0.conv.4"}
    st = {}
    for k, v in m.items():
        if k.endswith("weight") and not k.startswith("conv.1") and not k.startswith("conv.4"):
            st[k] = sd[v]
        else:
            for suf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                st[k + "." + suf] = sd[v + "." + suf]
    cd.load_state_dict(st, strict=True)
    xin = torch.from_numpy(O.prng.uniform("g1/x", (2, 320, 7, 7), -1, 1)).requires_grad_(True)
    cd.train()
    y = cd(xin)
    y.square().sum().backward()
    out["g1_convdw_train_y"] = y.detach().numpy()
    out["g1_convdw_train_dx"] = xin.grad.numpy()
    out["g1_convdw_dw_grad"] = cd.conv[0].weight.grad.numpy()
    cd.eval()
    out["g1_convdw_eval_y"] = cd(xin.detach()).detach().numpy()

    rt = RouterV2(96, 64)
    st = {"conv.0.weight": sd["extras.2.conv.0.weight"]}
    for suf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
        st["conv.1." + suf] = sd["extras.2.conv.1." + suf]
    rt.load_state_dict(st, strict=True)
    rt.eval()
    x1 = torch.from_numpy(O.prng.uniform("g2/x1", (2, 1024, 7, 7), -1, 1))
    x2 = torch.from_numpy(O.prng.uniform("g2/x2", (2, 96, 14, 14), -1, 1))
    out["g2_router_eval"] = rt(x1, x2).detach().numpy()

    gx = torch.from_numpy(O.prng.uniform("g5/x", (3, 5), -1, 1)).requires_grad_(True)
    gy = GradientReversalFunction.apply(gx, 0.37)
    (gy * torch.arange(15.0).view(3, 5)).sum().backward()
    out["g5_grl_y"] = gy.detach().numpy(); out["g5_grl_dx"] = gx.grad.numpy()

    # ---- G4 full KRN (backbone = stub): train-mode loss, eval-mode keypoints, gradient norm
    B = 4
    x, yk = O.synth_batch(B)
    out["g4_synth_sum"] = np.array([x.double().sum().item()])
    model = get_model(_cfg())
    ref_keys = list(model.state_dict().keys())
    out["g4_state_keys"] = np.array(ref_keys)
    out["g4_state_shapes"] = np.array([str(tuple(v.shape)) for v in model.state_dict().values()])
    model.load_state_dict(O.init_state(11), strict=True)
    model.eval()
    with torch.no_grad():
        xc, yc = model(x)
    out["g4_eval_xc"] = xc.numpy(); out["g4_eval_yc"] = yc.numpy()
    model.train()
    loss, sm = model(x, yk)
    loss.backward()
    out["g4_train_loss"] = np.array([float(loss), sm["loss_x"], sm["loss_y"]])
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e9)
    out["g4_grad_norm"] = np.array([float(gn)])
    for k in ("base.0.0.weight", "base.7.conv.1.0.weight", "base.17.conv.2.weight", "extras.2.conv.0.weight", "head.0.weight",
              "head.0.bias", "extras.3.conv.4.weight", "base.3.conv.3.bias"):
        g = dict(model.named_parameters())[k].grad.double()
        out["g4_grad/" + k] = np.array([float(g.sum()), float((g * g).sum())])
    out["g4_running_mean_base01"] = model.state_dict()["base.0.1.running_mean"].numpy()
    out["g4_running_var_extras34"] = model.state_dict()["extras.3.conv.4.running_var"].numpy()

    # ---- G7 train_single_epoch_krn, 2 iterations, AdamW lr 1e-4 (pins forward->zero_grad->backward->clip->step)
    model = get_model(_cfg())
    model.load_state_dict(O.init_state(11), strict=True)
    cfg = _cfg()
    opt = get_optimizer(cfg, model)
    batches = [O.synth_batch(B, tag="it%d" % i) for i in range(2)]
    losses = []
    orig_fwd = model.forward

    def spy(xx, yy=None):
        r = orig_fwd(xx, yy)
        if yy is not None:
            losses.append([float(r[0]), r[1]["loss_x"], r[1]["loss_y"]])
        return r

    model.forward = spy
    train_single_epoch_krn(1, cfg, model, batches, opt, None, dev)
    print()
    out["g7_losses"] = np.array(losses)
    cs = O.checksum(model.state_dict())
    out["g7_keys"] = np.array(list(cs.keys())); out["g7_checksums"] = np.stack(list(cs.values()))

    # ---- G7s same loop with SGD(momentum): the update is proportional to the gradient, so the post-step state is a
    # well-conditioned function of it (step 1 of Adam is lr*sign(g), which turns fp32 noise on near-zero gradients
    # into +-lr flips and is only reproducible between two runs of the very same kernels)
    model = get_model(_cfg())
    model.load_state_dict(O.init_state(11), strict=True)
    cfg_s = _cfg(optimizer="sgd", lr=0.05, momentum=0.9, weight_decay=5e-5)
    opt = get_optimizer(cfg_s, model)
    losses = []
    orig_fwd = model.forward
    model.forward = spy
    train_single_epoch_krn(1, cfg_s, model, batches, opt, None, dev)
    print()
    out["g7s_losses"] = np.array(losses)
    cs = O.checksum(model.state_dict())
    out["g7s_keys"] = np.array(list(cs.keys())); out["g7s_checksums"] = np.stack(list(cs.values()))
    for k in G7S_TENSORS:
        out["g7s_final/" + k] = model.state_dict()[k].numpy().copy()

    # ---- G6 RevGrad forward + 2 DANN iterations (dann.py)
    cfgd = _cfg(dann=True, max_epochs=5)
    rg = get_model(cfgd)
    out["g6_state_keys"] = np.array(list(rg.state_dict().keys()))
    rg.load_state_dict(O.init_state(11, dann=True), strict=True)
    rg.train()
    xs, ys = O.synth_batch(B, tag="src0"); xt, _ = O.synth_batch(B, tag="tgt0")
    (lp, sm), dom = rg(xs, y=ys, alpha=0.3)
    out["g6_fwd"] = np.array([float(lp), sm["loss_x"], sm["loss_y"]]); out["g6_dom"] = dom.detach().numpy()
    rg = get_model(cfgd)
    rg.load_state_dict(O.init_state(11, dann=True), strict=True)
    optd = get_optimizer(cfgd, rg)
    src = [O.synth_batch(B, tag="src%d" % i) for i in range(2)]
    tgt = [O.synth_batch(B, tag="tgt%d" % i)[0] for i in range(2)]
    import torch.nn.functional as F
    rec = []
    orig_bce = F.binary_cross_entropy_with_logits

    def spy_bce(inp, tg, **kw):
        r = orig_bce(inp, tg, **kw)
        rec.append(float(r))
        return r

    nn.functional.binary_cross_entropy_with_logits = spy_bce
    train_dann_single_epoch_krn(1, cfgd, rg, src, tgt, optd, None, dev)
    print()
    nn.functional.binary_cross_entropy_with_logits = orig_bce
    out["g6_dann_bce"] = np.array(rec)  # [src0, tgt0, src1, tgt1]
    out["g6_alphas"] = np.array([O.dann_alpha(i, 1, 2, 5) for i in range(2)])
    cs = O.checksum(rg.state_dict())
    out["g6_keys"] = np.array(list(cs.keys())); out["g6_checksums"] = np.stack(list(cs.values()))

    np.savez_compressed(os.path.join(HERE, "krn_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "krn_golden.npz"), {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("g4_train") or k.startswith("g7_l")})


if __name__ == "__main__":
    main()
