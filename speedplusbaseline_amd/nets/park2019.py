"""Keypoint Regression Network on the MI355X -- host-side mirror of the reference module surface
(reference src/nets/park2019.py:32-165; backbone torchvision==0.9 mobilenet_v2.features[:-1], park2019.py:107-108).

Same class names, constructor arguments, attributes (nK, base, extras, head, loss), forward contract and state_dict
keys as the reference, so checkpoints and callers move across unchanged.  The torch.nn modules below are PARAMETER
CONTAINERS ONLY: KeypointRegressionNet.forward never calls them -- the arithmetic runs in libspb_hip.so through the
C++ network plan (speedplusbaseline_amd/engine.py).  There is no CPU execution path: forward raises unless the model
has been moved to the GPU.
"""
import logging
import os

import torch
import torch.nn as nn

from ..engine import KrnEngine

logger = logging.getLogger(__name__)

_MBV2 = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def _conv_bn_relu6(cin, cout, k=3, stride=1, groups=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout),
                         nn.ReLU6(inplace=True))


class _InvertedResidualParams(nn.Module):
    """state-dict layout of torchvision's InvertedResidual: self.conv = [expand CBR], dw CBR, project conv, BN"""

    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        hid = cin * expand
        self.use_res_connect = stride == 1 and cin == cout
        layers = []
        if expand != 1:
            layers.append(_conv_bn_relu6(cin, hid, 1))
        layers += [_conv_bn_relu6(hid, hid, 3, stride, groups=hid), nn.Conv2d(hid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, *a, **k):
        raise RuntimeError("backbone blocks are parameter containers; run the whole KeypointRegressionNet (HIP plan)")


def _mobilenet_v2_features_minus_last(weights_path=None):
    feats = [_conv_bn_relu6(3, 32, 3, 2)]
    cin = 32
    for t, c, n, s in _MBV2:
        for i in range(n):
            feats.append(_InvertedResidualParams(cin, c, s if i == 0 else 1, t))
            cin = c
    base = nn.ModuleList(feats)
    for m in base.modules():  # torchvision's from-scratch init
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
    path = weights_path or os.environ.get("SPB_MOBILENETV2_WEIGHTS", "")
    if path:
        sd = torch.load(path, map_location="cpu")
        sd = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.") and not k.startswith("features.18.")}
        base.load_state_dict(sd, strict=True)
        logger.info("   - MobileNetV2 backbone weights loaded from %s", path)
    else:
        logger.warning("   - ImageNet MobileNetV2 weights cannot be downloaded here; set SPB_MOBILENETV2_WEIGHTS to a "
                       "torchvision mobilenet_v2 state_dict file. Using torchvision's from-scratch init.")
    return base


class ConvDw(nn.Module):
    """depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN + ReLU (reference park2019.py:32-58)"""

    def __init__(self, inp, oup, stride):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(inp, inp, 3, stride=stride, padding=1, groups=inp, bias=False), nn.BatchNorm2d(inp), nn.ReLU(inplace=True),
            nn.Conv2d(inp, oup, 1, stride=1, padding=0, bias=False), nn.BatchNorm2d(oup), nn.ReLU(inplace=True))
        self.depth = oup

    def forward(self, x):
        raise RuntimeError("ConvDw is a parameter container here; run the whole KeypointRegressionNet (HIP plan)")


class RouterV2(nn.Module):
    """1x1 conv + BN + LeakyReLU(0.2), space-to-depth by `stride`, concat in front of x1 (reference park2019.py:60-80)"""

    def __init__(self, inp, oup, stride=2):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(inp, oup, 1, stride=1, bias=False), nn.BatchNorm2d(oup), nn.LeakyReLU(0.2, inplace=True))
        self.stride = stride

    def forward(self, x1, x2):
        raise RuntimeError("RouterV2 is a parameter container here; run the whole KeypointRegressionNet (HIP plan)")


class RouterV3(nn.Module):
    """kept importable for drop-in parity; never instantiated by the reference either (park2019.py:82-97)"""

    def __init__(self, inp, oup, stride=1, mode="bilinear"):
        super().__init__()
        self.mode = mode
        self.conv = nn.Sequential(nn.Conv2d(inp, oup, 1, stride=1, bias=False), nn.BatchNorm2d(oup), nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x1, x2):
        raise RuntimeError("RouterV3 has no HIP implementation (dead code in the reference)")


# ------------------------------------------------------------------------------------------------------ engine glue
class HipBackedMixin:
    """Keeps nn.Parameters / buffers as views into the engine's flat device arenas once the module lives on the GPU."""

    _spb_owner = True
    _spb_dann = False

    def _spb_precision(self):
        return getattr(self, "precision", None) or os.environ.get("SPB_PRECISION", "fp32")

    def _spb_names(self):
        return {n: p for n, p in self.named_parameters()}, {n: b for n, b in self.named_buffers()}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if getattr(self, "_spb_owner", True) and not getattr(self, "_spb_in_attach", False):
            dev = next(self.parameters()).device
            if dev.type == "cuda":
                self._spb_attach(dev)
            else:
                self.__dict__["_engine"] = None
        return out

    def _spb_attach(self, dev):
        eng = self.__dict__.get("_engine_obj")
        if eng is None:
            eng = KrnEngine(self.nK, dann=self._spb_dann, deterministic=bool(getattr(self, "deterministic", False)))
            self.__dict__["_engine_obj"] = eng
        params, buffers = self._spb_names()
        old = {n: p.detach().clone() for n, p in params.items()}
        oldb = {n: b.detach().clone() for n, b in buffers.items()}
        eng.attach(dev, self._spb_precision())
        names = [i[0] for i in eng.param_infos]
        if sorted(names) != sorted(params.keys()):
            raise RuntimeError("module parameters do not match the HIP plan: %s" % (set(names) ^ set(params.keys())))
        with torch.no_grad():
            for info in eng.param_infos:
                v = eng.param_view(info)
                v.copy_(old[info[0]])
                params[info[0]].data = v
                params[info[0]].grad = None
            for name, shape, off, numel in eng.buffer_infos:
                v = eng.buffers[off: off + numel].view(shape)
                v.copy_(oldb[name])
                buffers[name].data = v
            for i, name in enumerate(eng.bn_names):
                eng.nbt[i] = int(oldb[name])
                buffers[name].data = eng.nbt[i]
        self.__dict__["_engine"] = eng

    def engine(self):
        eng = self.__dict__.get("_engine")
        if eng is None:
            raise RuntimeError("%s runs on the MI355X only: move it to the GPU first (model.to('cuda')). There is no CPU "
                               "or eager-PyTorch path in this build." % type(self).__name__)
        return eng


def _backward_arena(eng):
    """Gradient arena of one generic-path backward pass.  Reproducible engines accumulate exactly only into regions registered with the
    library (spb_det_register): the bound arena `eng.grads`, so that is where the pass runs there -- an unregistered temporary made
    every float atomic a counted miss and the plan's flush fail with SPB_E_STATE -- and the result is copied out."""
    if eng.deterministic:
        eng.grads.zero_()
        return eng.grads
    return torch.zeros_like(eng.params)


def _grad_views(eng, arena):
    if arena is eng.grads:
        arena = arena.clone()
    return tuple(eng.param_view(i, arena) for i in eng.param_infos)


class _KrnLossFn(torch.autograd.Function):
    """autograd bridge of the generic path (loss.backward() with any torch optimizer): one HIP forward, one HIP backward"""

    @staticmethod
    def forward(ctx, module, x, y, *params):
        eng = module.engine()
        pred, scal, _ = eng.forward(x, y, training=module.training, slot=0)
        ctx.module, ctx.B = module, x.shape[0]
        return scal[0].clone(), scal[1:3].clone(), pred

    @staticmethod
    def backward(ctx, gloss, glxy, gpred):
        eng = ctx.module.engine()
        arena = _backward_arena(eng)
        eng.use_loss_scale(ctx.B, 0, False)       # float16: the upstream gradient (the caller's scaler.scale(loss)) carries the scale here
        try:
            eng.backward(ctx.B, slot=0, grads=arena, gscale=float(gloss))
        finally:
            eng.use_loss_scale(ctx.B, 0, True)
        return (None, None, None) + _grad_views(eng, arena)


class KeypointRegressionNet(HipBackedMixin, nn.Module):
    def __init__(self, num_keypoints, precision=None, backbone_weights=None, deterministic=False):
        super().__init__()
        self.nK = num_keypoints
        self.precision = precision
        self.deterministic = bool(deterministic)      # the reproducible library build (KrnEngine(deterministic=True)); fp32 / bf16
        self.base = _mobilenet_v2_features_minus_last(backbone_weights)
        self.extras = nn.ModuleList([ConvDw(320, 1024, stride=1), ConvDw(1024, 1024, stride=1), RouterV2(96, 64),
                                     ConvDw(1024 + 64 * 4, 1024, stride=1)])
        self.head = nn.ModuleList([nn.Conv2d(1024, 2 * num_keypoints, kernel_size=7)])
        self.loss = nn.MSELoss(reduction="mean")

    def forward(self, x, y=None):
        """training: (loss, {'loss_x': float, 'loss_y': float}); testing: (xc.cpu(), yc.cpu())  (park2019.py:126-165)"""
        eng = self.engine() if self._spb_owner else self._spb_parent.engine()
        owner = self if self._spb_owner else self._spb_parent
        if y is not None:
            params = [p for _, p in owner.named_parameters()]
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                loss, lxy, _ = _KrnLossFn.apply(owner, x, y, *params)
            else:
                _, scal, _ = eng.forward(x, y, training=self.training, slot=0)
                loss, lxy = scal[0], scal[1:3]
            lx, ly = lxy.tolist()  # the reference returns host floats here too (park2019.py:159-160)
            return loss, {"loss_x": lx, "loss_y": ly}
        pred, _, _ = eng.forward(x, None, training=self.training, slot=0)
        return pred[:, 0::2].cpu(), pred[:, 1::2].cpu()
