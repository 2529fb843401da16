"""TEST INFRASTRUCTURE (oracle/): CPU restatement of the KRN / RevGrad hot path in plain functional PyTorch.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module -- never the product.

What it restates (reference file:line):
  * ConvDw, RouterV2, KeypointRegressionNet.forward, head de-interleave + summed MSE   park2019.py:32-58,60-80,126-165
  * GradientReversalFunction, RevGrad.forward (hooked feature = base[-1] output)        revgrad.py:36-56,82-96
  * the KRN training step order  forward -> zero_grad -> backward -> clip_grad_norm_(1.0) -> step   trainer.py:72-98
  * the DANN step (alpha schedule, two forwards, BCE-with-logits on ones/zeros)         dann.py:68-100
  * torchvision==0.9.0 mobilenet_v2.features[:-1] (third-party, NOT in /root/reference; requirements.txt:4, call site
    park2019.py:107-108): restated from the published architecture -- stem 3x3/s2 conv 3->32 + BN + ReLU6, then
    InvertedResidual (t,c,n,s) = (1,16,1,1),(6,24,2,2),(6,32,3,2),(6,64,4,2),(6,96,3,1),(6,160,3,2),(6,320,1,1),
    each [1x1 expand+BN+ReLU6 if t!=1] -> dw3x3(stride)+BN+ReLU6 -> 1x1 project+BN, skip iff stride 1 and Cin==Cout.
    PARITY UNPINNED against torchvision itself for this sub-graph: the reference holds no test or golden vector at the
    torchvision boundary and torchvision is not installed here; what is pinned is the parameter count (5 643 862), the
    350 state-dict keys / shapes, the block-13 tap depth (96), the [B,320,7,7] feature the reference's own code relies
    on, and the arithmetic against an independent third-party implementation of the same published architecture
    (Hugging Face transformers' MobileNetV2Model with these weights: tests/test_backbone_pin_cpu.py, 1e-9 in float64).
Everything the reference itself wrote is pinned by tests/golden/*.npz (generated from the imported reference modules by
tests/golden/make_golden.py).

State is a flat dict name -> tensor with the reference's state_dict keys, so the same dict drives the oracle, the
golden generator and the HIP plan.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import portable_rng as prng

MBV2_CFG = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
BN_EPS = 1e-5
BN_MOM = 0.1


def block_specs():
    """[(index, t, cin, cout, stride)] for features[1..17]"""
    out, cin, k = [], 32, 1
    for t, c, n, s in MBV2_CFG:
        for i in range(n):
            out.append((k, t, cin, c, s if i == 0 else 1))
            cin = c
            k += 1
    return out


def krn_param_shapes(num_keypoints=11, dann=False):
    """OrderedDict key -> shape in reference state_dict order (parameters and BN buffers)."""
    sd = OrderedDict()
    pre = "net." if dann else ""

    def conv(name, cout, cin, k, bias=False):
        sd[pre + name + ".weight"] = (cout, cin, k, k)
        if bias:
            sd[pre + name + ".bias"] = (cout,)

    def bn(name, c):
        sd[pre + name + ".weight"] = (c,); sd[pre + name + ".bias"] = (c,)
        sd[pre + name + ".running_mean"] = (c,); sd[pre + name + ".running_var"] = (c,)
        sd[pre + name + ".num_batches_tracked"] = ()

    conv("base.0.0", 32, 3, 3); bn("base.0.1", 32)
    for k, t, cin, cout, s in block_specs():
        p = "base.%d.conv." % k
        hid = cin * t
        i = 0
        if t != 1:
            conv(p + "0.0", hid, cin, 1); bn(p + "0.1", hid); i = 1
        conv(p + "%d.0" % i, hid, 1, 3); bn(p + "%d.1" % i, hid)
        conv(p + "%d" % (i + 1), cout, hid, 1); bn(p + "%d" % (i + 2), cout)
    for e, cin in ((0, 320), (1, 1024), (2, None), (3, 1280)):
        p = "extras.%d.conv." % e
        if e == 2:
            conv(p + "0", 64, 96, 1); bn(p + "1", 64)
            continue
        conv(p + "0", cin, 1, 3); bn(p + "1", cin)
        conv(p + "3", 1024, cin, 1); bn(p + "4", 1024)
    conv("head.0", 2 * num_keypoints, 1024, 7, bias=True)
    if dann:
        sd["domain_classifier.0.weight"] = (1280, 320, 1, 1); sd["domain_classifier.0.bias"] = (1280,)
        sd["domain_classifier.3.weight"] = (1, 1280, 1, 1); sd["domain_classifier.3.bias"] = (1,)
    return sd


def init_state(num_keypoints=11, dann=False, seed=2021, dtype=torch.float32):
    """Deterministic, box-independent random state (no pretrained weights exist offline)."""
    sd = OrderedDict()
    for name, shape in krn_param_shapes(num_keypoints, dann).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            sd[name] = torch.from_numpy(prng.uniform(name, shape, -0.2, 0.2, seed)).to(dtype)
        elif name.endswith("running_var"):
            sd[name] = torch.from_numpy(prng.uniform(name, shape, 0.5, 1.5, seed)).to(dtype)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            a = math.sqrt(6.0 / fan_in)
            sd[name] = torch.from_numpy(prng.uniform(name, shape, -a, a, seed)).to(dtype)
        elif ".bias" in name and ("head" in name or "domain_classifier" in name):
            sd[name] = torch.from_numpy(prng.uniform(name, shape, -0.1, 0.1, seed)).to(dtype)
        elif name.endswith(".weight"):  # BN gamma
            sd[name] = torch.from_numpy(prng.uniform(name, shape, 0.5, 1.5, seed)).to(dtype)
        else:  # BN beta
            sd[name] = torch.from_numpy(prng.uniform(name, shape, -0.2, 0.2, seed)).to(dtype)
    return sd


def synth_batch(B, num_keypoints=11, seed=2021, tag="src", hw=224):
    """images U[0,1) NCHW f32 (transforms.py:192-196 range) and keypoint targets U[0,1) [B,2,K]"""
    x = torch.from_numpy(prng.uniform("img/" + tag, (B, 3, hw, hw), 0.0, 1.0, seed))
    y = torch.from_numpy(prng.uniform("kpt/" + tag, (B, 2, num_keypoints), 0.0, 1.0, seed))
    return x, y


def param_names(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


# ------------------------------------------------------------------------------------------------- functional net
class _Net:
    momentum = BN_MOM  # class-level knob: tests set 1.0 to make running stats equal the batch statistics
    # class-level knob: emulate the bf16 storage/operand rounding points of the HIP path (straight-through in backward):
    #   every convolution output is stored as bf16; matrix-core operands (input and weight of the 1x1 convs,
    #   the head and the domain conv; rounds 1-5: the stem's as well) are rounded to bf16; depthwise arithmetic stays f32 on the small maps, and on the maps
    #   at least DW_TILE_MIN wide (round 4: csrc/dwconv_tile.hip, forward) the activated operand is staged as bf16 and the
    #   taps are bf16 (v_dot2c_f32_bf16, f32 accumulation); materialised tensors (skip outputs, concat) bf16.
    quant = False
    DW_TILE_MIN = 28
    # class-level knob: {BatchNorm name (without prefix handling: the key bn() receives, prefix included): tensor} -- the raw convolution
    # outputs of ANOTHER implementation's forward pass.  bn() then normalises THAT tensor (straight-through to the convolution it
    # replaces), so the backward pass runs through the other implementation's forward state: what the parity tests use to hold the HIP
    # backward kernels to float64 without the forward pass's rounding noise in between (tests/test_parity_conditioned_gpu.py)
    forced = None

    def __init__(self, sd, training, prefix=""):
        self.sd, self.training, self.p = sd, training, prefix

    @classmethod
    def q(cls, t):
        if not cls.quant:
            return t
        return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())

    def conv(self, x, name, stride=1, padding=0, groups=1):
        w = self.sd[self.p + name + ".weight"]
        b = self.sd.get(self.p + name + ".bias")
        if groups == 1:  # matrix-core layers (pointwise, head, domain conv): operands rounded.  NOT the stem since round 6: image and stem
            if name != "base.0.0":   # weights enter the matrix cores as hi + lo pairs (csrc/stem_mfma.hip), the product is exact to ~1e-5
                x, w = self.q(x), self.q(w)
        elif x.shape[-1] >= self.DW_TILE_MIN:  # depthwise layers of the large maps: bf16 operand tile and bf16 taps
            x, w = self.q(x), self.q(w)
        z = F.conv2d(x, w, None, stride, padding, 1, groups)
        if b is not None:  # head / domain conv: f32 bias after the (rounded, for the domain conv) product
            return z + b.view(1, -1, 1, 1)
        return self.q(z)

    def bn(self, x, name):
        sd, n = self.sd, self.p + name
        if self.forced is not None and n in self.forced:
            x = x + (self.forced[n].to(x.dtype) - x).detach()
        if self.training:
            sd[n + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[n + ".running_mean"], sd[n + ".running_var"], sd[n + ".weight"], sd[n + ".bias"],
                            self.training, self.momentum, BN_EPS)


def krn_features(sd, x, training, prefix=""):
    """returns (feature [B,320,7,7] = base[-1] output, tap = base[13] output)"""
    net = _Net(sd, training, prefix)
    x = F.relu6(net.bn(net.conv(x, "base.0.0", 2, 1), "base.0.1"))
    tap = None
    for k, t, cin, cout, s in block_specs():
        p = "base.%d.conv." % k
        hid = cin * t
        y, i = x, 0
        if t != 1:
            y = F.relu6(net.bn(net.conv(y, p + "0.0"), p + "0.1")); i = 1
        y = F.relu6(net.bn(net.conv(y, p + "%d.0" % i, s, 1, hid), p + "%d.1" % i))
        y = net.bn(net.conv(y, p + "%d" % (i + 1)), p + "%d" % (i + 2))
        x = net.q(x + y) if (s == 1 and cin == cout) else y
        if k == 13:
            tap = x
    return x, tap


def _conv_dw(net, x, e):
    p = "extras.%d.conv." % e
    x = F.relu(net.bn(net.conv(x, p + "0", 1, 1, x.shape[1]), p + "1"))
    return F.relu(net.bn(net.conv(x, p + "3"), p + "4"))


def reorg(x, s=2):
    """space-to-depth with the reference's channel order: out[b,(i*s+j)*C+c,h,w] = in[b,c,h*s+i,w*s+j]"""
    B, C, H, W = x.shape
    x = x.view(B, C, H // s, s, W // s, s).permute(0, 3, 5, 1, 2, 4)  # b, i, j, c, h, w
    return x.reshape(B, s * s * C, H // s, W // s)


def krn_predict(sd, x, training, prefix=""):
    """raw head output [B, 2K] (interleaved x0,y0,x1,...) and the DANN feature"""
    feat, tap = krn_features(sd, x, training, prefix)
    net = _Net(sd, training, prefix)
    h = _conv_dw(net, feat, 0)
    h = _conv_dw(net, h, 1)
    r = F.leaky_relu(net.bn(net.conv(tap, "extras.2.conv.0"), "extras.2.conv.1"), 0.2)
    h = net.q(torch.cat((reorg(r, 2), h), dim=1))
    h = _conv_dw(net, h, 3)
    out = net.conv(h, "head.0")
    return out.reshape(x.shape[0], -1), feat


def krn_loss(out, y):
    """sum_k mean_b (xc-tx)^2 + sum_k mean_b (yc-ty)^2"""
    xc, yc = out[:, 0::2], out[:, 1::2]
    lx = ((xc - y[:, 0]) ** 2).mean(0).sum()
    ly = ((yc - y[:, 1]) ** 2).mean(0).sum()
    return lx + ly, lx, ly


def krn_forward(sd, x, y=None, training=True, prefix=""):
    out, _ = krn_predict(sd, x, training, prefix)
    if y is None:
        return out[:, 0::2], out[:, 1::2]
    return krn_loss(out, y)


class _GRL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return -ctx.lam * g, None


def revgrad_forward(sd, x, y=None, alpha=None, training=True):
    """RevGrad.forward: (pose output, domain logits [B])"""
    out, feat = krn_predict(sd, x, training, "net.")
    pose = krn_loss(out, y) if y is not None else (out[:, 0::2], out[:, 1::2])
    if alpha is None:
        return pose
    d = _GRL.apply(feat, alpha)
    q = _Net.q
    d = q(F.conv2d(q(d), q(sd["domain_classifier.0.weight"])))
    d = q(F.relu(d + sd["domain_classifier.0.bias"].view(1, -1, 1, 1)))
    d = F.avg_pool2d(d, 7)
    d = F.conv2d(d, sd["domain_classifier.3.weight"], sd["domain_classifier.3.bias"])
    return pose, d.reshape(-1)


def dann_alpha(idx, epoch, n_batches, max_epochs):
    p = float(idx + epoch * n_batches) / max_epochs / n_batches
    return 2.0 / (1.0 + math.exp(-10 * p)) - 1.0


# ------------------------------------------------------------------------------------------------- training steps
def _leafify(sd):
    names = param_names(sd)
    for k in names:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    return names


def make_optimizer(kind, params, lr, momentum, weight_decay):
    """build.py:60-78 (cfg.momentum doubles as RMSprop alpha / Adam beta1)"""
    if kind == "sgd":
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    if kind == "rmsprop":
        return torch.optim.RMSprop(params, lr=lr, alpha=momentum, weight_decay=weight_decay)
    if kind == "adam":
        return torch.optim.Adam(params, lr=lr, betas=(momentum, 0.999), weight_decay=weight_decay)
    if kind == "adamw":
        return torch.optim.AdamW(params, lr=lr, betas=(momentum, 0.999), weight_decay=weight_decay)
    raise ValueError(kind)


class KrnTrainer:
    """KRN train step on CPU in the reference's order (trainer.py:72-98): forward, zero_grad, backward,
    clip_grad_norm_(1.0), optimizer.step()."""

    def __init__(self, sd, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01):
        self.sd = sd
        self.names = _leafify(sd)
        self.opt = make_optimizer(kind, [sd[k] for k in self.names], lr, momentum, weight_decay)

    def step(self, x, y):
        loss, lx, ly = krn_forward(self.sd, x, y, training=True)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_([self.sd[k] for k in self.names], 1.0)
        self.opt.step()
        return float(loss), float(lx), float(ly), float(gn)


class DannTrainer:
    """DANN step (dann.py:68-100)."""

    def __init__(self, sd, kind="adamw", lr=1e-3, momentum=0.9, weight_decay=0.01):
        self.sd = sd
        self.names = _leafify(sd)
        self.opt = make_optimizer(kind, [sd[k] for k in self.names], lr, momentum, weight_decay)

    def step(self, xs, ys, xt, alpha):
        B = xs.shape[0]
        self.opt.zero_grad(set_to_none=True)
        (lp, lx, ly), ds = revgrad_forward(self.sd, xs, ys, alpha, True)
        l_src = F.binary_cross_entropy_with_logits(ds, torch.ones(B))
        _, dt = revgrad_forward(self.sd, xt, None, alpha, True)
        l_tgt = F.binary_cross_entropy_with_logits(dt, torch.zeros(B))
        (lp + l_src + l_tgt).backward()
        gn = torch.nn.utils.clip_grad_norm_([self.sd[k] for k in self.names], 1.0)
        self.opt.step()
        return float(lp), float(l_src), float(l_tgt), float(gn)


def checksum(sd, names=None):
    """order-independent digest of a state: per-tensor (sum, sum of squares) in float64"""
    names = names if names is not None else list(sd.keys())
    out = OrderedDict()
    for k in names:
        t = sd[k].detach().double()
        out[k] = np.array([float(t.sum()), float((t * t).sum())])
    return out
