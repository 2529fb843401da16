"""DANN for KRN on the MI355X -- mirror of reference src/nets/revgrad.py:36-96.

GradientReversalFunction keeps the reference's autograd semantics for callers that use it on their own tensors (it is a
pure tensor-view op: forward clone, backward -lambda*g).  Inside RevGrad the reversal is folded into the HIP domain-head
input-gradient GEMM (out_scale = -alpha), and the forward hook on net.base[-1] is replaced by the plan exposing the
[B,320,7,7] feature directly.
"""
import torch
import torch.nn as nn

from .park2019 import HipBackedMixin, KeypointRegressionNet, _backward_arena, _grad_views


class GradientReversalFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = lambda_
        return x.clone()

    @staticmethod
    def backward(ctx, grads):
        return grads.neg() * ctx.lambda_, None


class _RevGradFn(torch.autograd.Function):
    """generic-path autograd bridge: (pose loss, domain logits) of one forward; backward accumulates both"""

    @staticmethod
    def forward(ctx, module, x, y, alpha, slot, *params):
        eng = module.engine()
        pred, scal, dom = eng.forward(x, y, training=module.training, slot=slot, domain=True)
        ctx.module, ctx.B, ctx.slot, ctx.alpha, ctx.pose = module, x.shape[0], slot, float(alpha), y is not None
        if y is None:
            scal = torch.zeros(3, device=pred.device)
        return scal[0].clone(), scal[1:3].clone(), pred, dom

    @staticmethod
    def backward(ctx, gloss, glxy, gpred, gdom):
        eng = ctx.module.engine()
        arena = _backward_arena(eng)              # (reproducible engines: the registered arena, see park2019._backward_arena)
        dl = gdom.contiguous().float() if gdom is not None else torch.zeros(ctx.B, device=arena.device)
        eng.backward(ctx.B, slot=ctx.slot, grads=arena, gscale=float(gloss) if ctx.pose else 0.0, with_pose=ctx.pose,
                     dlogit=dl, alpha=ctx.alpha)
        return (None, None, None, None, None) + _grad_views(eng, arena)


class RevGrad(HipBackedMixin, nn.Module):
    _spb_dann = True

    def __init__(self, num_keypoints, precision=None, deterministic=False):
        super().__init__()
        self.nK = num_keypoints
        self.precision = precision
        self.deterministic = bool(deterministic)
        self.net = KeypointRegressionNet(num_keypoints)
        self.net._spb_owner = False
        self.net.__dict__["_spb_parent"] = self
        self.feature = None  # the reference fills this from a forward hook; the HIP plan keeps it in its workspace
        self.domain_classifier = nn.Sequential(nn.Conv2d(320, 1280, 1, stride=1, padding=0, bias=True), nn.ReLU(inplace=True),
                                               nn.AvgPool2d(7), nn.Conv2d(1280, 1, 1))

    def forward(self, x, y=None, alpha=None):
        """out1 = KRN output (loss, sm) or (xc, yc);  with alpha: (out1, domain logits [B])  (revgrad.py:82-96)"""
        if alpha is None:
            return self.net(x, y)
        eng = self.engine()
        params = [p for _, p in self.named_parameters()]
        slot = 0 if y is not None else 1  # source pass / target pass of one DANN step (dann.py:81,89)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            loss, lxy, pred, dom = _RevGradFn.apply(self, x, y, alpha, slot, *params)
        else:
            pred, scal, dom = eng.forward(x, y, training=self.training, slot=slot, domain=True)
            loss, lxy = (scal[0], scal[1:3]) if y is not None else (None, None)
        if y is not None:
            lx, ly = lxy.tolist()
            out1 = (loss, {"loss_x": lx, "loss_y": ly})
        else:
            out1 = (pred[:, 0::2].cpu(), pred[:, 1::2].cpu())
        return out1, dom
