import sys, types, torch
sys.argv = [sys.argv[0]]
from oracle import krn_oracle as O
from src.nets import get_model
from torch.nn.utils import clip_grad_norm_
dev = "cuda"
x, y = O.synth_batch(4)
sd0 = O.init_state(11)
def cfg(**kw):
    c = types.SimpleNamespace(model_name="krn", num_keypoints=11, num_classes=5000, dann=True, optimizer="sgd", lr=0.05,
                              momentum=0.9, weight_decay=5e-5, fp16=False, precision="fp32", max_epochs=5, texture_ratio=0.5, deterministic=True)
    c.__dict__.update(kw); return c
def run(steps):
    model = get_model(cfg()); model.net.load_state_dict(sd0, strict=True)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    model = model.to(dev).train()
    out = []
    for _ in range(steps):
        (loss, _), dom = model(x.to(dev), y.to(dev), alpha=0.3)
        l2 = loss + torch.nn.functional.binary_cross_entropy_with_logits(dom, torch.ones_like(dom))
        opt.zero_grad(set_to_none=True)
        l2.backward()
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        out.append((float(loss), dom.detach().cpu().clone(), g))
    torch.cuda.synchronize()
    print("misses", model.engine().det_misses())
    return out
a, b = run(2), run(2)
for s in range(2):
    print("step", s, "loss", a[s][0], b[s][0], "dom equal", torch.equal(a[s][1], b[s][1]))
    bad = [(n, float((a[s][2][n] - b[s][2][n]).abs().max())) for n in a[s][2] if not torch.equal(a[s][2][n], b[s][2][n])]
    print("   grads differing:", len(bad), bad[:8])
