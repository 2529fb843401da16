// The per-element update rule shared by the arena-wide optimizer pass (elemwise.hip) and the fully connected layers' fused
// weight-gradient + update kernel (spn_fc.hip).  Reference: build.py:60-78 (torch.optim.SGD / RMSprop / Adam / AdamW),
// trainer.py:177-184 (clip_grad_value_ before the step).
#pragma once
#include "common.h"

namespace {

// one element of the update; m / v are the moment values (read and written back by the caller when the kind uses them)
__device__ __forceinline__ float optim_one(const spb_optim_args_t& a, float gs, float lr, float bias_c1, float bias_c2, float p, float g,
                                           float& m, float& v) {
  g *= gs;
  if (a.clip_value > 0.f) g = fminf(fmaxf(g, -a.clip_value), a.clip_value);
  if (a.kind == 3) {  // adamw (decoupled decay)
    p *= 1.f - lr * a.weight_decay;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / sqrtf(bias_c2) + a.eps;
    p -= (lr / bias_c1) * (m / denom);
  } else if (a.kind == 2) {  // adam (L2 folded into the gradient)
    g += a.weight_decay * p;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / sqrtf(bias_c2) + a.eps;
    p -= (lr / bias_c1) * (m / denom);
  } else if (a.kind == 1) {  // rmsprop (alpha = beta2 slot), no momentum, not centred
    g += a.weight_decay * p;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    p -= lr * g / (sqrtf(v) + a.eps);
  } else {  // sgd with momentum (beta1), dampening 0
    g += a.weight_decay * p;
    if (a.beta1 != 0.f && a.m) {
      m = a.first_step ? g : a.beta1 * m + g;
      g = m;
    }
    p -= lr * g;
  }
  return p;
}

}  // namespace
