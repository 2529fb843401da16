// Spacecraft Pose Network (AlexNet trunk + two attitude heads) building blocks, forward and backward.
// Reference: src/nets/spn.py:50-143 (SpacecraftPoseNet), :37-48 (softmax_cross_entropy_with_logits), step order and loss
// assembly in src/core/trainer.py:136-185.
//
// First version of this row.  Every convolution and fully connected layer runs on the matrix cores through the
// pointwise GEMM kernels of gemm_pw.hip (spb_pwconv_gemm with the bias + ReLU epilogue, spb_pwconv_wgrad); this file
// supplies what turns AlexNet into GEMMs and the layers in between, NHWC throughout:
//   im2col / im2col_rgb   K x K patches -> rows of [M = B*OH*OW, K*K*C] (zero padding, stride); grouped convolutions
//                         use the full patch with block-diagonal weights (their FLOPs are irrelevant at this size)
//   col2im                the adjoint gather for the input gradient (stride 1)
//   maxpool 3x3 s2        forward with argmax byte, backward as a gather over the (<= 4) windows covering a pixel
//   lrn (size 2)          nn.LocalResponseNorm(2, alpha, beta, k): out[c] = x[c] * (k + alpha/2 (x[c-1]^2 + x[c]^2))^-beta
//   relu_drop_bwd         gradient through ReLU (and inverted dropout): g = dy * (y > 0) * scale
//   dropout               keep-mask from a counter hash (seed, element), y *= mask / (1 - p), mask kept for the test
//   softce                -sum t * log_softmax(x) per row (mean over rows), and its gradient
//   colsum                bias gradients
// The SPN step is bound by weight traffic (152 M parameters), not by these kernels.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// dst[m][k], k = gi*Kg + (ky*KW + kx)*cig + c_local for channel c = gi*cig + c_local of group gi (Kg = Kpad / groups; one
// contiguous column slab per group, so a grouped convolution is one dense GEMM per group); 8 channels per thread
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C, int KH, int KW, int st,
                              int pad, int OH, int OW, int Kpad, int groups) {
  const int CV = C >> 3, cig = C / groups, Kg = Kpad / groups;
  const long long total = (long long)B * OH * OW * KH * KW * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int tap = (int)(r % (KH * KW)); r /= KH * KW;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const int iy = oy * st - pad + tap / KW, ix = ox * st - pad + tap % KW;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) ld8<T>(src + ((size_t)(b * H + iy) * W + ix) * C + cv * 8, v);
    const int c = cv * 8, gi = c / cig;
    st8<T>(dst + ((size_t)(b * OH + oy) * OW + ox) * Kpad + (size_t)gi * Kg + (size_t)tap * cig + (c - gi * cig), v);
  }
}

// first layer: fp32 NCHW image, C = 3: k = (ci*KH + ky)*KW + kx (the order of nn.Conv2d's own weight rows, and the one in which
// consecutive k are consecutive pixels of an image row), zero padded up to Kpad; one thread = 8 consecutive k = one 16-byte store
template <typename T>
__global__ void im2col_rgb_kernel(const float* __restrict__ x, T* __restrict__ dst, int B, int H, int W, int KH, int KW, int st,
                                  int OH, int OW, int Kpad) {
  const int KV = Kpad >> 3, KK = 3 * KH * KW;
  const long long total = (long long)B * OH * OW * KV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int kv = (int)(i % KV);
    long long r = i / KV;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kv * 8 + j;
      v[j] = 0.f;
      if (k < KK) {
        const int ci = k / (KH * KW), rem = k - ci * KH * KW, ky = rem / KW, kx = rem - ky * KW;
        v[j] = x[((size_t)(b * 3 + ci) * H + oy * st + ky) * W + ox * st + kx];   // valid padding
      }
    }
    st8<T>(dst + i * 8, v);
  }
}

// The same column matrix (bf16), one workgroup per (image, output row): the KH input rows of the three colours are read with
// coalesced loads, ALL in flight at once (<= RGB_NLD per thread), rounded to bf16 into LDS, and the output row's OW column-matrix
// rows are then written as 16-byte vectors gathered from LDS.  The per-element kernel above issues 8 dependent-address scalar
// loads per 16-byte store (67 us for the 71 MB of conv1 at bs=32, 85-95 us beside the parameter update).
constexpr int RGB_NLD = 32;
__global__ __launch_bounds__(256) void im2col_rgb_band_kernel(const float* __restrict__ x, bf16_t* __restrict__ dst, int H, int W, int KH,
                                                              int KW, int st, int OH, int OW, int Kpad) {
  extern __shared__ __attribute__((aligned(16))) char smem_rgb[];
  bf16_t* band = reinterpret_cast<bf16_t*>(smem_rgb);          // [3][KH][W]
  const int t = threadIdx.x;
  const int b = blockIdx.x / OH, oy = blockIdx.x % OH;
  const int total = 3 * KH * W;
  const float* xb = x + (size_t)b * 3 * H * W;
  float v[RGB_NLD];
#pragma unroll
  for (int u = 0; u < RGB_NLD; ++u) {
    const int e = t + 256 * u, ec = e < total ? e : total - 1;
    const int xx = ec % W, r = ec / W, ci = r / KH, ky = r - ci * KH;
    v[u] = xb[((size_t)ci * H + oy * st + ky) * W + xx];
  }
#pragma unroll
  for (int u = 0; u < RGB_NLD; ++u) {
    const int e = t + 256 * u;
    if (e < total) band[e] = f2bf(v[u]);
  }
  __syncthreads();
  const int KV = Kpad >> 3, KK = 3 * KH * KW, taps = KH * KW;
  bf16_t* out = dst + (size_t)(b * OH + oy) * OW * Kpad;
  for (int i = t; i < OW * KV; i += 256) {
    const int ox = i / KV, kv = i - ox * KV;
    int k = kv * 8;
    int ci = k / taps, rem = k - ci * taps, ky = rem / KW, kx = rem - ky * KW;
    unsigned h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = (k + j < KK) ? band[(ci * KH + ky) * W + ox * st + kx] : 0u;
      if (++kx == KW) { kx = 0; if (++ky == KH) { ky = 0; ++ci; } }
    }
    *reinterpret_cast<uint4*>(out + (size_t)i * 8) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  }
}

// dx[b,iy,ix,c] = sum over taps of dcol[(b, iy+pad-ky, ix+pad-kx)][(ky*KW+kx)*C + c]   (stride 1)
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int B, int H, int W, int C, int KH, int KW, int pad,
                              int OH, int OW, int Kpad, int groups) {
  const int CV = C >> 3, cig = C / groups, Kg = Kpad / groups;
  const long long total = (long long)B * H * W * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int c = cv * 8, gi = c / cig;
    const size_t cbase = (size_t)gi * Kg + (c - gi * cig);
    for (int ky = 0; ky < KH; ++ky) {
      const int oy = iy + pad - ky;
      if (oy < 0 || oy >= OH) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int ox = ix + pad - kx;
        if (ox < 0 || ox >= OW) continue;
        float v[8];
        ld8<T>(dcol + ((size_t)(b * OH + oy) * OW + ox) * Kpad + cbase + (size_t)(ky * KW + kx) * cig, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    }
    st8<T>(dx + i * 8, acc);
  }
}

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ arg, int B, int H, int W,
                                   int C, int OH, int OW) {
  const int CV = C >> 3;
  const long long total = (long long)B * OH * OW * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    float m[8]; unsigned char am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -3.0e38f; am[j] = 0; }
    for (int k = 0; k < 9; ++k) {   // scan order of the reference kernel: first maximum wins
      float v[8];
      ld8<T>(x + ((size_t)(b * H + oy * 2 + k / 3) * W + ox * 2 + k % 3) * C + cv * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (v[j] > m[j]) { m[j] = v[j]; am[j] = (unsigned char)k; }
    }
    st8<T>(y + i * 8, m);
    if (arg) {
#pragma unroll
      for (int j = 0; j < 8; ++j) arg[i * 8 + j] = am[j];
    }
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ arg, const T* __restrict__ y,
                                   T* __restrict__ dx, int B, int H, int W, int C, int OH, int OW) {
  const int CV = C >> 3;
  const long long total = (long long)B * H * W * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int oy = iy >= 1 ? (iy - 1) / 2 : 0; oy <= iy / 2 && oy < OH; ++oy) {   // windows 2oy..2oy+2 containing iy
      const int ky = iy - 2 * oy;
      if (ky < 0 || ky > 2) continue;
      for (int ox = ix >= 1 ? (ix - 1) / 2 : 0; ox <= ix / 2 && ox < OW; ++ox) {
        const int kx = ix - 2 * ox;
        if (kx < 0 || kx > 2) continue;
        const size_t o = (((size_t)(b * OH + oy) * OW + ox) * CV + cv) * 8;
        float g[8];
        ld8<T>(dy + o, g);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (arg[o + j] == ky * 3 + kx) acc[j] += g[j];
      }
    }
    if (y) {        // the pooled tensor was a ReLU output: its backward in the same pass (relu_bwd: g * (y > 0))
      float yv[8];
      ld8<T>(y + i * 8, yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = yv[j] > 0.f ? acc[j] : 0.f;
    }
    st8<T>(dx + i * 8, acc);
  }
}

// nn.LocalResponseNorm(2): s[c] = k + alpha/2 * (x[c-1]^2 + x[c]^2);  y[c] = x[c] * s[c]^-beta
template <typename T>
__global__ void lrn_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long npix, int C, float alpha, float beta, float k) {
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float xc = ldf<T>(x + i), xp = c > 0 ? ldf<T>(x + i - 1) : 0.f;
    const float s = k + 0.5f * alpha * (xp * xp + xc * xc);
    stf<T>(y + i, xc * powf(s, -beta));
  }
}
// dx[c] = g[c] s[c]^-b - alpha*beta*x[c] * (g[c] x[c] s[c]^(-b-1) + g[c+1] x[c+1] s[c+1]^(-b-1))
template <typename T>
__global__ void lrn_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ dx, long long npix, int C, float alpha,
                               float beta, float k) {
  const long long total = npix * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float xc = ldf<T>(x + i), xp = c > 0 ? ldf<T>(x + i - 1) : 0.f;
    const float gc = ldf<T>(g + i);
    const float sc = k + 0.5f * alpha * (xp * xp + xc * xc);
    float t = gc * xc * powf(sc, -beta - 1.f);
    if (c + 1 < C) {
      const float xn = ldf<T>(x + i + 1), gn = ldf<T>(g + i + 1);
      const float sn = k + 0.5f * alpha * (xc * xc + xn * xn);
      t += gn * xn * powf(sn, -beta - 1.f);
    }
    stf<T>(dx + i, gc * powf(sc, -beta) - alpha * beta * xc * t);
  }
}

// g = dy * (y > 0) * scale [+ add]   (ReLU, and inverted dropout when y is the post-dropout activation)
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ add, T* __restrict__ g,
                                long long n, float scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = ldf<T>(dy + i);
    if (add) v += ldf<T>(add + i);
    stf<T>(g + i, ldf<T>(y + i) > 0.f ? v * scale : 0.f);
  }
}

__device__ __forceinline__ unsigned hash32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}
// nn.Dropout(p): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from a counter hash of (seed, element index)
template <typename T>
__global__ void dropout_kernel(T* __restrict__ y, unsigned char* __restrict__ mask, long long n, float p, unsigned long long seed,
                               int use_given_mask) {
  const float scale = 1.f / (1.f - p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned char keep;
    if (use_given_mask) keep = mask[i];
    else {
      keep = (hash32(seed * 0x100000001B3ull + (unsigned long long)i) * (1.0f / 4294967296.0f)) >= p ? 1 : 0;
      mask[i] = keep;
    }
    stf<T>(y + i, keep ? ldf<T>(y + i) * scale : 0.f);
  }
}

// one workgroup per row: loss_row = lse * sum(t) - sum(t*x);  dlogits = (softmax * sum(t) - t) * gscale
// out[0] += weight * mean-over-rows contribution, out[slot] += unweighted mean (for the two reported losses)
template <typename T>
__global__ __launch_bounds__(256) void softce_kernel(const T* __restrict__ logits, const float* __restrict__ target, T* __restrict__ dlogits,
                                                     float* out, int slot, int Bn, int Cn, float weight, float* rows,
                                                     const float* gscale = nullptr) {
  __shared__ float red[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const T* x = logits + (size_t)b * Cn;
  const float* tg = target + (size_t)b * Cn;
  float m = -3.0e38f;
  for (int c = t; c < Cn; c += 256) m = fmaxf(m, ldf<T>(x + c));
  red[t] = m; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] = fmaxf(red[t], red[t + s]); __syncthreads(); }
  m = red[0]; __syncthreads();
  float se = 0.f, st = 0.f, sx = 0.f;
  for (int c = t; c < Cn; c += 256) { const float v = ldf<T>(x + c); se += expf(v - m); st += tg[c]; sx += tg[c] * v; }
  red[t] = se; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  se = red[0]; __syncthreads();
  red[t] = st; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  st = red[0]; __syncthreads();
  red[t] = sx; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  sx = red[0];
  const float lse = m + logf(se);
  if (t == 0) {
    const float l = (lse * st - sx) / (float)Bn;
    if (out) {
      atomicAdd(out, weight * l);
      atomicAdd(out + slot, l);
    }
    if (rows) rows[b] = lse * st - sx;      // reduction='none' (spn.py:43-44): the per-sample loss
  }
  if (dlogits) {
    const float gs = weight / (float)Bn * (gscale ? *gscale : 1.f);   // gscale: the AMP loss scale (device scalar; the loss value stays unscaled)
    for (int c = t; c < Cn; c += 256) stf<T>(dlogits + (size_t)b * Cn + c, (expf(ldf<T>(x + c) - lse) * st - tg[c]) * gs);
  }
}

// Class-split form of softce_kernel for wide rows (5000 classes: one workgroup per row walked the row three times, 31 us on 32
// workgroups).  Stage 1: grid (B, SCE_S), each workgroup reduces ITS class range of the row to (max, sum exp(x - max), sum t, sum t*x) in
// `part` [B][SCE_S][4].  Stage 2: grid (B, SCE_S), every workgroup recombines the SCE_S partials of its row in the same fixed order (so all
// of them get the identical log-sum-exp), workgroup (b, 0) adds the row's loss, and each writes the gradient of its class range.
constexpr int SCE_S = 8;
template <typename T>
__device__ __forceinline__ float sce_block_reduce(float v, float* red, int t, bool is_max) {
  red[t] = v; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] = is_max ? fmaxf(red[t], red[t + s]) : red[t] + red[t + s]; __syncthreads(); }
  const float r = red[0]; __syncthreads();
  return r;
}
template <typename T>
__global__ __launch_bounds__(256) void softce_part_kernel(const T* __restrict__ logits, const float* __restrict__ target, float* __restrict__ part, int Cn) {
  __shared__ float red[256];
  const int b = blockIdx.x, sp = blockIdx.y, t = threadIdx.x;
  const int per = (Cn + SCE_S - 1) / SCE_S, c0 = sp * per, c1 = min(Cn, c0 + per);
  const T* x = logits + (size_t)b * Cn;
  const float* tg = target + (size_t)b * Cn;
  float m = -3.0e38f;
  for (int c = c0 + t; c < c1; c += 256) m = fmaxf(m, ldf<T>(x + c));
  m = sce_block_reduce<T>(m, red, t, true);
  float se = 0.f, st = 0.f, sx = 0.f;
  for (int c = c0 + t; c < c1; c += 256) { const float v = ldf<T>(x + c); se += expf(v - m); st += tg[c]; sx += tg[c] * v; }
  se = sce_block_reduce<T>(se, red, t, false);
  st = sce_block_reduce<T>(st, red, t, false);
  sx = sce_block_reduce<T>(sx, red, t, false);
  if (t == 0) {
    float* p = part + ((size_t)b * SCE_S + sp) * 4;
    p[0] = m; p[1] = se; p[2] = st; p[3] = sx;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void softce_fin_kernel(const T* __restrict__ logits, const float* __restrict__ target, T* __restrict__ dlogits,
                                                         float* out, int slot, int Bn, int Cn, float weight, float* rows,
                                                         const float* __restrict__ part, const float* gscale) {
  const int b = blockIdx.x, sp = blockIdx.y, t = threadIdx.x;
  const float* p = part + (size_t)b * SCE_S * 4;
  float m = -3.0e38f;
#pragma unroll
  for (int i = 0; i < SCE_S; ++i) m = fmaxf(m, p[i * 4]);
  float se = 0.f, st = 0.f, sx = 0.f;
#pragma unroll
  for (int i = 0; i < SCE_S; ++i) { se += p[i * 4 + 1] * expf(p[i * 4] - m); st += p[i * 4 + 2]; sx += p[i * 4 + 3]; }
  const float lse = m + logf(se);
  if (sp == 0 && t == 0) {
    const float l = (lse * st - sx) / (float)Bn;
    if (out) { atomicAdd(out, weight * l); atomicAdd(out + slot, l); }
    if (rows) rows[b] = lse * st - sx;
  }
  if (dlogits) {
    const int per = (Cn + SCE_S - 1) / SCE_S, c0 = sp * per, c1 = min(Cn, c0 + per);
    const T* x = logits + (size_t)b * Cn;
    const float* tg = target + (size_t)b * Cn;
    const float gs = weight / (float)Bn * (gscale ? *gscale : 1.f);
    for (int c = c0 + t; c < c1; c += 256) stf<T>(dlogits + (size_t)b * Cn + c, (expf(ldf<T>(x + c) - lse) * st - tg[c]) * gs);
  }
}
// rows wider than this use the class-split pair of launches (library-owned scratch for the partials, one row set per stream slot)
static int g_softce_split_min_c = 2048;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_softce_split(int min_classes) { g_softce_split_min_c = min_classes; return 0; }
#endif
template <typename T>
static int softce_split(const void* logits, const float* target, void* dlogits, float* out, int slot, int B, int C, float weight, float* rows,
                        const float* gscale, hipStream_t s) {
  // partials: [slot 0..2][B <= 4096][SCE_S][4] floats; the slots keep the two heads (concurrent streams) and the rows form apart
  static float* part = nullptr;
  if (!part && hipMalloc(&part, (size_t)3 * 4096 * SCE_S * 4 * sizeof(float)) != hipSuccess) return SPB_E_STATE;
  if (B > 4096) return SPB_E_SHAPE;
  float* pp = part + (size_t)(slot < 0 || slot > 2 ? 0 : slot) * 4096 * SCE_S * 4;
  hipLaunchKernelGGL(softce_part_kernel<T>, dim3(B, SCE_S), dim3(256), 0, s, (const T*)logits, target, pp, C);
  hipLaunchKernelGGL(softce_fin_kernel<T>, dim3(B, SCE_S), dim3(256), 0, s, (const T*)logits, target, (T*)dlogits, out, slot, B, C, weight, rows,
                     (const float*)pp, gscale);
  return 0;
}

// out[n] += sum_m g[m][n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ out, long long M, int N,
                                                     long long rows_per_block) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const long long m0 = (long long)blockIdx.y * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float s = 0.f;
  for (long long m = m0; m < m1; ++m) s += ldf<T>(g + m * N + n);
  atomicAdd(out + n, s);
}

// N % 8 == 0, N <= 2048: a block walks its rows RP at a time, every thread owning 8 adjacent columns (16-byte loads for
// bf16), then sums its row lanes through LDS: M*N/2048 blocks' worth of parallelism instead of one thread per column
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* __restrict__ g, float* __restrict__ out, long long M, int N,
                                                      long long rows_per_block) {
  __shared__ float red[2048];
  const int CV = N >> 3, RP = 256 / CV;
  const int t = threadIdx.x, cv = t % CV, r = t / CV;
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < RP)
    for (long long m = m0 + r; m < m1; m += RP) {
      float v[8];
      ld8<T>(g + m * N + cv * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
  if (r < RP)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r * N + cv * 8 + j] = s[j];
  __syncthreads();
  for (int n = t; n < N; n += 256) {
    float a = 0.f;
    for (int q = 0; q < RP; ++q) a += red[q * N + n];
    atomicAdd(out + n, a);
  }
}

// nn.Conv2d weight [Cout][Cin/G][KH][KW] -> Wp [Cout][Kg], row co = its group's filter in (ky,kx,c_local) order (zero padded up
// to Kg), i.e. G stacked [Cout/G][Kg] GEMM operands; WpT [G][Kg][Cout/G] holds the per-group transposes (input gradient)
template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ W, T* __restrict__ Wp, T* __restrict__ WpT, int Cout, int Cin, int G, int KH,
                                 int KW, int Kg, int chw) {
  const long long total = (long long)Cout * Kg;
  const int cog = Cout / G, cig = Cin / G;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i / Kg), k = (int)(i % Kg);
    float v = 0.f;
    if (k < KH * KW * cig) v = chw ? W[(size_t)co * cig * KH * KW + k] : W[((size_t)co * cig + (k % cig)) * KH * KW + k / cig];
    stf<T>(Wp + i, v);
    const int gi = co / cog;
    if (WpT) stf<T>(WpT + ((size_t)gi * Kg + k) * cog + (co - gi * cog), v);
  }
}
__global__ void unpack_conv_grad_kernel(const float* __restrict__ dWp, float* __restrict__ dW, int Cout, int Cin, int G, int KH, int KW,
                                        int Kg, int chw) {
  const int cig = Cin / G;
  const long long total = (long long)Cout * cig * KH * KW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % (KH * KW));
    const int ci = (int)((i / (KH * KW)) % cig);
    const int co = (int)(i / ((long long)KH * KW * cig));
    dW[i] = chw ? dWp[(size_t)co * Kg + (size_t)ci * KH * KW + tap] : dWp[(size_t)co * Kg + (size_t)tap * cig + ci];
  }
}

inline unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  return (unsigned)(g > 65535 * 4 ? 65535 * 4 : (g < 1 ? 1 : g));
}

}  // namespace

#define SPN_T(dtype, CALL_BF, CALL_F32) \
  if ((dtype) == SPB_BF16) { CALL_BF; } else if ((dtype) == SPB_F32) { CALL_F32; } else return SPB_E_ARG;

extern "C" int spb_im2col(int dtype, const void* src, void* dst, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                          int Kpad, int groups, spb_stream_t stream) {
  if (!src || !dst || B <= 0 || (C & 7) || Kpad < KH * KW * C || (Kpad & 7) || groups < 1 || (C % groups) || ((C / groups) & 7) ||
      (Kpad % groups))
    return SPB_E_ARG;
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  const long long total = (long long)B * OH * OW * KH * KW * (C >> 3);
  hipStream_t s = (hipStream_t)stream;
  if (Kpad != KH * KW * C) {
    const size_t es = dtype == SPB_BF16 ? 2 : 4;
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)B * OH * OW * Kpad * es, s);
    if (e != hipSuccess) return (int)e;
  }
  SPN_T(dtype, hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad, groups),
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, (float*)dst, B, H, W, C, KH, KW, stride, pad, OH, OW, Kpad, groups))
  SPB_CHECK_LAUNCH();
  return 0;
}

static int g_rgb_band = 1;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_im2col_rgb_band(int on) { g_rgb_band = on; return 0; }
#endif
extern "C" int spb_im2col_rgb(int dtype, const float* x, void* dst, int B, int H, int W, int KH, int KW, int stride, int Kpad,
                              spb_stream_t stream) {
  if (!x || !dst || B <= 0 || Kpad < KH * KW * 3 || (Kpad & 7)) return SPB_E_ARG;
  const int OH = (H - KH) / stride + 1, OW = (W - KW) / stride + 1;
  const long long total = (long long)B * OH * OW * (Kpad >> 3);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SPB_BF16 && g_rgb_band && 3 * KH * W <= 256 * RGB_NLD && (size_t)3 * KH * W * 2 <= 64 * 1024) {
    hipLaunchKernelGGL(im2col_rgb_band_kernel, dim3((unsigned)(B * OH)), dim3(256), (size_t)3 * KH * W * sizeof(bf16_t), s, x, (bf16_t*)dst, H, W,
                       KH, KW, stride, OH, OW, Kpad);
    SPB_CHECK_LAUNCH();
    return 0;
  }
  SPN_T(dtype, hipLaunchKernelGGL(im2col_rgb_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16_t*)dst, B, H, W, KH, KW, stride, OH, OW, Kpad),
        hipLaunchKernelGGL(im2col_rgb_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)dst, B, H, W, KH, KW, stride, OH, OW, Kpad))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_col2im(int dtype, const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int pad, int Kpad,
                          int groups, spb_stream_t stream) {
  if (!dcol || !dx || B <= 0 || (C & 7) || groups < 1 || (C % groups) || ((C / groups) & 7) || (Kpad % groups)) return SPB_E_ARG;
  const int OH = H + 2 * pad - KH + 1, OW = W + 2 * pad - KW + 1;
  const long long total = (long long)B * H * W * (C >> 3);
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dcol, (bf16_t*)dx, B, H, W, C, KH, KW, pad, OH, OW, Kpad, groups),
        hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dcol, (float*)dx, B, H, W, C, KH, KW, pad, OH, OW, Kpad, groups))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_maxpool3s2_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int B, int H, int W, int C,
                                  spb_stream_t stream) {
  if (!x || !y || B <= 0 || (C & 7) || H < 3 || W < 3) return SPB_E_ARG;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const long long total = (long long)B * OH * OW * (C >> 3);
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, argmax, B, H, W, C, OH, OW),
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, (float*)y, argmax, B, H, W, C, OH, OW))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_maxpool3s2_bwd(int dtype, const void* dy, const unsigned char* argmax, void* dx, int B, int H, int W, int C,
                                  spb_stream_t stream) {
  if (!dy || !argmax || !dx || B <= 0 || (C & 7)) return SPB_E_ARG;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const long long total = (long long)B * H * W * (C >> 3);
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dy, argmax, (const bf16_t*)nullptr, (bf16_t*)dx, B, H, W, C, OH, OW),
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dy, argmax, (const float*)nullptr, (float*)dx, B, H, W, C, OH, OW))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_maxpool3s2_relu_bwd(int dtype, const void* dy, const unsigned char* argmax, const void* y, void* dx, int B, int H, int W,
                                       int C, spb_stream_t stream) {
  if (!dy || !argmax || !y || !dx || B <= 0 || (C & 7)) return SPB_E_ARG;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const long long total = (long long)B * H * W * (C >> 3);
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dy, argmax, (const bf16_t*)y, (bf16_t*)dx, B, H, W, C, OH, OW),
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dy, argmax, (const float*)y, (float*)dx, B, H, W, C, OH, OW))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_lrn2_fwd(int dtype, const void* x, void* y, long long npix, int C, float alpha, float beta, float k,
                            spb_stream_t stream) {
  if (!x || !y || npix <= 0 || C <= 0) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(lrn_fwd_kernel<bf16_t>, dim3(grid_for(npix * C)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, npix, C, alpha, beta, k),
        hipLaunchKernelGGL(lrn_fwd_kernel<float>, dim3(grid_for(npix * C)), dim3(256), 0, s, (const float*)x, (float*)y, npix, C, alpha, beta, k))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_lrn2_bwd(int dtype, const void* x, const void* g, void* dx, long long npix, int C, float alpha, float beta,
                            float k, spb_stream_t stream) {
  if (!x || !g || !dx || npix <= 0 || C <= 0) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(lrn_bwd_kernel<bf16_t>, dim3(grid_for(npix * C)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, npix, C, alpha, beta, k),
        hipLaunchKernelGGL(lrn_bwd_kernel<float>, dim3(grid_for(npix * C)), dim3(256), 0, s, (const float*)x, (const float*)g, (float*)dx, npix, C, alpha, beta, k))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_relu_bwd(int dtype, const void* dy, const void* y, const void* add, void* g, long long n, float scale,
                            spb_stream_t stream) {
  if (!dy || !y || !g || n <= 0) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)add, (bf16_t*)g, n, scale),
        hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)dy, (const float*)y, (const float*)add, (float*)g, n, scale))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dropout(int dtype, void* y, unsigned char* mask, long long n, float p, unsigned long long seed, int use_given_mask,
                           spb_stream_t stream) {
  if (!y || !mask || n <= 0 || p < 0.f || p >= 1.f) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)y, mask, n, p, seed, use_given_mask),
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (float*)y, mask, n, p, seed, use_given_mask))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_softce(int dtype, const void* logits, const float* target, void* dlogits, float* out, int slot, int B, int C,
                          float weight, spb_stream_t stream) {
  if (!logits || !target || !out || B <= 0 || C <= 0 || slot < 1) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (C >= g_softce_split_min_c) {
    const int e = dtype == SPB_BF16 ? softce_split<bf16_t>(logits, target, dlogits, out, slot, B, C, weight, nullptr, nullptr, s)
                : dtype == SPB_F32 ? softce_split<float>(logits, target, dlogits, out, slot, B, C, weight, nullptr, nullptr, s) : SPB_E_ARG;
    if (e) return e;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  SPN_T(dtype, hipLaunchKernelGGL(softce_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, target, (bf16_t*)dlogits, out, slot, B, C, weight, (float*)nullptr),
        hipLaunchKernelGGL(softce_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, target, (float*)dlogits, out, slot, B, C, weight, (float*)nullptr))
  SPB_CHECK_LAUNCH();
  return 0;
}
// the same with the gradient multiplied by a device scalar: GradScaler's scale(loss).backward() (reference trainer.py:171-173)
extern "C" int spb_softce_scaled(int dtype, const void* logits, const float* target, void* dlogits, float* out, int slot, int B, int C,
                                 float weight, const float* gscale, spb_stream_t stream) {
  if (!logits || !target || !out || B <= 0 || C <= 0 || slot < 1 || slot > 2) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (C >= g_softce_split_min_c) {
    const int e = dtype == SPB_BF16 ? softce_split<bf16_t>(logits, target, dlogits, out, slot, B, C, weight, nullptr, gscale, s)
                : dtype == SPB_F32 ? softce_split<float>(logits, target, dlogits, out, slot, B, C, weight, nullptr, gscale, s) : SPB_E_ARG;
    if (e) return e;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  SPN_T(dtype, hipLaunchKernelGGL(softce_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, target, (bf16_t*)dlogits, out, slot, B, C, weight, (float*)nullptr, gscale),
        hipLaunchKernelGGL(softce_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, target, (float*)dlogits, out, slot, B, C, weight, (float*)nullptr, gscale))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_softce_rows(int dtype, const void* logits, const float* target, float* rows, int B, int C, spb_stream_t stream) {
  if (!logits || !target || !rows || B <= 0 || C <= 0) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  SPN_T(dtype, hipLaunchKernelGGL(softce_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, target, (bf16_t*)nullptr, (float*)nullptr, 0, B, C, 1.f, rows),
        hipLaunchKernelGGL(softce_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, target, (float*)nullptr, (float*)nullptr, 0, B, C, 1.f, rows))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_colsum(int dtype, const void* g, float* out, long long M, int N, spb_stream_t stream) {
  if (!g || !out || M <= 0 || N <= 0) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!(N & 7) && N <= 2048 && M >= 64) {
    const int RP = 256 / (N >> 3);
    long long rpb = (M + 255) / 256;     // ~256 blocks: each ends in N same-address atomics (~25 ns apiece, serialised)
    rpb = (rpb + RP - 1) / RP * RP;
    const dim3 g8((unsigned)((M + rpb - 1) / rpb));
    SPN_T(dtype, hipLaunchKernelGGL(colsum8_kernel<bf16_t>, g8, dim3(256), 0, s, (const bf16_t*)g, out, M, N, rpb),
          hipLaunchKernelGGL(colsum8_kernel<float>, g8, dim3(256), 0, s, (const float*)g, out, M, N, rpb))
    SPB_CHECK_LAUNCH();
    return 0;
  }
  const long long rpb = M > 4096 ? 1024 : (M > 256 ? 64 : M);
  const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + rpb - 1) / rpb));
  SPN_T(dtype, hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, out, M, N, rpb),
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)g, out, M, N, rpb))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_pack_conv(int dtype, const float* W, void* Wp, void* WpT, int Cout, int Cin, int groups, int KH, int KW, int Kg,
                                 int chw_order, spb_stream_t stream) {
  if (!W || !Wp || Cout <= 0 || Cin <= 0 || groups <= 0 || (Cout % groups) || (Cin % groups) || Kg < KH * KW * (Cin / groups)) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for((long long)Cout * Kg);
  SPN_T(dtype, hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, W, (bf16_t*)Wp, (bf16_t*)WpT, Cout, Cin, groups, KH, KW, Kg, chw_order),
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(grid), dim3(256), 0, s, W, (float*)Wp, (float*)WpT, Cout, Cin, groups, KH, KW, Kg, chw_order))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_unpack_conv_grad(const float* dWp, float* dW, int Cout, int Cin, int groups, int KH, int KW, int Kg,
                                        int chw_order, spb_stream_t stream) {
  if (!dWp || !dW || Cout <= 0 || Cin <= 0 || groups <= 0 || (Cout % groups) || (Cin % groups) || Kg < KH * KW * (Cin / groups)) return SPB_E_ARG;
  hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(grid_for((long long)Cout * (Cin / groups) * KH * KW)), dim3(256), 0, (hipStream_t)stream,
                     dWp, dW, Cout, Cin, groups, KH, KW, Kg, chw_order);
  SPB_CHECK_LAUNCH();
  return 0;
}
