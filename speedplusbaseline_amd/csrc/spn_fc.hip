// SPN fully connected layers at training batch sizes (reference src/nets/spn.py:71-99, fc6..fc11; bs=32 in
// BASELINE configs[5]).  With M <= 64 rows every fc pass is a stream over the weight matrix (37.7 M / 16.8 M / 20.5 M
// elements): the three kernels below read or write each weight element exactly once and keep the skinny operand
// (activations / gradients, <= 1.2 MB) in L2.  bf16 operands, f32 accumulation on the matrix cores
// (v_mfma_f32_16x16x32_bf16).  "MP" = 16*MT is the padded row count (32 or 64).
//
//   fc_fwd     accT[n][m] += sum_k  W[n][k]  X[m][k]      A = W rows straight from HBM (each lane 2 x 16 B of one row),
//                                                         B = X rows from L2; split over k across waves and workgroups
//   fc_dgrad   accT[k][m] += sum_n  W[n][k]  G[m][n]      the reduction runs down W's slow axis: 32 x 128 tiles go through
//                                                         wave-private LDS and come back with ds_read_b64_tr_b16
//   fc_wgrad   dW[n][k]    = sum_m GT[n][m] XT[k][m]      one MFMA per 16x16 output tile (the whole batch is one k-step);
//                                                         plain 16-byte stores, no atomics, no zero fill
//   fc_epi     per feature: (bias, ReLU, inverted dropout) or (ReLU/dropout backward, bias gradient); writes the
//              row-major tensor, its transpose (operand of fc_wgrad) and re-zeroes the accumulator
// accT is a feature-major f32 accumulator [features][MP] that is zero between uses (fc_epi / unflatten clean it).
#include "common.h"
#include "optim_math.h"

namespace {

typedef s16x4_t __attribute__((address_space(3))) * lds_v4;

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) {
  union { uint4 u; bf16x8_t v; } x;
  x.u = *reinterpret_cast<const uint4*>(p);
  return x.v;
}
__device__ __forceinline__ bf16x8_t zero_frag() {
  union { uint4 u; bf16x8_t v; } x;
  x.u = make_uint4(0, 0, 0, 0);
  return x.v;
}
__device__ __forceinline__ f32x4_t mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return SPB_MFMA16(a, b, c);
}

// cross-wave sum of per-lane accumulators through LDS, then atomics into accT[f0 + ...][MP]
// c[idx] : idx = tile * MT + mt, lane (li, lq) holds m = mt*16 + li, features f0 + tile*16 + lq*4 + e
template <int NT, int MT>
__device__ __forceinline__ void reduce_store(f32x4_t (&c)[NT * MT], float* red, float* accT, int f0, int F, int M, int w, int l) {
  constexpr int MP = 16 * MT, CNT = NT * MT * 4;
  const int li = l & 15, lq = l >> 4;
#pragma unroll
  for (int i = 0; i < NT * MT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(w * CNT + i * 4 + e) * 64 + l] = c[i][e];
  __syncthreads();
  for (int j = w; j < CNT; j += 4) {
    const float v = red[j * 64 + l] + red[(CNT + j) * 64 + l] + red[(2 * CNT + j) * 64 + l] + red[(3 * CNT + j) * 64 + l];
    const int e = j & 3, i = j >> 2, mt = i % MT, tile = i / MT;
    const int f = f0 + tile * 16 + lq * 4 + e, m = mt * 16 + li;
    if (f < F && m < M) atomicAdd(accT + (size_t)f * MP + m, v);
  }
}

// ---------------------------------------------------------------------------------------------------------- forward
// grid = ceil(N/32) * KS workgroups of 4 waves; a workgroup owns 32 weight rows and the 64-wide k-steps
// [ks*per, (ks+1)*per); its waves take every 4th step.
template <int MT>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, float* __restrict__ accT,
                                                     int M, int N, int K, int KS) {
  __shared__ float red[4 * 2 * MT * 4 * 64];
  const int grp = blockIdx.x / KS, ks = blockIdx.x % KS;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int n0 = grp * 32;
  const int steps = K >> 6, per = (steps + KS - 1) / KS;
  const int sb = ks * per, se = min(steps, sb + per);
  const bf16_t* wr[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) wr[rt] = W + (size_t)min(n0 + rt * 16 + li, N - 1) * K + lq * 16;
  const bf16_t* xr[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    xv[mt] = mt * 16 + li < M;
    xr[mt] = X + (size_t)min(mt * 16 + li, M - 1) * K + lq * 16;
  }
  f32x4_t c[2 * MT];
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = sb + w; s < se; s += 4) {
    const int kb = s << 6;
    bf16x8_t a[2][2], b[MT][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { a[rt][0] = ld_frag(wr[rt] + kb); a[rt][1] = ld_frag(wr[rt] + kb + 8); }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      b[mt][0] = xv[mt] ? ld_frag(xr[mt] + kb) : zero_frag();
      b[mt][1] = xv[mt] ? ld_frag(xr[mt] + kb + 8) : zero_frag();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        c[rt * MT + mt] = mfma(a[rt][0], b[mt][0], c[rt * MT + mt]);
        c[rt * MT + mt] = mfma(a[rt][1], b[mt][1], c[rt * MT + mt]);
      }
  }
  reduce_store<2, MT>(c, red, accT, n0, N, M, w, l);
}

// ------------------------------------------------------------------------------------------------- input gradient
// grid = (K/128) * NS workgroups of 4 waves; a workgroup owns 128 input features (columns of W) and the 32-row
// n-steps [ns*per, (ns+1)*per); its waves take every 4th step, each through its own LDS tile.
constexpr int DG_LD = 128 + 8;
template <int MT>
__global__ __launch_bounds__(256) void fc_dgrad_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ W, float* __restrict__ accT,
                                                       int M, int N, int K, int NS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem) + (threadIdx.x >> 6) * 32 * DG_LD;
  float* red = reinterpret_cast<float*>(smem);   // reused after the main loop (needs 4*8*MT*4*64*4 bytes)
  const int kt = blockIdx.x / NS, ns = blockIdx.x % NS;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int k0 = kt * 128;
  const int steps = (N + 31) >> 5, per = (steps + NS - 1) / NS;
  const int sb = ns * per, se = min(steps, sb + per);
  const int lr = l >> 4, lc = (l & 15) * 8;          // staging: row lr + 4*i, 8 columns at lc
  f32x4_t c[8 * MT];
#pragma unroll
  for (int i = 0; i < 8 * MT; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 st[8];
  auto fetch = [&](int s) {
    const int nb = s << 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = nb + lr + 4 * i;
      st[i] = n < N ? *reinterpret_cast<const uint4*>(W + (size_t)n * K + k0 + lc) : make_uint4(0, 0, 0, 0);
    }
  };
  int s = sb + w;
  if (s < se) fetch(s);
  for (; s < se; s += 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(tile + (lr + 4 * i) * DG_LD + lc) = st[i];
    asm volatile("" ::: "memory");   // tile stores stay ahead of the transpose loads
    const int nb = s << 5;
    bf16x8_t b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + li, n = nb + lq * 8;
      b[mt] = (m < M && n < N) ? ld_frag(G + (size_t)m * N + n) : zero_frag();
    }
    if (s + 4 < se) fetch(s + 4);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const bf16_t* p = tile + (lq * 8 + (li >> 2)) * DG_LD + it * 16 + (li & 3) * 4;
      union { struct { s16x4_t lo, hi; } h; bf16x8_t v; } a;
      a.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
      a.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * DG_LD));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) c[it * MT + mt] = mfma(a.v, b[mt], c[it * MT + mt]);
    }
    asm volatile("" ::: "memory");   // ... and the next tile's stores behind them
  }
  __syncthreads();   // all waves done with their tiles before the region is reused for the reduction
  reduce_store<8, MT>(c, red, accT, k0, K, M, w, l);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// grid = ceil(N/64) * ceil(K/(64*KW)) workgroups of 4 waves; a workgroup owns 64 output rows n; wave w writes the
// 64-wide column blocks kb = (blk*4 + w)*KPW .. +KPW.
// FUSED: the layer's share of the optimizer step in the epilogue (spb_fc_wgrad_update): every dW element is produced by exactly
// one lane (the batch is the whole reduction), so clip_grad_value_ + update + bf16 shadow are applied to it in registers --
// the 600 MB of float32 weight gradients are neither written nor read back (26 instead of 34 bytes of HBM traffic per
// parameter for gradient + AdamW).  opt.params / m / v / shadow_bf16 point at this layer's [N][K] weight; opt.grads, when
// given, still receives the raw gradient.
template <int MT, bool FUSED = false>
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const bf16_t* __restrict__ GT, const bf16_t* __restrict__ XT, float* __restrict__ dW,
                                                       int N, int K, int KPW, const spb_optim_args_t opt = spb_optim_args_t{}) {
  float gs = 1.f, lr = 0.f, bc1 = 1.f, bc2 = 1.f;
  bool has_m = false, has_v = false;
  if constexpr (FUSED) {
    gs = opt.gmul ? *opt.gmul : 1.f;
    lr = opt.hyper ? opt.hyper[0] : opt.lr;
    bc1 = opt.hyper ? opt.hyper[1] : opt.bias_c1;
    bc2 = opt.hyper ? opt.hyper[2] : opt.bias_c2;
    has_m = opt.m && (opt.kind >= 2 || (opt.kind == 0 && opt.beta1 != 0.f));
    has_v = opt.v && opt.kind >= 1;
  }
  constexpr int MP = 16 * MT;
  const int kblocks = (K + 63) >> 6;
  const int kgroups = (kblocks + 4 * KPW - 1) / (4 * KPW);
  const int ng = blockIdx.x / kgroups, kg = blockIdx.x % kgroups;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int n0 = ng * 64;
  bf16x8_t b[4][MT / 2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int h = 0; h < MT; h += 2) {   // MP = 32: one 32-deep step; MP = 64: two
      const int n = min(n0 + nt * 16 + li, N - 1);
      b[nt][h / 2] = ld_frag(GT + (size_t)n * MP + h * 16 + lq * 8);
    }
  const int kb0 = (kg * 4 + w) * KPW;
  for (int kb = kb0; kb < min(kblocks, kb0 + KPW); ++kb) {
    const int k0 = kb << 6;
    bf16x8_t a[4][MT / 2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int h = 0; h < MT / 2; ++h) {
        const int k = min(k0 + tt * 16 + li, K - 1);
        a[tt][h] = ld_frag(XT + (size_t)k * MP + h * 32 + lq * 8);
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + nt * 16 + li;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        f32x4_t c = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < MT / 2; ++h) c = mfma(a[tt][h], b[nt][h], c);
        // lane: column j = n (li), rows i = lq*4+e <-> k = k0 + tt*16 + lq*4 + e : 16 contiguous bytes
        const int k = k0 + tt * 16 + lq * 4;
        if constexpr (FUSED) {
          if (n < N && k + 3 < K) {       // K % 4 == 0 (checked by the launcher): whole vectors only
            const size_t o = (size_t)n * K + k;
            f32x4_t p = *reinterpret_cast<const f32x4_t*>(opt.params + o);
            f32x4_t m = has_m ? *reinterpret_cast<const f32x4_t*>(opt.m + o) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            f32x4_t v = has_v ? *reinterpret_cast<const f32x4_t*>(opt.v + o) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float me = m[e], ve = v[e];
              p[e] = optim_one(opt, gs, lr, bc1, bc2, p[e], c[e], me, ve);
              m[e] = me; v[e] = ve;
            }
            if (has_m) *reinterpret_cast<f32x4_t*>(opt.m + o) = m;
            if (has_v) *reinterpret_cast<f32x4_t*>(opt.v + o) = v;
            *reinterpret_cast<f32x4_t*>(opt.params + o) = p;
            if (opt.shadow_bf16)
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(opt.shadow_bf16) + o) = make_uint2(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]));
            if (dW) *reinterpret_cast<f32x4_t*>(dW + o) = c;
          }
        } else {
        if (n < N && k + 3 < K) *reinterpret_cast<f32x4_t*>(dW + (size_t)n * K + k) = c;
        else if (n < N) for (int e = 0; e < 4; ++e) if (k + e < K) dW[(size_t)n * K + k + e] = c[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------- epilogues
__device__ __forceinline__ unsigned hash32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}

// A thread owns 4 batch rows of one feature (block = 256/MG features x MG row groups, MG = MP/4).
// mode 0 (forward): v = acc + bias; ReLU if relu; inverted dropout with probability p if p > 0 (mask [M][F] written, or read
//   when given); Y [M][F], YT [F][MP].
// mode 1 (backward): v = (acc or src) * scale where H > 0 (H NULL: everywhere); Y, YT as above; db[f] = sum_m v.
template <int MT>
__global__ __launch_bounds__(256) void fc_epi_kernel(float* __restrict__ accT, const bf16_t* __restrict__ src, const float* __restrict__ bias,
                              const bf16_t* __restrict__ H, bf16_t* __restrict__ Y, bf16_t* __restrict__ YT, unsigned char* __restrict__ mask,
                              float* __restrict__ db, int M, int F, int mode, int relu, float p, float scale, unsigned long long seed, int given) {
  constexpr int MP = 16 * MT, MG = MP / 4, FPB = 256 / MG;
  __shared__ float red[256];
  const int fl = threadIdx.x % FPB, mg = threadIdx.x / FPB;
  const int f = blockIdx.x * FPB + fl;
  const bool fok = f < F;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (fok) {
    if (accT) {
      float4* a4 = reinterpret_cast<float4*>(accT + (size_t)f * MP + mg * 4);
      const float4 q = *a4;
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      *a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int m = mg * 4 + e; v[e] = m < M ? bf2f(src[(size_t)m * F + f]) : 0.f; }
    }
  }
  float s = 0.f;
  const float bv = (fok && mode == 0 && bias) ? bias[f] : 0.f;
  const float ds = p > 0.f ? 1.f / (1.f - p) : 1.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = mg * 4 + e;
    float x = v[e];
    if (m >= M || !fok) x = 0.f;
    else if (mode == 0) {
      x += bv;
      if (relu) x = fmaxf(x, 0.f);
      if (p > 0.f) {
        x = bf2f(f2bf(x));   // the activation is rounded to the storage type before dropout scales it (as the separate kernels did)
        const size_t i = (size_t)m * F + f;
        unsigned char keep;
        if (given) keep = mask[i];
        else { keep = (hash32(seed * 0x100000001B3ull + (unsigned long long)i) * (1.0f / 4294967296.0f)) >= p ? 1 : 0; mask[i] = keep; }
        x = keep ? x * ds : 0.f;
      }
    } else {
      x = (!H || bf2f(H[(size_t)m * F + f]) > 0.f) ? x * scale : 0.f;
      x = bf2f(f2bf(x));
      s += x;
    }
    v[e] = x;
  }
  if (fok) {
    if (Y)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (mg * 4 + e < M) Y[(size_t)(mg * 4 + e) * F + f] = f2bf(v[e]);
    if (YT) *reinterpret_cast<uint2*>(YT + (size_t)f * MP + mg * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
  if (mode == 1 && db) {      // bias gradient: sum over the MG row groups of a feature
    red[threadIdx.x] = s;
    __syncthreads();
    if (mg == 0 && fok) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < MG; ++q) a += red[q * FPB + fl];
      db[f] = a;
    }
  }
}

// pool5 output NHWC [B][HW][C] -> f [B][C*HW] in the reference's NCHW flatten order (spn.py:131 x.view(-1, 9216)) + fT
template <int MT>
__global__ void flatten_kernel(const bf16_t* __restrict__ P, bf16_t* __restrict__ Fm, bf16_t* __restrict__ FT, int B, int HW, int Cn) {
  constexpr int MP = 16 * MT;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;    // f = c*HW + hw
  if (f >= HW * Cn) return;
  const int c = f / HW, hw = f % HW;
  bf16_t v[MP];
#pragma unroll
  for (int m = 0; m < MP; ++m) v[m] = m < B ? P[((size_t)m * HW + hw) * Cn + c] : (bf16_t)0;
#pragma unroll
  for (int m = 0; m < MP; ++m) if (m < B) Fm[(size_t)m * HW * Cn + f] = v[m];
  uint4* o = reinterpret_cast<uint4*>(FT + (size_t)f * MP);
#pragma unroll
  for (int i = 0; i < MP / 8; ++i)
    o[i] = make_uint4(v[8 * i] | ((unsigned)v[8 * i + 1] << 16), v[8 * i + 2] | ((unsigned)v[8 * i + 3] << 16),
                      v[8 * i + 4] | ((unsigned)v[8 * i + 5] << 16), v[8 * i + 6] | ((unsigned)v[8 * i + 7] << 16));
}
// accT [C*HW][MP] (NCHW feature order) -> gradient of pool5's output, NHWC bf16; accT re-zeroed
template <int MT>
__global__ void unflatten_kernel(float* __restrict__ accT, bf16_t* __restrict__ Gp, int B, int HW, int Cn) {
  constexpr int MP = 16 * MT;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;    // i = hw*C + c : coalesced NHWC writes
  if (i >= HW * Cn) return;
  const int hw = i / Cn, c = i % Cn;
  float* a = accT + (size_t)(c * HW + hw) * MP;
#pragma unroll
  for (int m = 0; m < MP; ++m) {
    const float x = a[m];
    a[m] = 0.f;
    if (m < B) Gp[((size_t)m * HW + hw) * Cn + c] = f2bf(x);
  }
}

}  // namespace

#define FC_MT(M, CALL2, CALL4) \
  if ((M) <= 32) { CALL2; } else { CALL4; }

extern "C" int spb_fc_fwd(const void* X, const void* W, float* accT, int M, int N, int K, spb_stream_t stream) {
  if (!X || !W || !accT || M <= 0 || N <= 0 || K <= 0) return SPB_E_ARG;
  if (M > 64 || (K & 63)) return SPB_E_UNSUPPORTED;
  const int groups = (N + 31) / 32, steps = K >> 6;
  int KS = 1024 / groups; if (KS > steps / 8) KS = steps / 8; if (KS < 1) KS = 1;
  hipStream_t s = (hipStream_t)stream;
  FC_MT(M, hipLaunchKernelGGL(fc_fwd_kernel<2>, dim3(groups * KS), dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)W, accT, M, N, K, KS),
        hipLaunchKernelGGL(fc_fwd_kernel<4>, dim3(groups * KS), dim3(256), 0, s, (const bf16_t*)X, (const bf16_t*)W, accT, M, N, K, KS))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_fc_dgrad(const void* G, const void* W, float* accT, int M, int N, int K, spb_stream_t stream) {
  if (!G || !W || !accT || M <= 0 || N <= 0 || K <= 0) return SPB_E_ARG;
  if (M > 64 || (K & 127) || (N & 7)) return SPB_E_UNSUPPORTED;
  const int kt = K / 128, steps = (N + 31) / 32;
  int NS = 1024 / kt; if (NS > steps / 8) NS = steps / 8; if (NS < 1) NS = 1;
  hipStream_t s = (hipStream_t)stream;
  const int MT = M <= 32 ? 2 : 4;
  size_t lds = 4 * 32 * DG_LD * sizeof(bf16_t);
  const size_t redb = (size_t)4 * 8 * MT * 4 * 64 * sizeof(float);
  if (redb > lds) lds = redb;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)fc_dgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)fc_dgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  FC_MT(M, hipLaunchKernelGGL(fc_dgrad_kernel<2>, dim3(kt * NS), dim3(256), lds, s, (const bf16_t*)G, (const bf16_t*)W, accT, M, N, K, NS),
        hipLaunchKernelGGL(fc_dgrad_kernel<4>, dim3(kt * NS), dim3(256), lds, s, (const bf16_t*)G, (const bf16_t*)W, accT, M, N, K, NS))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_fc_wgrad(const void* GT, const void* XT, float* dW, int M, int N, int K, spb_stream_t stream) {
  if (!GT || !XT || !dW || M <= 0 || N <= 0 || K <= 0) return SPB_E_ARG;
  if (M > 64 || (K & 3)) return SPB_E_UNSUPPORTED;
  const int kblocks = (K + 63) / 64, KPW = 2;
  const int kgroups = (kblocks + 4 * KPW - 1) / (4 * KPW), ngroups = (N + 63) / 64;
  hipStream_t s = (hipStream_t)stream;
  FC_MT(M, hipLaunchKernelGGL(fc_wgrad_kernel<2>, dim3(ngroups * kgroups), dim3(256), 0, s, (const bf16_t*)GT, (const bf16_t*)XT, dW, N, K, KPW),
        hipLaunchKernelGGL(fc_wgrad_kernel<4>, dim3(ngroups * kgroups), dim3(256), 0, s, (const bf16_t*)GT, (const bf16_t*)XT, dW, N, K, KPW))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_fc_wgrad_update(const void* GT, const void* XT, int M, int N, int K, const spb_optim_args_t* opt, spb_stream_t stream) {
  if (!GT || !XT || !opt || !opt->params || M <= 0 || N <= 0 || K <= 0) return SPB_E_ARG;
  if (opt->n != (long long)N * K || opt->kind < 0 || opt->kind > 3) return SPB_E_ARG;
  if (opt->kind >= 2 && (!opt->m || !opt->v)) return SPB_E_ARG;
  if (opt->kind == 1 && !opt->v) return SPB_E_ARG;
  if (opt->max_norm > 0.f) return SPB_E_UNSUPPORTED;      // a global-norm clip needs every gradient first
  if (M > 64 || (K & 3)) return SPB_E_UNSUPPORTED;
  if (((uintptr_t)opt->params | (uintptr_t)opt->grads | (uintptr_t)opt->m | (uintptr_t)opt->v) & 15 || ((uintptr_t)opt->shadow_bf16 & 7)) return SPB_E_ARG;
  const int kblocks = (K + 63) / 64, KPW = 2;
  const int kgroups = (kblocks + 4 * KPW - 1) / (4 * KPW), ngroups = (N + 63) / 64;
  hipStream_t s = (hipStream_t)stream;
  FC_MT(M, hipLaunchKernelGGL((fc_wgrad_kernel<2, true>), dim3(ngroups * kgroups), dim3(256), 0, s, (const bf16_t*)GT, (const bf16_t*)XT, opt->grads, N, K, KPW, *opt),
        hipLaunchKernelGGL((fc_wgrad_kernel<4, true>), dim3(ngroups * kgroups), dim3(256), 0, s, (const bf16_t*)GT, (const bf16_t*)XT, opt->grads, N, K, KPW, *opt))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_fc_epilogue(const spb_fc_epi_args_t* a, spb_stream_t stream) {
  if (!a || (!a->accT && !a->src) || a->M <= 0 || a->F <= 0) return SPB_E_ARG;
  if (a->M > 64) return SPB_E_UNSUPPORTED;
  if (a->mode == 0 && a->p > 0.f && !a->mask) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int fpb = a->M <= 32 ? 32 : 16;
  const dim3 grid((a->F + fpb - 1) / fpb), blk(256);
  FC_MT(a->M,
        hipLaunchKernelGGL(fc_epi_kernel<2>, grid, blk, 0, s, a->accT, (const bf16_t*)a->src, a->bias, (const bf16_t*)a->H, (bf16_t*)a->Y,
                           (bf16_t*)a->YT, a->mask, a->db, a->M, a->F, a->mode, a->relu, a->p, a->scale, a->seed, a->mask_given),
        hipLaunchKernelGGL(fc_epi_kernel<4>, grid, blk, 0, s, a->accT, (const bf16_t*)a->src, a->bias, (const bf16_t*)a->H, (bf16_t*)a->Y,
                           (bf16_t*)a->YT, a->mask, a->db, a->M, a->F, a->mode, a->relu, a->p, a->scale, a->seed, a->mask_given))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_flatten(const void* P, void* Fm, void* FT, int B, int HW, int C, spb_stream_t stream) {
  if (!P || !Fm || !FT || B <= 0 || HW <= 0 || C <= 0) return SPB_E_ARG;
  if (B > 64) return SPB_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((HW * C + 63) / 64), blk(64);
  FC_MT(B, hipLaunchKernelGGL(flatten_kernel<2>, grid, blk, 0, s, (const bf16_t*)P, (bf16_t*)Fm, (bf16_t*)FT, B, HW, C),
        hipLaunchKernelGGL(flatten_kernel<4>, grid, blk, 0, s, (const bf16_t*)P, (bf16_t*)Fm, (bf16_t*)FT, B, HW, C))
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_unflatten_grad(float* accT, void* Gp, int B, int HW, int C, spb_stream_t stream) {
  if (!accT || !Gp || B <= 0 || HW <= 0 || C <= 0) return SPB_E_ARG;
  if (B > 64) return SPB_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((HW * C + 255) / 256), blk(256);
  FC_MT(B, hipLaunchKernelGGL(unflatten_kernel<2>, grid, blk, 0, s, accT, (bf16_t*)Gp, B, HW, C),
        hipLaunchKernelGGL(unflatten_kernel<4>, grid, blk, 0, s, accT, (bf16_t*)Gp, B, HW, C))
  SPB_CHECK_LAUNCH();
  return 0;
}
