// Pointwise (1x1) convolution GEMMs of the 14x14 / 7x7 maps with a long reduction: split-K over the waves of a workgroup.
//
//   forward   Y[M,N]  = act(bn(A))[M,K] * W[N,K]^T        + sum(y), sum(y^2)            (project convs: K = 384..1280)
//   dgrad     dA[M,N] = bn_bwd(G,Z)[M,K] * Wt[N,K]^T      + mask, sum(g), sum(g*xhat)    (expand convs:  K = 384..1280)
// Same contract as spb_pwconv_gemm (gemm_pw.hip): reference park2019.py:51-53,64-66 and the torchvision MobileNetV2
// expand / project convolutions (park2019.py:107-108).
//
// Why another GEMM.  At bs=48 these layers have M = 9408 or 2352 rows and N = 64..320 columns: the tiled kernel
// (64-row x 64-column tiles, the four waves side by side along M) launches 111..294 workgroups on 256 CUs -- 444..1176
// waves on 1024 SIMDs -- and each of them walks the whole reduction in 12..40 barrier-separated stages.  rocprofv3 SQ
// counters of those launches: waves parked 60 % of their cycles, VALU (the BatchNorm transform of the A tile) 20 %, and
// more than half of the SIMDs never see a wave.  Here
//   * a workgroup owns a SMALL output tile (16*RF rows x 16*NF columns) and its four waves split the REDUCTION: wave w
//     takes the 32-deep chunks w, w+4, w+8, ... (adjacent waves read adjacent 64-byte pieces of the same rows);
//   * operands go global -> registers directly in the matrix-core layout (lane (i, q): row i, k = 8q..8q+7 = one 16-byte
//     load), D chunks ahead, no LDS staging and NO barrier in the reduction loop;
//   * the BatchNorm(+activation) / BatchNorm-backward transform runs on the fragment in registers (coefficients from an
//     LDS table, broadcast reads);
//   * the four partial tiles meet once, in LDS, and the epilogue (activation mask, residual gradient, batch sums, 8- or
//     16-byte stores) runs on the summed tile.
// 2352 rows x 960 -> 160 (input gradient of an expand convolution at 7x7): 441 workgroups x 4 waves, 7-8 chunks each.
#include "common.h"
#include <hip/hip_ext.h>

namespace {

constexpr int SKK = 32;   // reduction depth of one MFMA 16x16x32 step

template <int RF, int NF> struct SkShape {
  static constexpr int BM = 16 * RF, BN = 16 * NF;
  static constexpr int LDR = BN + 4;                        // f32 row stride of a partial tile in LDS
  static constexpr int VW = (BM * BN / 256) >= 8 ? 8 : 4;   // consecutive output channels per thread in the epilogue
  static constexpr int NVT = BN / VW;                       // vector columns per tile row
  static constexpr int ROWS = 256 / NVT;                    // tile rows per epilogue sweep
  static constexpr int SWEEPS = (BM + ROWS - 1) / ROWS;
};

template <int RF, int NF>
constexpr size_t sk_tile_bytes() {
  size_t a = (size_t)4 * SkShape<RF, NF>::BM * SkShape<RF, NF>::LDR * sizeof(float);
  size_t b = (size_t)2 * SkShape<RF, NF>::ROWS * SkShape<RF, NF>::BN * sizeof(float);
  return a > b ? a : b;
}

template <int VW> struct RawV;
template <> struct RawV<8> { uint4 u; };
template <> struct RawV<4> { uint2 u; };
template <int VW> __device__ __forceinline__ RawV<VW> ldv(const bf16_t* p);
template <> __device__ __forceinline__ RawV<8> ldv<8>(const bf16_t* p) { RawV<8> r; r.u = *reinterpret_cast<const uint4*>(p); return r; }
template <> __device__ __forceinline__ RawV<4> ldv<4>(const bf16_t* p) { RawV<4> r; r.u = *reinterpret_cast<const uint2*>(p); return r; }
__device__ __forceinline__ void cvtv(const RawV<8>& r, float* v) {
  spb_unpack2(r.u.x, v[0], v[1]); spb_unpack2(r.u.y, v[2], v[3]);      // (bf16 or, in the -DSPB_F16 twin, IEEE half: common.h)
  spb_unpack2(r.u.z, v[4], v[5]); spb_unpack2(r.u.w, v[6], v[7]);
}
__device__ __forceinline__ void cvtv(const RawV<4>& r, float* v) {
  spb_unpack2(r.u.x, v[0], v[1]); spb_unpack2(r.u.y, v[2], v[3]);
}
template <int VW> __device__ __forceinline__ void stv(bf16_t* p, const float* v);
template <> __device__ __forceinline__ void stv<8>(bf16_t* p, const float* v) { st8<bf16_t>(p, v); }
template <> __device__ __forceinline__ void stv<4>(bf16_t* p, const float* v) {
  uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

// PRO 1: a = act(bn(A));  PRO 2: a = bn_backward(g = A, z = A2).   EPI 0: y = acc * out_scale;  1: y = acc + batch sums;
// 2: g = (acc + res) * act'(bn(Zout)) + sum(g), sum(g * xhat).   D = chunks in flight per wave.
template <int PRO, int EPI, int RF, int NF, int D>
__global__ __launch_bounds__(256) void pw_sk_kernel(const spb_gemm_args_t g) {
  typedef SkShape<RF, NF> S;
  constexpr int BM = S::BM, BN = S::BN, LDR = S::LDR, VW = S::VW, NVT = S::NVT, ROWS = S::ROWS, SWEEPS = S::SWEEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = g.M, K = g.K, N = g.N;
  const int lda = g.lda > 0 ? g.lda : K, ldc = g.ldc > 0 ? g.ldc : N;
  const int Kp = (K + SKK - 1) / SKK * SKK, KT = Kp / SKK;
  float* coef = reinterpret_cast<float*>(smem);                 // [3][Kp]
  float* ecoef = coef + 3 * Kp;                                  // [4][BN] scale, shift, mean, inverse std
  float* red = ecoef + 4 * BN;                                   // [4][BM][LDR] partial tiles, then the statistics scratch

  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int NT = (N + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (lid % NT) * BN, m0 = (lid / NT) * BM;

  const bf16_t* Ag = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* A2g = (PRO == 2 && g.A2) ? reinterpret_cast<const bf16_t*>(g.A2) : Ag;   // identity prologue: p1 == 0
  const bf16_t* Bg = reinterpret_cast<const bf16_t*>(g.Bw);
  bf16_t* Yg = reinterpret_cast<bf16_t*>(g.Y);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(g.res);
  const bf16_t* Zg = reinterpret_cast<const bf16_t*>(g.Zout);

  // ---- operand rows of this lane (clamped: rows past the end repeat the last one and are never stored)
  size_t arow[RF], brow[NF];
#pragma unroll
  for (int i = 0; i < RF; ++i) { const int m = m0 + i * 16 + li; arow[i] = (size_t)(m < M ? m : M - 1) * lda; }
#pragma unroll
  for (int j = 0; j < NF; ++j) { const int n = n0 + j * 16 + li; brow[j] = (size_t)(n < N ? n : N - 1) * K; }

  uint4 ra[D][RF], ra2[D][PRO == 2 ? RF : 1], rb[D][NF];
#define SK_LOAD(SLOT, s_)                                                                       \
  {                                                                                             \
    const int k_ = (s_) * SKK + lq * 8;                                                         \
    const int kc_ = k_ < K ? k_ : K - 8;                                                        \
    _Pragma("unroll") for (int i = 0; i < RF; ++i) {                                            \
      ra[SLOT][i] = *reinterpret_cast<const uint4*>(Ag + arow[i] + kc_);                        \
      if (PRO == 2) ra2[SLOT][PRO == 2 ? i : 0] = *reinterpret_cast<const uint4*>(A2g + arow[i] + kc_); \
    }                                                                                           \
    _Pragma("unroll") for (int j = 0; j < NF; ++j) rb[SLOT][j] = *reinterpret_cast<const uint4*>(Bg + brow[j] + kc_); \
  }
  // the first D chunks of this wave are in flight while the coefficient table is built
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int s = w + 4 * d;
    SK_LOAD(d, s < KT ? s : KT - 1);
  }
  // output-side operands of the epilogue (dgrad): issued now, consumed after the reduction
  const int vcol = t % NVT, vrow0 = t / NVT;
  const int nE = n0 + vcol * VW;
  const bool colok = nE < N;
  RawV<VW> zr[EPI == 2 ? SWEEPS : 1], rr[EPI == 2 ? SWEEPS : 1];
  if (EPI == 2) {
#pragma unroll
    for (int sw = 0; sw < SWEEPS; ++sw) {
      const int m = m0 + vrow0 + sw * ROWS;
      const size_t o = (size_t)(m < M ? m : M - 1) * ldc + (colok ? nE : 0);
      zr[EPI == 2 ? sw : 0] = ldv<VW>(Zg + o);
      if (Rg) rr[EPI == 2 ? sw : 0] = ldv<VW>(Rg + o);
    }
  }

  // ---- prologue coefficients for every reduction channel, from the producer's raw batch sums
  BNEpiPre epre;
  if (EPI == 2) bn_epi_issue(g.epi, n0, N, BN, t, epre);
  bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);
  if (EPI == 2) bn_epi_finish<true>(g.epi, n0, N, BN, BN, ecoef, t, epre);
  __syncthreads();

  f32x4_t acc[RF][NF];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);

  // ---- reduction: chunk s = w + 4*it; slot it % D holds it, and is refilled with chunk s + 4*D right after use
  for (int s0 = w; s0 < KT; s0 += 4 * D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int s = s0 + 4 * d;
      if (s < KT) {
        const int kb = s * SKK + lq * 8;
        const float4 c0a = *reinterpret_cast<const float4*>(coef + kb), c0b = *reinterpret_cast<const float4*>(coef + kb + 4);
        const float4 c1a = *reinterpret_cast<const float4*>(coef + Kp + kb), c1b = *reinterpret_cast<const float4*>(coef + Kp + kb + 4);
        const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
        const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
        float c2[8];
        if (PRO == 2) {
          const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb), c2b = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb + 4);
          c2[0] = c2a.x; c2[1] = c2a.y; c2[2] = c2a.z; c2[3] = c2a.w; c2[4] = c2b.x; c2[5] = c2b.y; c2[6] = c2b.z; c2[7] = c2b.w;
        }
        bf16x8_t af[RF];
#pragma unroll
        for (int i = 0; i < RF; ++i) {
          Raw8<bf16_t> r1; r1.u = ra[d][i];
          float a[8], x[8];
          cvt8(r1, a);
          if (PRO == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = a[j] * c0[j] + c1[j];
              x[j] = __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f);
            }
          } else {
            Raw8<bf16_t> r2; r2.u = ra2[d][PRO == 2 ? i : 0];
            float a2[8];
            cvt8(r2, a2);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = a[j] * c0[j] + a2[j] * c1[j] + c2[j];
          }
          uint4 pa;
          pa.x = pack_bf16x2(x[0], x[1]); pa.y = pack_bf16x2(x[2], x[3]); pa.z = pack_bf16x2(x[4], x[5]); pa.w = pack_bf16x2(x[6], x[7]);
          if (kb >= K) pa = make_uint4(0, 0, 0, 0);     // reduction padding: clamped (finite) weights times an explicit zero
          af[i] = __builtin_bit_cast(bf16x8_t, pa);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const bf16x8_t bfv = __builtin_bit_cast(bf16x8_t, rb[d][j]);
#pragma unroll
          for (int i = 0; i < RF; ++i) acc[i][j] = SPB_MFMA16(af[i], bfv, acc[i][j]);
        }
        const int sn = s + 4 * D;
        SK_LOAD(d, sn < KT ? sn : KT - 1);              // clamped, unconditional: no load inside a branch of its own
      }
    }
  }
#undef SK_LOAD

  // ---- the four partial tiles meet in LDS (C layout: column = lane & 15, row = (lane >> 4) * 4 + r)
  float* mine = red + (size_t)w * BM * LDR;
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(i * 16 + lq * 4 + r) * LDR + j * 16 + li] = acc[i][j][r];
  __syncthreads();

  float s1[VW], s2[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int sw = 0; sw < SWEEPS; ++sw) {
    const int r = vrow0 + sw * ROWS;
    const int m = m0 + r;
    if (r < BM && m < M && colok) {
      float v[VW];
#pragma unroll
      for (int j = 0; j < VW; j += 4) {
        float4 p = *reinterpret_cast<const float4*>(red + r * LDR + vcol * VW + j);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          const float4 q = *reinterpret_cast<const float4*>(red + (size_t)ww * BM * LDR + r * LDR + vcol * VW + j);
          p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w;
        }
        v[j] = p.x; v[j + 1] = p.y; v[j + 2] = p.z; v[j + 3] = p.w;
      }
      const size_t o = (size_t)m * ldc + nE;
      if (EPI == 0) {
#pragma unroll
        for (int j = 0; j < VW; ++j) v[j] *= g.out_scale;
        stv<VW>(Yg + o, v);
      } else if (EPI == 1) {
#pragma unroll
        for (int j = 0; j < VW; ++j) { v[j] = rnd<bf16_t>(v[j]); s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        stv<VW>(Yg + o, v);
      } else {
        float z[VW], rv[VW];
        cvtv(zr[EPI == 2 ? sw : 0], z);
        if (Rg) {
          cvtv(rr[EPI == 2 ? sw : 0], rv);
#pragma unroll
          for (int j = 0; j < VW; ++j) v[j] += rv[j];
        }
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const float u = z[j] * ecoef[vcol * VW + j] + ecoef[BN + vcol * VW + j];
          v[j] = rnd<bf16_t>(v[j] * act_grad(u, g.epi.act, g.epi.slope));
          s1[j] += v[j];
          s2[j] += v[j] * z[j];
        }
        stv<VW>(Yg + o, v);
      }
    }
  }

  // ---- per-channel batch sums: rows of the tile in LDS, one atomic per channel and workgroup
  if (EPI != 0) {
    __syncthreads();                          // everyone is done reading the partial tiles
    float* Rs = red;                          // [2][ROWS][BN]
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      Rs[vrow0 * BN + vcol * VW + j] = s1[j];
      Rs[ROWS * BN + vrow0 * BN + vcol * VW + j] = s2[j];
    }
    __syncthreads();
    if (t < 2 * BN) {
      const int which = t / BN, c = t % BN;
      float s = 0.f;
      for (int r = 0; r < ROWS; ++r) s += Rs[which * ROWS * BN + r * BN + c];
      if (n0 + c < N) {
        if (EPI == 2 && which == 1) {          // sum g*z -> sum g*xhat = invstd * (sum g*z - mean * sum g)
          float sg = 0.f;
          for (int r = 0; r < ROWS; ++r) sg += Rs[r * BN + c];
          const float mu = ecoef[2 * BN + c], is = ecoef[3 * BN + c];   // kept from the prologue
          s = is * (s - mu * sg);
        }
        atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + (size_t)which * N + n0 + c, s);
      }
    }
  }
}

template <int PRO, int EPI, int RF, int NF, int D>
int launch_sk(const spb_gemm_args_t& g, hipStream_t stream) {
  typedef SkShape<RF, NF> S;
  const int NT = (g.N + S::BN - 1) / S::BN, MT = (g.M + S::BM - 1) / S::BM;
  const int Kp = (g.K + SKK - 1) / SKK * SKK;
  const size_t lds = (size_t)(3 * Kp + 4 * S::BN) * sizeof(float) + sk_tile_bytes<RF, NF>();
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_sk_kernel<PRO, EPI, RF, NF, D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_done = true;
  }
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_sk_kernel<PRO, EPI, RF, NF, D>), dim3(NT * MT), dim3(256), (unsigned)lds, stream, nullptr,
                          (hipEvent_t)g.stop_event, 0, g);
  else
    hipLaunchKernelGGL((pw_sk_kernel<PRO, EPI, RF, NF, D>), dim3(NT * MT), dim3(256), lds, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

int g_sk_on = 1;          // spb_debug_set_gemm_sk(0): every small-M GEMM back on the tiled kernel
int g_sk_min_k = 192;
int g_sk_max_m = 4096;     // the 7x7 maps at bs=48 (2352 rows): 19 vs 24 us (input gradient, K=960), 16 vs 20 us (forward); at 9408
                          // rows the tiled kernel already has 147+ row tiles and is as fast (13.6 vs 13.2 us) -- measured, round 2
int g_sk_max_n = 320;      // wider outputs re-read the weights per 16-row tile: the tiled kernel keeps them
int g_sk_rf = 0;          // > 0: force the row-fragment count (experiments)
int g_sk_min_wgs = 200;   // dispatch_sk: fewest workgroups a larger row tile may leave

template <int PRO, int EPI>
int dispatch_sk(const spb_gemm_args_t& g, hipStream_t stream) {
  const int NT = (g.N + 63) / 64;
  // rows per workgroup: the largest 16 * rf that still gives ~200 workgroups.  Round 5: the old rule (>= 512 workgroups) put every 7x7 layer on
  // 16-row tiles -- 441 / 735 workgroups that each pull their whole [64 x K] weight slab from L2 (96 MB of weight reads for the 1024 -> 320
  // input gradient, 48 us under load); 32-row tiles halve that: KRN step 2.62 -> 2.57 ms (rf 1 | 2 | 4 everywhere: 2.622 | 2.573 | 2.602)
  int rf = 1;
  for (int cand = 4; cand >= 2; cand >>= 1)
    if ((long long)((g.M + 16 * cand - 1) / (16 * cand)) * NT >= g_sk_min_wgs) { rf = cand; break; }
  if (g_sk_rf > 0) rf = g_sk_rf;
  if (rf >= 4) return launch_sk<PRO, EPI, 4, 4, 2>(g, stream);
  if (rf == 2) return launch_sk<PRO, EPI, 2, 4, 2>(g, stream);
  return launch_sk<PRO, EPI, 1, 4, 3>(g, stream);
}

}  // namespace

// bf16 only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the tiled kernel
int spb_gemm_sk(const spb_gemm_args_t* a, hipStream_t stream) {
  if (!g_sk_on || a->K < g_sk_min_k || a->M > g_sk_max_m || a->N < 64 || a->N > g_sk_max_n || (a->K & 7) || (a->N & 7)) return SPB_E_UNSUPPORTED;
  if (a->pro_mode == 1 && a->epi_mode == 1) return dispatch_sk<1, 1>(*a, stream);
  if (a->pro_mode == 2 && a->epi_mode == 2) return dispatch_sk<2, 2>(*a, stream);
  if (a->pro_mode == 2 && a->epi_mode == 0 && a->bias == nullptr && a->out_act == SPB_ACT_NONE) return dispatch_sk<2, 0>(*a, stream);
  return SPB_E_UNSUPPORTED;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_sk(int on, int min_k, int rf) {
  g_sk_on = on & 1; if (min_k > 0) g_sk_min_k = min_k; g_sk_rf = rf;
  g_sk_max_n = (on & 2) ? (1 << 30) : 320;    // on & 2: also the wide layers (experiments)
  g_sk_max_m = (on & 4) ? 40000 : 4096;       // on & 4: also the 14x14 maps
  return 0;
}
#endif
