// Pointwise (1x1) convolution as NHWC row-major GEMMs on the gfx950 matrix cores.
//
//   forward   Y[M,N]  = act(bn(A))[M,K] * W[N,K]^T        + per-channel sum(y), sum(y^2) for the next BN
//   dgrad     dA[M,K] = bn_bwd(G,Z)[M,N] * Wt[K,N]^T      + output-side activation mask and sum(g), sum(g*xhat)
//   wgrad     dW[N,K] = bn_bwd(G,Z)[M,N]^T * act(bn(X))[M,K]
//   join      Y[M,N]  = (bn(A) + bn2(A2))[M,K] * W[N,K]^T   + the same sums; the first column tile also writes the joined
//             operand (PRO 3: the residual add of the previous inverted-residual block, folded into this block's expand conv)
//
// Replaces nn.Conv2d(k=1) + nn.BatchNorm2d + ReLU6/ReLU/LeakyReLU at reference park2019.py:51-53,64-66,
// revgrad.py:76 and the torchvision MobileNetV2 expand/project convolutions (park2019.py:107-108).
//
// Design notes (MI355X):
//   * 92 % of KRN FLOPs but only ~30 FLOP/B: these GEMMs are HBM-bound except at the 7x7 maps, so the BN affine and
//     the activation are applied while the A tile is staged (registers -> LDS), and the BN batch sums of the output are
//     reduced in the epilogue -- each activation tensor is written once and read once per pass.
//   * 256 threads = 4 waves; workgroup tile 128 x BN; MFMA 16x16x32 bf16 (or the exact 16x16x4 f32 form in parity mode).
//   * the output tile is staged through LDS so global stores are full 16-byte vectors along the channel axis.
//   * workgroups are persistent over M tiles (fixed N tile) so the per-channel sums stay in registers; one set of
//     atomics per workgroup at the end, spread over `oR` replicas of the accumulator.
//   * logical workgroup ids are remapped so the N tiles that share an A tile run on one XCD (one L2).
#include <cstring>
#include "common.h"
#include <hip/hip_ext.h>

int spb_gemm_sk(const spb_gemm_args_t* a, hipStream_t stream);   // gemm_sk.hip
int spb_gemm_os(const spb_gemm_args_t* a, hipStream_t stream);   // gemm_os.hip
int spb_gemm_big(const spb_gemm_args_t* a, hipStream_t stream);  // gemm_big.hip
int spb_gemm_rs(const spb_gemm_args_t* a, hipStream_t stream);   // gemm_rs.hip
int spb_gemm_st(const spb_gemm_args_t* a, hipStream_t stream);   // gemm_st.hip
// (round 5: the IEEE-half twin links these files too -- it holds the KRN kernels now, build.py SOURCES_F16)

// phase timestamps for scratch/ubench_gemm.hip (compiled out in the product build)
#ifndef SPB_TS
#define SPB_TS(i)
#endif

namespace {

// BK = reduction chunk per LDS stage: 32 for the streaming (large M, small K) layers, 64 for the 14x14 / 7x7 maps where a
// launch has few workgroups and the K loop is latency bound (fewer, fatter stages).  BK=128 was measured slower: its two
// prefetch register sets push the dgrad variant to 256 VGPRs (1 wave per SIMD).
template <typename T, int BK> struct LdsPad { static constexpr int LDK = BK + 8; };
template <int BK> struct LdsPad<float, BK> { static constexpr int LDK = BK + 4; };

template <typename T, int RF, int BN, int BK>
constexpr size_t gemm_region_bytes() {
  constexpr int BM = 64 * RF;
  size_t a = (size_t)(BM + BN) * LdsPad<T, BK>::LDK * sizeof(T);
  size_t b = (size_t)BM * (BN + 8) * sizeof(T);
  size_t c = (size_t)2 * 256 * 8 * sizeof(float);  // stats reduction scratch
  size_t m = a > b ? a : b;
  return m > c ? m : c;
}

// RF = 16-row fragments per wave (workgroup tile = 64*RF rows x BN columns)
template <typename T, int RF, int BN, int BK, int PRO, int EPI>
__global__ __launch_bounds__(256, (RF == 1 && sizeof(T) == 2) ? (((BK == 64 && PRO >= 2) || BN == 128) ? 2 : 4) : 1) void pw_gemm_kernel(const spb_gemm_args_t g) {
  constexpr int BM = 64 * RF;
  constexpr int GBK = BK;
  constexpr int LDK = LdsPad<T, BK>::LDK;
  constexpr int BKV = BK / 8;            // vec8 per tile row
  constexpr int AROWS = 256 / BKV;       // tile rows covered by one pass of the 256 threads
  constexpr int NA = BM / AROWS;         // A vectors per thread per stage
  constexpr int LDO = BN + 8;
  constexpr int NF = BN / 16;           // 16-wide column fragments per wave
  constexpr int NV = BN / 8;            // 8-wide column vectors per tile row
  constexpr int VR = 256 / NV;          // rows covered per epilogue sweep
  constexpr int VRI = (BM + VR - 1) / VR;
  constexpr int NBV = (BN * BKV + 255) / 256;  // B-tile vec8 loads per thread

  extern __shared__ __attribute__((aligned(16))) char smem[];
  SPB_TS(0);
  const int M = g.M, K = g.K, N = g.N;
  const int lda = g.lda > 0 ? g.lda : K, ldc = g.ldc > 0 ? g.ldc : N;   // row strides of A (and A2) / Y (and res, Zout)
  const int Kp = (K + GBK - 1) / GBK * GBK;
  float* coef = reinterpret_cast<float*>(smem);  // [3][Kp]
  float* ecoef = coef + 3 * Kp;                   // [4][BN] (EPI 2): scale, shift, mean, inverse std of the input-side BN
  T* As = reinterpret_cast<T*>(smem + (size_t)(3 * Kp + (EPI == 2 ? 4 * BN : 0)) * sizeof(float));
  T* Bs = As + BM * LDK;
  T* Os = As;

  const int t = threadIdx.x;
  const int l = t & 63, w = t >> 6;
  const int li = l & 15, lq = l >> 4;

  const int NT = (N + BN - 1) / BN;
  const int MT = (M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % NT;
  const int GM = gridDim.x / NT;
  const int n0 = nt * BN;

  // epilogue ownership: 8 consecutive output channels, a strided set of rows
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < N;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // backward epilogue: scale/shift of the input-side BN are read from LDS where they are used, and the second BN sum
  // is accumulated as sum g*z and turned into sum g*xhat = invstd*(sum g*z - mean*sum g) in the final reduction.
  // (With mean/invstd/scale/shift in registers this variant sat at 164 VGPRs = 2 workgroups per CU, and the
  // 882-workgroup layers ran in two rounds.)
  float e_bias[8];
  if (EPI == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e_bias[j] = (colok && g.bias) ? g.bias[nE + j] : 0.f;
  }

  const T* Ag = reinterpret_cast<const T*>(g.A);
  const T* A2g = reinterpret_cast<const T*>(g.A2);
  const T* Bg = reinterpret_cast<const T*>(g.Bw);
  T* Yg = reinterpret_cast<T*>(g.Y);
  const T* Rg = reinterpret_cast<const T*>(g.res);
  const T* Zg = reinterpret_cast<const T*>(g.Zout);
  T* Ymat = reinterpret_cast<T*>(g.Ymat);

  const int kvA = t % BKV, rowA = t / BKV;  // A tile: rows rowA + AROWS*i
  const int KT = Kp / GBK;

  // Two register sets: the loads of reduction chunks kt+1 and kt+2 are in flight while chunk kt is on the matrix cores
  // (one set left the K loop of a 48-workgroup layer latency-bound at 2.3 us per chunk).  Clamped addresses, no
  // branches around loads.
  Raw8<T> ra[2][NA], ra2[2][NA], rb[2][NBV];
#define LOAD_TILE(S, m0_, kt)                                                                 \
    {                                                                                         \
      const int k = (kt) * GBK + kvA * 8;                                                     \
      const int kc = k < K ? k : K - 8;                                                       \
      _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                        \
        const int m = (m0_) + rowA + AROWS * i;                                               \
        const size_t o = (size_t)(m < M ? m : M - 1) * lda + kc;                                \
        ra[S][i] = ldraw<T>(Ag + o);                                                          \
        if (PRO >= 2) { if (A2g) ra2[S][i] = ldraw<T>(A2g + o); }                             \
      }                                                                                       \
      _Pragma("unroll") for (int i = 0; i < NBV; ++i) {                                       \
        const int e = t + 256 * i;                                                            \
        const int rb_ = e / BKV, kb = (kt) * GBK + (e % BKV) * 8;                             \
        const int n = n0 + (rb_ < BN ? rb_ : BN - 1);                                         \
        rb[S][i] = ldraw<T>(Bg + (size_t)(n < N ? n : N - 1) * K + (kb < K ? kb : K - 8));    \
      }                                                                                       \
    }
#define STORE_TILE(S, m0_, kt)                                                                \
    {                                                                                         \
      const int k = (kt) * GBK + kvA * 8;                                                     \
      _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                        \
        const int m = (m0_) + rowA + AROWS * i;                                               \
        float v[8], a1[8], a2[8];                                                             \
        const bool ok = (m < M && k < K);                                                     \
        cvt8(ra[S][i], a1);                                                                   \
        if (PRO >= 2) {                                                                       \
          if (A2g) cvt8(ra2[S][i], a2);                                                       \
          else { _Pragma("unroll") for (int j = 0; j < 8; ++j) a2[j] = 0.f; }                 \
        }                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                       \
          float x;                                                                            \
          if (PRO == 1) x = act_fwd(a1[j] * coef[k + j] + coef[Kp + k + j], g.pro.act, g.pro.slope); \
          else x = a1[j] * coef[k + j] + a2[j] * coef[Kp + k + j] + coef[2 * Kp + k + j];     \
          v[j] = ok ? x : 0.f;                                                                \
        }                                                                                     \
        st8<T>(As + (rowA + AROWS * i) * LDK + kvA * 8, v);                                   \
        if (PRO == 3) { if (nt == 0 && ok) st8<T>(Ymat + (size_t)m * lda + k, v); }           \
      }                                                                                       \
      _Pragma("unroll") for (int i = 0; i < NBV; ++i) {                                       \
        const int e = t + 256 * i;                                                            \
        const int rb_ = e / BKV, kb = (kt) * GBK + (e % BKV) * 8;                             \
        if (rb_ < BN) {                                                                       \
          float v[8];                                                                         \
          cvt8(rb[S][i], v);                                                                  \
          const bool ok = (n0 + rb_ < N) && (kb < K);                                         \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;               \
          st8<T>(Bs + rb_ * LDK + (e % BKV) * 8, v);                                          \
        }                                                                                     \
      }                                                                                       \
    }
#define MMA_TILE()                                                                                            \
    if constexpr (sizeof(T) == 2) {                                                                            \
      _Pragma("unroll") for (int ks = 0; ks < GBK / 32; ++ks) {                                                \
        bf16x8_t af[RF];                                                                                       \
        _Pragma("unroll") for (int i = 0; i < RF; ++i)                                                         \
          af[i] = *reinterpret_cast<const bf16x8_t*>(As + (w * 16 * RF + i * 16 + li) * LDK + ks * 32 + lq * 8); \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                       \
          const bf16x8_t bfv = *reinterpret_cast<const bf16x8_t*>(Bs + (j * 16 + li) * LDK + ks * 32 + lq * 8); \
          _Pragma("unroll") for (int i = 0; i < RF; ++i)                                                       \
            acc[i][j] = SPB_MFMA16(af[i], bfv, acc[i][j]);               \
        }                                                                                                      \
      }                                                                                                        \
    } else {                                                                                                   \
      _Pragma("unroll") for (int kk = 0; kk < GBK / 4; ++kk) {                                                 \
        float af[RF];                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RF; ++i) af[i] = As[(w * 16 * RF + i * 16 + li) * LDK + kk * 4 + lq]; \
        _Pragma("unroll") for (int j = 0; j < NF; ++j) {                                                       \
          const float bfv = Bs[(j * 16 + li) * LDK + kk * 4 + lq];                                             \
          _Pragma("unroll") for (int i = 0; i < RF; ++i)                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfv, acc[i][j], 0, 0, 0);                  \
        }                                                                                                      \
      }                                                                                                        \
    }

  // the first two operand chunks go out BEFORE the coefficient table: the table's loads (the producer's batch sums) and
  // the operand loads then share one memory round trip instead of following each other
  if (lid / NT < MT) {
    LOAD_TILE(0, (lid / NT) * BM, 0);
    if (KT > 1) LOAD_TILE(1, (lid / NT) * BM, 1);
  }
  // ---- prologue coefficients for every reduction channel (derived from the producer's raw batch sums)
  BNEpiPre epre;
  if (EPI == 2) bn_epi_issue(g.epi, n0, N, BN, t, epre);
  if constexpr (PRO == 3) bn_join_table(g.pro, g.pro2, K, Kp, coef, t);
  else bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);
  if (EPI == 2) bn_epi_finish<true>(g.epi, n0, N, BN, BN, ecoef, t, epre);
  lds_barrier();  // coefficients visible
  SPB_TS(1);

  for (int mt = lid / NT; mt < MT; mt += GM) {
    const int m0 = mt * BM;
    f32x4_t acc[RF][NF];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < KT; kt += 2) {
      STORE_TILE(0, m0, kt);
      lds_barrier();
      if (kt + 2 < KT) LOAD_TILE(0, m0, kt + 2);
      MMA_TILE();
      lds_barrier();
      if (kt + 1 < KT) {
        STORE_TILE(1, m0, kt + 1);
        lds_barrier();
        if (kt + 3 < KT) LOAD_TILE(1, m0, kt + 3);
        MMA_TILE();
        lds_barrier();
      }
    }

    SPB_TS(2);
    // next M tile's first chunks and this tile's output-side operands: in flight during the epilogue
    if (mt + GM < MT) {
      LOAD_TILE(0, (mt + GM) * BM, 0);
      if (KT > 1) LOAD_TILE(1, (mt + GM) * BM, 1);
    }
    Raw8<T> zr[EPI == 2 ? VRI : 1], rr[EPI == 2 ? VRI : 1];
    if (EPI == 2) {
#pragma unroll
      for (int s = 0; s < VRI; ++s) {
        const int m = m0 + vrow0 + s * VR;
        const size_t o = (size_t)(m < M ? m : M - 1) * ldc + (colok ? nE : 0);
        zr[EPI == 2 ? s : 0] = ldraw<T>(Zg + o);
        if (Rg) rr[EPI == 2 ? s : 0] = ldraw<T>(Rg + o);
      }
    }

    // ---- accumulators -> LDS (C layout: col = lane&15, row = (lane>>4)*4 + r)
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Os[(w * 16 * RF + i * 16 + lq * 4 + r) * LDO + j * 16 + li] = from_f<T>(acc[i][j][r]);
    lds_barrier();

    // ---- coalesced epilogue: 16-byte vectors along the channel axis
    if (colok) {
#pragma unroll
      for (int s = 0; s < VRI; ++s) {
        const int r = vrow0 + s * VR;
        const int m = m0 + r;
        if (r < BM && m < M) {
          float v[8];
          ld8<T>(Os + r * LDO + vcol * 8, v);
          const size_t o = (size_t)m * ldc + nE;
          if (EPI == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * g.out_scale + e_bias[j], g.out_act, 0.f);
            st8<T>(Yg + o, v);
          } else if (EPI == 1) {
            st8<T>(Yg + o, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
          } else {
            float z[8], rv[8], e_sc[8], e_sh[8];
            cvt8(zr[EPI == 2 ? s : 0], z);
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
              *reinterpret_cast<float4*>(e_sc + j) = *reinterpret_cast<const float4*>(ecoef + vcol * 8 + j);
              *reinterpret_cast<float4*>(e_sh + j) = *reinterpret_cast<const float4*>(ecoef + BN + vcol * 8 + j);
            }
            if (Rg) {
              cvt8(rr[EPI == 2 ? s : 0], rv);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = z[j] * e_sc[j] + e_sh[j];
              v[j] = rnd<T>(v[j] * act_grad(u, g.epi.act, g.epi.slope));
              s1[j] += v[j];
              s2[j] += v[j] * z[j];
            }
            st8<T>(Yg + o, v);
          }
        }
      }
    }
    lds_barrier();
    SPB_TS(3);
  }
#undef LOAD_TILE
#undef STORE_TILE
#undef MMA_TILE

  // ---- per-channel batch sums: reduce over the workgroup, one atomic per channel per workgroup
  if (EPI != 0) {
    float* Rs = reinterpret_cast<float*>(As);  // [2][VR][BN]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Rs[vrow0 * BN + vcol * 8 + j] = s1[j];
      Rs[VR * BN + vrow0 * BN + vcol * 8 + j] = s2[j];
    }
    lds_barrier();
    if (t < 2 * BN) {
      const int which = t / BN, c = t % BN;
      float s = 0.f;
      for (int r = 0; r < VR; ++r) s += Rs[which * VR * BN + r * BN + c];
      if (n0 + c < N) {
        if (EPI == 2 && which == 1) {   // sum g*z -> sum g*xhat
          float sg = 0.f;
          for (int r = 0; r < VR; ++r) sg += Rs[r * BN + c];
          const float mu = ecoef[2 * BN + c], is = ecoef[3 * BN + c];   // kept from the prologue
          s = is * (s - mu * sg);
        }
        const int rep = blockIdx.x % g.oR;
        atomicAdd(g.osums + (size_t)rep * 2 * N + (size_t)which * N + n0 + c, s);
      }
    }
  }
  SPB_TS(4);
}

// Most workgroups of a tiled launch; beyond it the workgroups loop over M tiles.  The 64-row instances need 105-128 VGPRs, so
// 4 workgroups (1024 on the chip) are resident at once (scratch/occ_probe.hip); the 1323- and 1764-workgroup launches of the
// 14x14 / 28x28 layers ran a second, nearly empty round (phase timestamps: last workgroup started 7-11 us into a 12-19 us
// launch).  Measured in the step: 2048 -> 3.144 ms, 1280 -> 3.111, 1024 -> 3.109, 900 -> 3.116.  spb_debug_set_gemm_wg_cap
int g_gemm_wg_cap = 768;    // end of round 4 (with the 384-workgroup weight gradients beside it): 512 -> 2.669 ms, 640 -> 2.677, 768 -> 2.663, 896 -> 2.675, 1024 -> 2.685

template <typename T, int RF, int BN, int BK, int PRO, int EPI>
int launch_gemm(const spb_gemm_args_t& g, hipStream_t stream) {
  constexpr int BM = 64 * RF;
  constexpr int GBK = BK;
  const int NT = (g.N + BN - 1) / BN;
  const int MT = (g.M + BM - 1) / BM;
  // persistent over M tiles: at most g_gemm_wg_cap workgroups, and an even split of the tiles (19 tiles on 16 workgroup rows
  // made the slowest row take two tiles: 2x the layer time)
  const int cap = g_gemm_wg_cap / NT > 8 ? g_gemm_wg_cap / NT : 8;
  int GM = MT;
  if (GM > cap) {
    const int rounds = (MT + cap - 1) / cap;
    GM = (MT + rounds - 1) / rounds;
    if (GM >= 8 && (GM & 7)) GM = (GM + 7) / 8 * 8;  // multiple of 8 keeps the XCD remap bijective
  }
  const int Kp = (g.K + GBK - 1) / GBK * GBK;
  const size_t lds = (size_t)(3 * Kp + (EPI == 2 ? 4 * BN : 0)) * sizeof(float) + gemm_region_bytes<T, RF, BN, BK>();
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<T, RF, BN, BK, PRO, EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_gemm_kernel<T, RF, BN, BK, PRO, EPI>), dim3(NT * GM), dim3(256), (unsigned)lds, stream, nullptr,
                          (hipEvent_t)g.stop_event, 0, g);
  else
    hipLaunchKernelGGL((pw_gemm_kernel<T, RF, BN, BK, PRO, EPI>), dim3(NT * GM), dim3(256), lds, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for the 14x14 / 7x7 maps (M <= 16K rows, bf16).  There a launch has only 100-900 workgroups and
// the register-prefetch kernel above is latency bound: 1.6 us per 64-deep reduction stage, measured, because only
// two stages (8-16 KB per workgroup) are in flight against a ~3 us loaded memory latency.  Here the raw operand tiles go
// global -> LDS with `global_load_lds_dwordx4` (no VGPRs), DS stages deep, counted `s_waitcnt vmcnt(N)` + one raw
// `s_barrier` per stage; the BN / BN-backward transform moves to the fragment read (8 values per lane per MFMA step).
// The LDS image is lane-linear (DMA constraint), so the 16-byte slot of a row is XOR-swizzled with (row & 7) on the
// SOURCE address and on the fragment read (conflict-free ds_read_b128 instead of 8-way).
constexpr int DBM = 64, DBN = 64, DBK = 64, DS = 4;

// One LDS-DMA instruction: lane l's 16 bytes at gsrc land at LDS byte address lds_base + 16*l (lds_base wave-uniform).
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void pw_gemm_dma_kernel(const spb_gemm_args_t g) {
  typedef bf16_t T;
  constexpr int IPS = PRO == 2 ? 6 : 4;                        // DMA instructions per stage per wave
  constexpr int STAGE = (PRO == 2 ? 3 : 2) * DBM * DBK * 2;    // bytes: A [, A2], B tiles of 64 x 64 bf16
  constexpr int LDO = DBN + 8, NV = DBN / 8, VR = 256 / NV, VRI = DBM / VR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = g.M, K = g.K, N = g.N;
  const int lda = g.lda > 0 ? g.lda : K, ldc = g.ldc > 0 ? g.ldc : N;
  const int Kp = (K + DBK - 1) / DBK * DBK, KT = Kp / DBK;
  float* coef = reinterpret_cast<float*>(smem);                // [3][Kp]; PRO 0 (plain operands, SPN): no table
  char* stages = smem + (PRO == 0 ? 0 : (size_t)3 * Kp * sizeof(float));
  T* Os = reinterpret_cast<T*>(stages);                        // aliases the stage ring after the K loop

  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int NT = (N + DBN - 1) / DBN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (lid % NT) * DBN, m0 = (lid / NT) * DBM;

  const T* Ag = reinterpret_cast<const T*>(g.A);
  const T* A2g = (PRO == 2 && g.A2) ? reinterpret_cast<const T*>(g.A2) : Ag;  // identity prologue: p1 == 0, any finite data
  const T* Bg = reinterpret_cast<const T*>(g.Bw);
  T* Yg = reinterpret_cast<T*>(g.Y);
  const T* Rg = reinterpret_cast<const T*>(g.res);
  const T* Zg = reinterpret_cast<const T*>(g.Zout);

  // per-lane DMA sources: row (within the tile) = w*16 + i*8 + (l>>3); 16-byte slot l&7 holds k-vector slot^(row&7)
  const int drow = w * 16 + (l >> 3);
  const int dkv = (l & 7) ^ (drow & 7);                        // (row+8)&7 == row&7: same for i = 0, 1
  size_t arow[2], brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + drow + 8 * i, n = n0 + drow + 8 * i;
    arow[i] = (size_t)(m < M ? m : M - 1) * lda;
    brow[i] = (size_t)(n < N ? n : N - 1) * K;
  }
  const unsigned stages_lds = lds_addr(stages);
  const unsigned wave_ro = __builtin_amdgcn_readfirstlane((unsigned)(w * 16 * DBK * 2));  // provably wave-uniform for "s"
#define DMA_STAGE(kt_)                                                                          \
  {                                                                                             \
    const unsigned sb = stages_lds + (unsigned)(((kt_) % DS) * STAGE) + wave_ro;                \
    const int k = (kt_) * DBK + dkv * 8;                                                        \
    const int kc = k < K ? k : K - 8;                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
      const unsigned ro = (unsigned)(i * 8 * DBK * 2);                                          \
      dma16(Ag + arow[i] + kc, sb + ro);                                                        \
      if (PRO == 2) dma16(A2g + arow[i] + kc, sb + DBM * DBK * 2 + ro);                         \
      dma16(Bg + brow[i] + kc, sb + (PRO == 2 ? 2 : 1) * DBM * DBK * 2 + ro);                   \
    }                                                                                           \
  }
  for (int s = 0; s < DS - 1 && s < KT; ++s) DMA_STAGE(s);

  // ---- prologue coefficients (ordinary loads; they complete before the first counted wait)
  if (PRO != 0)
  bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < N;
  float e_sc[8], e_sh[8], e_mu[8], e_is[8], e_bias[8];
  if (EPI == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      e_sc[j] = 1.f; e_sh[j] = 0.f; e_mu[j] = 0.f; e_is[j] = 0.f;
      if (colok && g.epi.gamma != nullptr) {
        bn_moments(g.epi, nE + j, e_mu[j], e_is[j]);
        e_sc[j] = g.epi.gamma[nE + j] * e_is[j];
        e_sh[j] = g.epi.beta[nE + j] - e_mu[j] * e_sc[j];
      }
    }
  }
  if (EPI == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e_bias[j] = (colok && g.bias) ? g.bias[nE + j] : 0.f;
  }
  // output-side operands of the epilogue: issued now, consumed after the K loop
  Raw8<T> zr[EPI == 2 ? VRI : 1], rr[EPI == 2 ? VRI : 1];
  if (EPI == 2) {
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int m = m0 + vrow0 + s * VR;
      const size_t o = (size_t)(m < M ? m : M - 1) * ldc + (colok ? nE : 0);
      zr[EPI == 2 ? s : 0] = ldraw<T>(Zg + o);
      if (Rg) rr[EPI == 2 ? s : 0] = ldraw<T>(Rg + o);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // simplest correct start: everything issued so far has landed
  __syncthreads();

  f32x4_t acc[DBN / 16];
#pragma unroll
  for (int j = 0; j < DBN / 16; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
  const int frow = w * 16 + li;

  for (int kt = 0; kt < KT; ++kt) {
    // stage kt has landed once at most min(DS-2, KT-1-kt) younger stages are still in flight
    if (kt > 0) {
      const int rem = KT - 1 - kt;
      if (rem >= DS - 2) wait_vmcnt<(DS - 2) * IPS>();
      else if (rem == 1) wait_vmcnt<IPS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // every wave's share of stage kt is visible; everyone is done reading stage kt-1
    }
    if (kt + DS - 1 < KT) DMA_STAGE(kt + DS - 1);  // into the buffer stage kt-1 just vacated
    const char* sb = stages + (size_t)(kt % DS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < DBK / 32; ++ks) {
      const int v = ks * 4 + lq;                   // k-vector of this lane
      const int kb = kt * DBK + v * 8;
      const int so = frow * (DBK * 2) + ((v ^ (frow & 7)) << 4);
      Raw8<T> ar, a2r;
      ar.u = *reinterpret_cast<const uint4*>(sb + so);
      uint4 pa;
      if constexpr (PRO == 0) {
        pa = ar.u;
      } else {
      if (PRO == 2) a2r.u = *reinterpret_cast<const uint4*>(sb + DBM * DBK * 2 + so);
      float a[8], a2[8], x[8];
      cvt8(ar, a);
      if (PRO == 2) cvt8(a2r, a2);
      const float4 c0a = *reinterpret_cast<const float4*>(coef + kb), c0b = *reinterpret_cast<const float4*>(coef + kb + 4);
      const float4 c1a = *reinterpret_cast<const float4*>(coef + Kp + kb), c1b = *reinterpret_cast<const float4*>(coef + Kp + kb + 4);
      const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
      const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
      if (PRO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = a[j] * c0[j] + c1[j];
          x[j] = fminf(fmaxf(u, 0.f), act_h) + act_n * fminf(u, 0.f);
        }
      } else {
        const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb), c2b = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb + 4);
        const float c2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[j] * c0[j] + a2[j] * c1[j] + c2[j];
      }
      pa.x = (uint32_t)f2bf(x[0]) | ((uint32_t)f2bf(x[1]) << 16); pa.y = (uint32_t)f2bf(x[2]) | ((uint32_t)f2bf(x[3]) << 16);
      pa.z = (uint32_t)f2bf(x[4]) | ((uint32_t)f2bf(x[5]) << 16); pa.w = (uint32_t)f2bf(x[6]) | ((uint32_t)f2bf(x[7]) << 16);
      }
      if (kb >= K) pa = make_uint4(0, 0, 0, 0);    // reduction padding: clamped (finite) data times an explicit zero
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, pa);
#pragma unroll
      for (int j = 0; j < DBN / 16; ++j) {
        const int brow_ = j * 16 + li;
        uint4 pb = *reinterpret_cast<const uint4*>(sb + (PRO == 2 ? 2 : 1) * DBM * DBK * 2 + brow_ * (DBK * 2) + ((v ^ (brow_ & 7)) << 4));
        if (kb >= K || n0 + brow_ >= N) pb = make_uint4(0, 0, 0, 0);
        acc[j] = SPB_MFMA16(af, __builtin_bit_cast(bf16x8_t, pb), acc[j]);
      }
    }
  }
#undef DMA_STAGE
  __syncthreads();  // all DMA consumed (the last stages were waited with vmcnt(0)); the ring can be reused

  // ---- accumulators -> LDS (C layout: col = lane&15, row = (lane>>4)*4 + r), then coalesced 16-byte epilogue
#pragma unroll
  for (int j = 0; j < DBN / 16; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(w * 16 + lq * 4 + r) * LDO + j * 16 + li] = f2bf(acc[j][r]);
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (colok) {
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int r = vrow0 + s * VR;
      const int m = m0 + r;
      if (m < M) {
        float v[8];
        ld8<T>(Os + r * LDO + vcol * 8, v);
        const size_t o = (size_t)m * ldc + nE;
        if (EPI == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * g.out_scale + e_bias[j], g.out_act, 0.f);
          st8<T>(Yg + o, v);
        } else if (EPI == 1) {
          st8<T>(Yg + o, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        } else {
          float z[8], rv[8];
          cvt8(zr[EPI == 2 ? s : 0], z);
          if (Rg) {
            cvt8(rr[EPI == 2 ? s : 0], rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float u = z[j] * e_sc[j] + e_sh[j];
            v[j] = rnd<T>(v[j] * act_grad(u, g.epi.act, g.epi.slope));
            s1[j] += v[j];
            s2[j] += v[j] * ((z[j] - e_mu[j]) * e_is[j]);
          }
          st8<T>(Yg + o, v);
        }
      }
    }
  }
  if (EPI != 0) {
    __syncthreads();
    float* Rs = reinterpret_cast<float*>(stages);  // [2][VR][DBN]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Rs[vrow0 * DBN + vcol * 8 + j] = s1[j];
      Rs[VR * DBN + vrow0 * DBN + vcol * 8 + j] = s2[j];
    }
    __syncthreads();
    if (t < 2 * DBN) {
      const int which = t / DBN, c = t % DBN;
      float s = 0.f;
      for (int r = 0; r < VR; ++r) s += Rs[which * VR * DBN + r * DBN + c];
      if (n0 + c < N) atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + (size_t)which * N + n0 + c, s);
    }
  }
}

template <int PRO, int EPI>
int launch_gemm_dma(const spb_gemm_args_t& g, hipStream_t stream) {
  const int NT = (g.N + DBN - 1) / DBN, MT = (g.M + DBM - 1) / DBM;
  const int Kp = (g.K + DBK - 1) / DBK * DBK;
  const size_t lds = (PRO == 0 ? 0 : (size_t)3 * Kp * sizeof(float)) + (size_t)DS * (PRO == 2 ? 3 : 2) * DBM * DBK * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_dma_kernel<PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  hipLaunchKernelGGL((pw_gemm_dma_kernel<PRO, EPI>), dim3(NT * MT), dim3(256), lds, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

// The LDS-DMA ring variant is kept for experiments (spb_debug_set_gemm_dma(1)); measured in the full KRN step it is
// slower than the register-prefetch kernel on the small-M layers it was written for (dgrad 1.52 vs 1.38 ms / step).
bool g_disable_dma = true;
int g_dma_min_k = 64;      // spb_debug_set_gemm_dma(v): v == 1 -> every K >= 64, v > 1 -> only reductions K >= v
bool g_plain_dma = true;
int g_bk64_min_k = 256;
int g_wide_min_n = 1 << 30;    // spb_debug_set_gemm_wide_min_n
int g_bk64_dgrad_min_k = 1 << 30;   // spb_debug_set_gemm_bk64_dgrad_min_k

template <typename T, int PRO, int EPI>
int dispatch_bn(const spb_gemm_args_t& g, hipStream_t stream) {
  const int N = g.N;
  int bn;
  if (N <= 32) bn = 32;
  else if (N <= 64) bn = 64;
  else {
    const int p64 = (N + 63) / 64 * 64, p128 = (N + 127) / 128 * 128;
    bn = (p128 <= p64) ? 128 : 64;
  }
  if (sizeof(T) == 4 && bn == 128) bn = 64;  // parity mode: keep the LDS footprint small
  if (EPI == 2 && bn == 128) bn = 64;        // backward epilogue hoists 2 operand vectors per output row sweep
  // small M (the 14x14 and 7x7 maps): 64-row tiles and 64-column tiles so the launch has enough workgroups
  const bool small_m = g.M <= 40000;   // 28x28 and below: 64-row tiles at 4 waves/SIMD beat 128-row tiles at 2
  // wide outputs behind a long reduction (the 7x7 ConvDw layers, N = 1024, K = 320..1280): 64 x 128 tiles with 64-deep chunks
  if constexpr (sizeof(T) == 2 && PRO == 1 && EPI != 2) {
    if (small_m && bn == 128 && g.N >= g_wide_min_n && g.K >= g_bk64_min_k) return launch_gemm<T, 1, 128, 64, PRO, EPI>(g, stream);
  }
  if (small_m && bn == 128) bn = 64;
  if (small_m) {
    if (bn == 32) return launch_gemm<T, 1, 32, 32, PRO, EPI>(g, stream);
    if constexpr (PRO != 3) { if (g.K >= g_dma_min_k && sizeof(T) == 2 && !g_disable_dma) return launch_gemm_dma<PRO, EPI>(g, stream); }
    // long reductions (the 7x7 ConvDw layers, K up to 1280): 64-wide chunks halve the number of latency-bound steps
    // (forward-type only: the backward variant spills 87 dwords at 128 VGPRs with two 64-wide prefetch sets: 0.68 -> 0.75 ms)
    if (sizeof(T) == 2 && PRO == 1 && g.K >= g_bk64_min_k) return launch_gemm<T, 1, 64, 64, PRO, EPI>(g, stream);
    // backward-type with a long reduction: 32-column tiles leave room in the register file for 64-wide chunks (half the
    // latency-bound steps) and double the workgroup count of launches that fill less than half the chip
    if constexpr (sizeof(T) == 2 && PRO == 2) {
      if (g.K >= g_bk64_dgrad_min_k) return launch_gemm<T, 1, 64, 64, PRO, EPI>(g, stream);
    }
    return launch_gemm<T, 1, 64, 32, PRO, EPI>(g, stream);
  }
  if (bn == 32) return launch_gemm<T, 2, 32, 32, PRO, EPI>(g, stream);
  if (bn == 64) return launch_gemm<T, 2, 64, 32, PRO, EPI>(g, stream);
  return launch_gemm<T, 2, 128, 32, PRO, EPI>(g, stream);
}

template <typename T>
int dispatch_modes(const spb_gemm_args_t& g, hipStream_t stream) {
  if (g.pro_mode == 0 && g.epi_mode == 0) {   // plain operands (SPN): bf16 streams both tiles through the LDS-DMA ring
    if (sizeof(T) == 2 && g.K >= 64 && g_plain_dma) return launch_gemm_dma<0, 0>(g, stream);
    return dispatch_bn<T, 1, 0>(g, stream);   // the caller's identity `pro` reference makes the prologue a no-op
  }
  if (g.pro_mode == 1 && g.epi_mode == 1) return dispatch_bn<T, 1, 1>(g, stream);
  if (g.pro_mode == 1 && g.epi_mode == 0) return dispatch_bn<T, 1, 0>(g, stream);
  if (g.pro_mode == 2 && g.epi_mode == 0) return dispatch_bn<T, 2, 0>(g, stream);
  if (g.pro_mode == 2 && g.epi_mode == 2) return dispatch_bn<T, 2, 2>(g, stream);
  if (g.pro_mode == 3 && g.epi_mode == 1) return dispatch_bn<T, 3, 1>(g, stream);
  if (g.pro_mode == 3 && g.epi_mode == 0) return dispatch_bn<T, 3, 0>(g, stream);
  return SPB_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient: dW[n,k] += sum_m dz[m,n] * a[m,k].  The reduction axis (m) is the slow axis of both operands,
// so the staged row-major LDS tiles are read with the gfx950 transpose load (ds_read_b64_tr_b16): lane (i,q) of a
// 16-lane group gets 4 consecutive m for one column, exactly the MFMA operand layout.
constexpr int WT = 64;   // base output tile (n and k): FN = FK = 2 fragments of 16 per wave, 2 x 2 waves
constexpr int WM = 64;   // m rows per LDS stage

// PART: the tile of this row split is stored (plain stores) into slab `split` of g.part ([S][N*K] f32) for spb_partial_reduce
// instead of being added into dW with f32 atomics
// ACTK: activation of the input-side transform -- 0 none, 1 clamp (ReLU / ReLU6), 2 generic (the runtime form costs three vector
// instructions per element where none or one is needed, in a loop that is bound by its vector instruction count)
// FN, FK (round 6): 16-wide fragments per wave along n and k; the workgroup's tile is 32 FN x 32 FK.  Every output tile re-derives its
// operand columns (the BatchNorm-backward form of dz: two loads and three multiply-adds per element; BatchNorm + activation of the input) for
// every row of its split, and the loop is bound by exactly that work: with 64 x 64 tiles a 960 x 160 gradient transforms 5760 elements per
// row where 1120 are distinct.  128-wide tiles along the longer axis halve the repeats of the other operand (and the L2 / HBM re-reads).
template <typename T, bool PART, int ACTK = 2, int FN = 2, int FK = 2>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const spb_wgrad_args_t g, int rows_per_split) {
  constexpr int TN = 32 * FN, TK = 32 * FK;
  constexpr int LDN = TN + 8, LDK = TK + 8;
  constexpr int CVN = TN / 8, CVK = TK / 8;          // 8-column vectors per tile row
  constexpr int RPN = 256 / CVN, RPK = 256 / CVK;    // rows one pass of the 256 threads covers
  constexpr int PN = WM / RPN, PK = WM / RPK;        // passes per 64-row stage
  // two copies of each operand tile: the transform of stage s+1 writes the other copy while slower waves still read stage s, so ONE
  // barrier per stage is enough (with one copy a second barrier had to close every stage)
  __shared__ __attribute__((aligned(16))) T Ds2[2][WM * LDN];
  __shared__ __attribute__((aligned(16))) T Xs2[2][WM * LDK];
  __shared__ float cz[3][TN];
  __shared__ float ca[2][TK];

  const int M = g.M, K = g.K, N = g.N;
  const int NT = (N + TN - 1) / TN, KT = (K + TK - 1) / TK;
  // the NT*KT output tiles of one row split read the same rows of G and X: keep them on one XCD (one L2), back to back
  // (rocprofv3 FETCH_SIZE showed 3x the algorithmic bytes with the tiles of a split spread over the 8 XCDs)
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = lbid % (NT * KT), split = lbid / (NT * KT);
  const int n0 = (tile / KT) * TN, k0 = (tile % KT) * TK;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);

  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int li = l & 15, lq = l >> 4;
  const int wn = w >> 1, wk = w & 1;

  if (t < TN) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (n0 + t < N) bn_bwd_coef(g.pro_dz, n0 + t, p0, p1, p2);
    cz[0][t] = p0; cz[1][t] = p1; cz[2][t] = p2;
  }
  if (t >= 256 - TK) {                 // (TN + TK may be 256: the two ranges overlap in threads only when both tiles are 128 wide)
    const int c = t - (256 - TK);
    float sc = 0.f, sh = 0.f;
    if (k0 + c < K) bn_fwd_coef(g.pro_a, k0 + c, sc, sh);
    ca[0][c] = sc; ca[1][c] = sh;
  }
  lds_barrier();

  const T* Gg = reinterpret_cast<const T*>(g.G);
  const T* Zg = reinterpret_cast<const T*>(g.Zn);
  const T* Xg = reinterpret_cast<const T*>(g.X);

  f32x4_t acc[FN][FK];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FK; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const size_t ldg = g.ldg > 0 ? g.ldg : N, ldx = g.ldx > 0 ? g.ldx : K;   // row strides of G (and Zn) / X
  const int cvn = t % CVN, rwn = t / CVN;   // dz: rows rwn + RPN i, columns cvn*8 .. cvn*8+7
  const int cvk = t % CVK, rwk = t / CVK;   // input: rows rwk + RPK i, columns cvk*8 .. cvk*8+7
  const int nl = n0 + cvn * 8, kl = k0 + cvk * 8;
  const int nlc = nl < N ? nl : N - 8, klc = kl < K ? kl : K - 8;
  Raw8<T> gr[PN], zr[PN], xr[PK];
  // all global loads of a stage are issued together with clamped addresses; the next stage's loads are in flight
  // while the matrix cores work on the current one
#define WG_LOAD(mb_)                                                          \
  _Pragma("unroll") for (int i = 0; i < PN; ++i) {                            \
    const int m = (mb_) + rwn + RPN * i;                                      \
    const size_t mc_ = (size_t)(m < mend ? m : mend - 1);                     \
    gr[i] = ldraw<T>(Gg + mc_ * ldg + nlc);                                   \
    if (Zg) zr[i] = ldraw<T>(Zg + mc_ * ldg + nlc);                           \
  }                                                                           \
  _Pragma("unroll") for (int i = 0; i < PK; ++i) {                            \
    const int m = (mb_) + rwk + RPK * i;                                      \
    const size_t mc_ = (size_t)(m < mend ? m : mend - 1);                     \
    xr[i] = ldraw<T>(Xg + mc_ * ldx + klc);                                   \
  }
  if (mbeg < mend) WG_LOAD(mbeg);
  // this thread's 8 + 8 coefficient columns in registers: read from LDS inside the loop they cost 20 ds_read_b128 per 64-row stage (the
  // barriers are opaque to the compiler, so nothing loop-invariant is hoisted across them) in a kernel with ~40 registers to its name
  const float a_hi = act_hi(g.pro_a.act);
  float z0[8], z1[8], z2[8], a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    z0[j] = cz[0][cvn * 8 + j]; z1[j] = cz[1][cvn * 8 + j]; z2[j] = cz[2][cvn * 8 + j];
    a0[j] = ca[0][cvk * 8 + j]; a1[j] = ca[1][cvk * 8 + j];
  }
  int par = 0;
  for (int mb = mbeg; mb < mend; mb += WM, par ^= 1) {
    T* Ds = Ds2[par];
    T* Xs = Xs2[par];
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      const int r = rwn + RPN * i, m = mb + r;
      float v[8], gg[8], zz[8];
      cvt8(gr[i], gg);
      if (Zg) cvt8(zr[i], zz);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) zz[j] = 0.f;
      }
      const bool okn = (m < mend) && (nl < N);
      // rows past the split / columns past the matrix contribute zeros: masked on the PACKED vector (4 selects instead of 8 per operand)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = gg[j] * z0[j] + zz[j] * z1[j] + z2[j];
      if constexpr (sizeof(T) == 2) {
        uint4 q;
        q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
        if (!okn) q = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Ds + r * LDN + cvn * 8) = q;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = okn ? v[j] : 0.f;
        st8<T>(Ds + r * LDN + cvn * 8, v);
      }
    }
#pragma unroll
    for (int i = 0; i < PK; ++i) {
      const int r = rwk + RPK * i, m = mb + r;
      float v[8], xx[8];
      cvt8(xr[i], xx);
      const bool okk = (m < mend) && (kl < K);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = xx[j] * a0[j] + a1[j];
        v[j] = ACTK == 0 ? u : (ACTK == 1 ? __builtin_amdgcn_fmed3f(u, 0.f, a_hi) : act_fwd(u, g.pro_a.act, g.pro_a.slope));
      }
      if constexpr (sizeof(T) == 2) {
        uint4 q;
        q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
        if (!okk) q = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Xs + r * LDK + cvk * 8) = q;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = okk ? v[j] : 0.f;
        st8<T>(Xs + r * LDK + cvk * 8, v);
      }
    }
    lds_barrier();
    if (mb + WM < mend) WG_LOAD(mb + WM);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int mc = 0; mc < WM / 32; ++mc) {
        bf16x8_t pf[FN], qf[FK];
        typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
        const int row = mc * 32 + lq * 8 + (li >> 2);
#pragma unroll
        for (int f = 0; f < FN; ++f) {
          const bf16_t* pa = Ds + row * LDN + (wn * FN + f) * 16 + (li & 3) * 4;
          union { struct { s16x4_t lo, hi; } s; bf16x8_t v; } up;
          up.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa));
          up.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa + 4 * LDN));
          pf[f] = up.v;
        }
#pragma unroll
        for (int f = 0; f < FK; ++f) {
          const bf16_t* qa = Xs + row * LDK + (wk * FK + f) * 16 + (li & 3) * 4;
          union { struct { s16x4_t lo, hi; } s; bf16x8_t v; } uq;
          uq.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(qa));
          uq.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(qa + 4 * LDK));
          qf[f] = uq.v;
        }
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FK; ++b)
            acc[a][b] = SPB_MFMA16(pf[a], qf[b], acc[a][b]);
      }
    } else {
#pragma unroll 4
      for (int mm = 0; mm < WM / 4; ++mm) {
        float pf[FN], qf[FK];
#pragma unroll
        for (int f = 0; f < FN; ++f) pf[f] = Ds[(mm * 4 + lq) * LDN + (wn * FN + f) * 16 + li];
#pragma unroll
        for (int f = 0; f < FK; ++f) qf[f] = Xs[(mm * 4 + lq) * LDK + (wk * FK + f) * 16 + li];
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < FK; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a], qf[b], acc[a][b], 0, 0, 0);
      }
    }
  }
#undef WG_LOAD

#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * FN + a) * 16 + lq * 4 + r;
        const int k = k0 + (wk * FK + b) * 16 + li;
        if (n < N && k < K) {
          if (PART) g.part[(size_t)split * N * K + (size_t)n * K + k] = acc[a][b][r];
          else SPB_ATOMIC_W(g.dW + (size_t)n * K + k, acc[a][b][r]);
        }
      }
}

int g_wgrad_target_wgs = 384;    // row splits so that a launch has about this many workgroups (spb_debug_set_wgrad_target).  End of round 4, in the step:
                                 // 256 -> 2.693 ms, 384 -> 2.670, 512 -> 2.685, 768 -> 2.702, 1024 -> 2.727, 2048 -> 2.760 (fewer, longer workgroups on the side
                                 // stream take less from the launch stream's kernels and issue fewer atomics)
int g_wgrad_part_target = 512;   // the same for the partial-store form (its splits cost slab traffic instead of atomics)

// floats of partial-sum scratch launch_wgrad uses at most for this shape (the KRN plan sizes its workspace with it)
long long wgrad_part_floats(int M, int K, int N) {
  const int NT = (N + WT - 1) / WT, KT = (K + WT - 1) / WT;
  int S = spb_ceil_div(g_wgrad_part_target, NT * KT);
  const int maxS = spb_ceil_div(M, 2 * WM);
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  return S >= 2 ? (long long)S * N * K : 0;
}

// 128-wide tiles (tuning build only, spb_debug_set_wgrad_tile): they cut the operand re-derivations by 30-60 % -- and make the STEP slower, at every
// workgroup target (round 6, bs=48 bf16: 64 x 64 2.413-2.422 ms; tiles by cost 2.536 at 384 workgroups, 2.488 at 256, 2.457 at 192, 2.438 at 128, 2.479 at
// 96, 2.574 at 64; 64 x 128 only 2.435, 128 x 64 only 2.436, 128 x 128 only 2.550).  The weight gradients run on the side stream beside the backward chain:
// what they cost the step is what they take from the launch stream's kernels, and fewer, fatter workgroups (53-70 KB of LDS, 2-4x the time each) take more.
int g_wgrad_tile_mode = 0;       // 0: 64 x 64 only; 1: by cost; 2..5: force one tile
int g_wgrad_wide_target = 384;   // workgroup target of a launch with 128-wide tiles (their row splits issue the atomics of a whole tile each)

template <typename T, int FN, int FK>
int launch_wgrad_tile(const spb_wgrad_args_t& g, hipStream_t stream, int target) {
  constexpr int TN = 32 * FN, TK = 32 * FK;
  const int NT = (g.N + TN - 1) / TN, KT = (g.K + TK - 1) / TK;
  int S = spb_ceil_div(target, NT * KT);
  const int maxS = spb_ceil_div(g.M, 2 * WM);
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rps = spb_ceil_div(spb_ceil_div(g.M, S), WM) * WM;
  S = spb_ceil_div(g.M, rps);
  const int act = g.pro_a.act;
  if (act == SPB_ACT_NONE) hipLaunchKernelGGL((pw_wgrad_kernel<T, false, 0, FN, FK>), dim3(NT * KT * S), dim3(256), 0, stream, g, rps);
  else if (act == SPB_ACT_RELU || act == SPB_ACT_RELU6) hipLaunchKernelGGL((pw_wgrad_kernel<T, false, 1, FN, FK>), dim3(NT * KT * S), dim3(256), 0, stream, g, rps);
  else hipLaunchKernelGGL((pw_wgrad_kernel<T, false, 2, FN, FK>), dim3(NT * KT * S), dim3(256), 0, stream, g, rps);
  SPB_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int launch_wgrad(const spb_wgrad_args_t& g, hipStream_t stream) {
  const int NT = (g.N + WT - 1) / WT, KT = (g.K + WT - 1) / WT;
  const long long NK = (long long)g.N * g.K;
  const bool part = g.part != nullptr && g.part_cap >= 2 * NK && (reinterpret_cast<uintptr_t>(g.part) & 15) == 0;
  if (g.job_out) std::memset(g.job_out, 0, sizeof(*g.job_out));
  if (part) {
    int S = spb_ceil_div(g_wgrad_part_target, NT * KT);
    const int maxS = spb_ceil_div(g.M, 2 * WM);
    if (S > maxS) S = maxS;
    if (S > g.part_cap / NK) S = (int)(g.part_cap / NK);
    if (S < 1) S = 1;
    int rps = spb_ceil_div(spb_ceil_div(g.M, S), WM) * WM;
    S = spb_ceil_div(g.M, rps);
    if (S >= 2) {
      hipLaunchKernelGGL((pw_wgrad_kernel<T, true>), dim3(NT * KT * S), dim3(256), 0, stream, g, rps);
      SPB_CHECK_LAUNCH();
      spb_red_job_t job; job.src = g.part; job.dst = g.dW; job.stride = NK; job.n = (int)NK; job.nparts = S;
      if (g.job_out) { *g.job_out = job; return 0; }
      return spb_partial_reduce(&job, 1, stream);
    }
  }
#ifdef SPB_TUNING
  if constexpr (sizeof(T) == 2) {
    if (g_wgrad_tile_mode) {
      // elements transformed per row of the reduction: every tile re-derives its TN columns of dz (weight 2: two loads, three multiply-adds)
      // and its TK columns of the input (weight 1); padded columns are loaded and transformed like real ones
      auto cost = [&](int tn, int tk) { const int nt = (g.N + tn - 1) / tn, kt = (g.K + tk - 1) / tk; return (long long)nt * kt * (2 * tn + tk); };
      const long long c[4] = {cost(64, 64), cost(64, 128), cost(128, 64), cost(128, 128)};
      int pick = 0;
      for (int i = 1; i < 4; ++i)
        if (c[i] < c[pick]) pick = i;
      if (c[pick] * 8 > c[0] * 7) pick = 0;    // a wider tile must save an eighth of the work to be worth its larger atomics footprint per split
      if (g_wgrad_tile_mode > 1) pick = g_wgrad_tile_mode - 2;     // 2..5: force one tile (A/B)
      if (pick == 1) return launch_wgrad_tile<T, 2, 4>(g, stream, g_wgrad_wide_target);
      if (pick == 2) return launch_wgrad_tile<T, 4, 2>(g, stream, g_wgrad_wide_target);
      if (pick == 3) return launch_wgrad_tile<T, 4, 4>(g, stream, g_wgrad_wide_target);
    }
  }
#endif
  return launch_wgrad_tile<T, 2, 2>(g, stream, g_wgrad_target_wgs);
}

// lane l of a wave reads 8 bytes at in + l*4 elements through the transpose load; used by the unit test that pins
// the instruction's semantics on the box.
__global__ void trread_probe(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  const int l = threadIdx.x;
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

}  // namespace

extern "C" int spb_pwconv_gemm(int dtype, const spb_gemm_args_t* a, spb_stream_t stream) {
  if (!a || !a->A || !a->Bw) return SPB_E_ARG;
  if (a->M <= 0 || a->K <= 0 || a->N <= 0 || (a->K & 7) || (a->N & 7)) return SPB_E_SHAPE;
  if (!a->Y) {   // statistics only (epi_mode 1, 16-bit storage, K <= 32): the batch sums of a product that is never stored (gemm_st.hip, SO)
    if (a->epi_mode != 1 || !a->osums || a->oR < 1 || dtype != SPB_BF16) return SPB_E_ARG;
    if (a->pro_mode == 3 && (!a->A2 || !a->Ymat || a->pro.act != SPB_ACT_NONE || a->pro2.act != SPB_ACT_NONE)) return SPB_E_ARG;
    const int es = spb_gemm_st(a, (hipStream_t)stream);
    return es == SPB_E_UNSUPPORTED ? SPB_E_SHAPE : es;
  }
  if (a->epi_mode != 0 && (!a->osums || a->oR < 1)) return SPB_E_ARG;
  if (a->epi_mode == 2 && !a->Zout) return SPB_E_ARG;
  if (a->pro_mode == 3 && (!a->A2 || !a->Ymat || a->pro.act != SPB_ACT_NONE || a->pro2.act != SPB_ACT_NONE)) return SPB_E_ARG;
  if (dtype == SPB_BF16) {
    // 112x112 / 56x56 maps, forward: the streaming kernel (gemm_st.hip)
    const int es = spb_gemm_st(a, (hipStream_t)stream);
    if (es != SPB_E_UNSUPPORTED) return es;
    // 28x28 / 14x14 maps, short reduction, wide output (expand forwards, project input gradients): the row-slab kernel (gemm_rs.hip)
    const int er = spb_gemm_rs(a, (hipStream_t)stream);
    if (er != SPB_E_UNSUPPORTED) return er;
    // the 7x7 maps, wide output behind a long reduction (ConvDw extras, domain classifier): 128 x 128 tiles (gemm_big.hip)
    const int eb = spb_gemm_big(a, (hipStream_t)stream);
    if (eb != SPB_E_UNSUPPORTED) return eb;
    // 28x28 / 14x14 maps, medium reduction, narrow output: the one-shot kernel (gemm_os.hip)
    const int eo = spb_gemm_os(a, (hipStream_t)stream);
    if (eo != SPB_E_UNSUPPORTED) return eo;
    // the 14x14 / 7x7 maps with a long reduction: split-K over the waves of a workgroup (gemm_sk.hip)
    const int e = spb_gemm_sk(a, (hipStream_t)stream);
    if (e != SPB_E_UNSUPPORTED) return e;
    return dispatch_modes<bf16_t>(*a, (hipStream_t)stream);
  }
  if (dtype == SPB_F32) return dispatch_modes<float>(*a, (hipStream_t)stream);
  return SPB_E_ARG;
}

long long spb_wgrad_part_floats(int M, int K, int N) { return wgrad_part_floats(M, K, N); }   // for krn_plan.hip

extern "C" int spb_pwconv_wgrad(int dtype, const spb_wgrad_args_t* a, spb_stream_t stream) {
  if (!a || !a->G || !a->X || !a->dW) return SPB_E_ARG;
  if (a->M <= 0 || a->K <= 0 || a->N <= 0 || (a->K & 7) || (a->N & 7)) return SPB_E_SHAPE;
  if (dtype == SPB_BF16) return launch_wgrad<bf16_t>(*a, (hipStream_t)stream);
  if (dtype == SPB_F32) return launch_wgrad<float>(*a, (hipStream_t)stream);
  return SPB_E_ARG;
}

extern "C" int spb_debug_trread(const unsigned short* in4096, unsigned short* out256, spb_stream_t stream) {
  hipLaunchKernelGGL(trread_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, in4096, out256);
  SPB_CHECK_LAUNCH();
  return 0;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_wgrad_target(int wgs) {   // wgs < 0: target of the partial-store form
  if (wgs < 0) g_wgrad_part_target = -wgs; else g_wgrad_target_wgs = wgs < 1 ? 1 : wgs;
  return 0;
}
extern "C" int spb_debug_set_wgrad_tile(int mode, int wide_target) {   // mode 0: 64 x 64 tiles only (default), 1: by cost, 2..5: force 64x64 / 64x128 / 128x64 / 128x128
  g_wgrad_tile_mode = mode;
  if (wide_target > 0) g_wgrad_wide_target = wide_target;
  return 0;
}
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_dma(int on) { g_disable_dma = (on == 0); g_dma_min_k = on > 1 ? on : 64; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_plain_dma(int on) { g_plain_dma = (on != 0); return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_bk64_min_k(int k) { g_bk64_min_k = k; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_wide_min_n(int n) { g_wide_min_n = n; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_wg_cap(int n) { g_gemm_wg_cap = n < 64 ? 64 : n; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_bk64_dgrad_min_k(int k) { g_bk64_dgrad_min_k = k; return 0; }
#endif

extern "C" const char* spb_version(void) { return "speedplusbaseline_amd gfx950 r1"; }
