// Pointwise (1x1) convolution GEMMs with a medium reduction and a narrow output on the 28x28 / 14x14 maps: "one shot".
//
//   forward   Y[M,N]  = act(bn(A))[M,K] * W[N,K]^T        + sum(y), sum(y^2)            (project convs: K = 192..576, N = 32..96)
//   dgrad     dA[M,N] = bn_bwd(G,Z)[M,K] * Wt[N,K]^T      + mask, sum(g), sum(g*xhat)    (expand convs:  K = 192..576, N = 32..96)
// Same contract as spb_pwconv_gemm (gemm_pw.hip): reference park2019.py:51-53,64-66 and the torchvision MobileNetV2
// expand / project convolutions (park2019.py:107-108).
//
// Why another GEMM.  Phase timestamps of the tiled kernel on these shapes (scratch/ubench_gemm2.hip, M = 9408, K = 384,
// N = 64): coefficient prologue 2.5 us, then SIX barrier-separated 64-deep stages at 0.9 us each -- a memory round trip
// per stage, nothing to hide it behind with one workgroup per CU -- then 2 us in which the statistics barrier waits for the
// output stores to be acknowledged: 13.8 us for 8.4 MB.  And scratch/ubench_ldpat.hip: 147 workgroups pulling 48 / 96 KB
// each out of L2 take +1.1 / +3.7 us over an empty launch with global_load_dwordx4 into registers, and +0.0 us with
// global_load_lds_dwordx4 (same addresses).  Here a launch is one LDS-DMA burst and one pass:
//   * a workgroup owns 64 rows x all N columns (N <= 96); wave w owns rows 16w..16w+15 over the WHOLE reduction;
//   * A (and the second operand of the BatchNorm-backward prologue) and W go to LDS by LDS-DMA, chunk-major
//     ([32-deep chunk][row | n][64 bytes]): the fragment of (chunk, 16-row group) is then one contiguous kilobyte --
//     conflict-free ds_read_b128, and exactly one DMA instruction to fill; a wave loads the A rows it will consume
//     itself, so only W needs the workgroup barrier;
//   * the BatchNorm(+activation) / BatchNorm-backward transform runs on the A fragments in registers (coefficient table in
//     LDS);
//   * when K * (64 + N) does not fit the LDS, the reduction is cut into passes (one DMA burst + barrier each);
//   * the output stores are issued last: the statistics (DPP / v_permlane swaps inside the waves, LDS across them, one
//     atomic per channel and workgroup) never wait for store acknowledgements.
// bf16 only (the f32 parity mode keeps the tiled kernel).
#include "common.h"
#include <hip/hip_ext.h>

#ifndef SPB_TS
#define SPB_TS(i)
#endif
#ifndef OS_ABL             // ablation bits for scratch/ubench_gemm2.hip (0 in the product build)
#define OS_ABL 0
#endif
#ifndef SPB_TS_DECL       // register-held timestamps (scratch/ubench_gemm2.hip): a timestamp STORE would be waited for by the next barrier
#define SPB_TS_DECL
#define SPB_TSR(i)
#define SPB_TS_FLUSH
#endif

namespace {

constexpr int OBM = 64;

// NF = 16-column fragments (N <= 16 * NF).  KS = chunks per pass (host-chosen so that a pass fits the LDS)
template <int PRO, int EPI, int NF>
__global__ __launch_bounds__(256) void pw_os_kernel(const spb_gemm_args_t g, int KS) {
  constexpr int BN = 16 * NF, LDO = BN + 8;
  constexpr int NV = BN / 8;                  // 8-channel vectors per output row
  constexpr int VR = 256 / NV > OBM ? OBM : 256 / NV;   // rows per epilogue sweep
  constexpr int VRI = (OBM + VR - 1) / VR;
  constexpr int NTH = VR * NV;                // threads that take part in the epilogue (NV = 12: 252)
  constexpr int NA = PRO == 2 ? 2 : 1;        // A-side tensors
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SPB_TS_DECL;
  SPB_TSR(0);
  const int M = g.M, K = g.K, N = g.N;
  const int lda = g.lda > 0 ? g.lda : K, ldc = g.ldc > 0 ? g.ldc : N;
  const int KC = (K + 31) >> 5, Kp = KC * 32;
  float* coef = reinterpret_cast<float*>(smem);                       // [3][Kp]
  float* ecoef = coef + 3 * Kp;                                        // [4][BN] scale, shift, mean, inverse std
  float* red = ecoef + 4 * BN;                                         // [4 waves][2][BN]
  bf16_t* Os = reinterpret_cast<bf16_t*>(red + 8 * BN);                // [64][LDO] (NV = 12: also the statistics scratch)
  constexpr int OSB = (OBM * LDO * 2 > 2 * 21 * BN * 4 ? OBM * LDO * 2 : 2 * 21 * BN * 4);
  char* al = reinterpret_cast<char*>(Os) + ((OSB + 1023) & ~1023);     // [NA][KS][4 row groups][1 KB]
  char* wl = al + (size_t)NA * KS * 4096;                              // [KS][NF][1 KB]

  const int t = threadIdx.x, l = t & 63, li = l & 15, lq = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int NT = (N + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (lid % NT) * BN, m0 = (lid / NT) * OBM;

  const bf16_t* Ag = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* A2g = (PRO == 2 && g.A2) ? reinterpret_cast<const bf16_t*>(g.A2) : Ag;   // identity prologue: p1 == 0
  const bf16_t* Bg = reinterpret_cast<const bf16_t*>(g.Bw);
  bf16_t* Yg = reinterpret_cast<bf16_t*>(g.Y);
  const bf16_t* Rg = reinterpret_cast<const bf16_t*>(g.res);
  const bf16_t* Zg = reinterpret_cast<const bf16_t*>(g.Zout);

  // DMA lane roles: lane l fetches row (l >> 2) of a 16-row group, 16-byte piece (l & 3) of the 64-byte chunk
  const int drow = m0 + w * 16 + (l >> 2);
  const size_t arow = (size_t)(drow < M ? drow : M - 1) * lda;
  const unsigned al_lds = lds_addr(al), wl_lds = lds_addr(wl);
  auto issue_pass = [&](int c0, int nc) {
    for (int c = 0; c < ((OS_ABL & 4) ? 0 : nc); ++c) {            // this wave's own 16 rows, every chunk of the pass
      const int k = (c0 + c) * 32 + (l & 3) * 8;
      const int kc = k < K ? k : K - 8;
      dma16(Ag + arow + kc, al_lds + (unsigned)((c * 4 + w) << 10));
      if (PRO == 2) dma16(A2g + arow + kc, al_lds + (unsigned)(((KS + c) * 4 + w) << 10));
    }
    const int nI = (OS_ABL & 2) ? 0 : nc * NF;                   // W: instruction (c, j) = chunk c, 16-column group j
    for (int i = w; i < nI; i += 4) {
      const int c = i / NF, j = i - c * NF;
      const int n = n0 + j * 16 + (l >> 2), k = (c0 + c) * 32 + (l & 3) * 8;
      dma16(Bg + (size_t)(n < N ? n : N - 1) * K + (k < K ? k : K - 8), wl_lds + (unsigned)(i << 10));
    }
  };
  issue_pass(0, KS < KC ? KS : KC);
  // output-side operands of the dgrad epilogue
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < N && t < NTH;
  uint4 zr[EPI == 2 ? VRI : 1], rr[EPI == 2 ? VRI : 1];
  if (EPI == 2) {
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int m = m0 + vrow0 + s * VR;
      const size_t o = (size_t)(m < M ? m : M - 1) * ldc + (colok ? nE : 0);
      zr[EPI == 2 ? s : 0] = *reinterpret_cast<const uint4*>(Zg + o);
      if (Rg) rr[EPI == 2 ? s : 0] = *reinterpret_cast<const uint4*>(Rg + o);
    }
  }
  // ---- coefficient tables (their loads share the round trip of everything above)
  BNEpiPre epre;
  if (EPI == 2) bn_epi_issue(g.epi, n0, N, BN, t, epre);
  if (OS_ABL & 1) { for (int i = t; i < 3 * Kp; i += 256) coef[i] = 1.f; } else
  bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);
  if (EPI == 2) bn_epi_finish<true>(g.epi, n0, N, BN, BN, ecoef, t, epre);

  f32x4_t acc[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
  for (int c0 = 0; c0 < KC; c0 += KS) {
    const int nc = KC - c0 < KS ? KC - c0 : KS;
    if (c0 > 0) { __syncthreads(); issue_pass(c0, nc); }     // every wave is done with the previous pass's W
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA is invisible to the compiler's wait insertion
    __syncthreads();
    if (c0 == 0) SPB_TSR(1);
#pragma unroll 4
    for (int c = 0; c < ((OS_ABL & 8) ? 0 : nc); ++c) {
      const int kb = (c0 + c) * 32 + lq * 8;
      const float4 c0a = *reinterpret_cast<const float4*>(coef + kb), c0b = *reinterpret_cast<const float4*>(coef + kb + 4);
      const float4 c1a = *reinterpret_cast<const float4*>(coef + Kp + kb), c1b = *reinterpret_cast<const float4*>(coef + Kp + kb + 4);
      const float q0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
      const float q1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
      Raw8<bf16_t> r1; r1.u = *reinterpret_cast<const uint4*>(al + ((size_t)(c * 4 + w) << 10) + li * 64 + lq * 16);
      float a[8], x[8];
      cvt8(r1, a);
      if (PRO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = a[j] * q0[j] + q1[j];
          x[j] = __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f);
        }
      } else {
        const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb), c2b = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb + 4);
        const float q2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
        Raw8<bf16_t> r2; r2.u = *reinterpret_cast<const uint4*>(al + ((size_t)((KS + c) * 4 + w) << 10) + li * 64 + lq * 16);
        float a2[8];
        cvt8(r2, a2);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[j] * q0[j] + a2[j] * q1[j] + q2[j];
      }
      uint4 pa;
      pa.x = pack_bf16x2(x[0], x[1]); pa.y = pack_bf16x2(x[2], x[3]); pa.z = pack_bf16x2(x[4], x[5]); pa.w = pack_bf16x2(x[6], x[7]);
      if (kb >= K) pa = make_uint4(0, 0, 0, 0);      // reduction padding: clamped (finite) weights times an explicit zero
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, pa);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const uint4 pb = *reinterpret_cast<const uint4*>(wl + ((size_t)(c * NF + j) << 10) + li * 64 + lq * 16);
        acc[j] = SPB_MFMA16(af, __builtin_bit_cast(bf16x8_t, pb), acc[j]);
      }
    }
  }
  SPB_TSR(2);
  // ---- accumulators -> LDS (C layout: col = lane & 15, row = (lane >> 4) * 4 + r), then the coalesced 16-byte epilogue
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(w * 16 + lq * 4 + r) * LDO + j * 16 + li] = f2bf(acc[j][r]);
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  uint4 outv[VRI];                             // the stores are issued after the statistics (see the header)
  if (colok) {
    float e_sc[8], e_sh[8];
    if (EPI == 2) {
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        *reinterpret_cast<float4*>(e_sc + j) = *reinterpret_cast<const float4*>(ecoef + vcol * 8 + j);
        *reinterpret_cast<float4*>(e_sh + j) = *reinterpret_cast<const float4*>(ecoef + BN + vcol * 8 + j);
      }
    }
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int r = vrow0 + s * VR;
      const int m = m0 + r;
      const bool ok = r < OBM && m < M;
      outv[s] = *reinterpret_cast<const uint4*>(Os + (r < OBM ? r : 0) * LDO + vcol * 8);
      Raw8<bf16_t> oq; oq.u = outv[s];
      float v[8];
      cvt8(oq, v);
      if (EPI == 1) {
        if (ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
      } else {
        Raw8<bf16_t> zq; zq.u = zr[EPI == 2 ? s : 0];
        float z[8];
        cvt8(zq, z);
        if (Rg) {
          Raw8<bf16_t> rq; rq.u = rr[EPI == 2 ? s : 0];
          float rv[8];
          cvt8(rq, rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = z[j] * e_sc[j] + e_sh[j];
          v[j] = rnd<bf16_t>(v[j] * act_grad(u, g.epi.act, g.epi.slope));
          if (ok) { s1[j] += v[j]; s2[j] += v[j] * z[j]; }
        }
        outv[s].x = pack_bf16x2(v[0], v[1]); outv[s].y = pack_bf16x2(v[2], v[3]);
        outv[s].z = pack_bf16x2(v[4], v[5]); outv[s].w = pack_bf16x2(v[6], v[7]);
      }
    }
  }
  SPB_TSR(3);
  // ---- per-channel batch sums.  Threads with equal t % NV own the same 8 channels: inside a wave those are the lanes
  // l = vcol (mod NV).  For NV = 8 / 4 (BN = 64 / 32) that is a DPP rotation class plus the two half swaps; for NV = 12 the
  // classes do not align with the lanes and the partials meet in LDS instead.
  if constexpr (NV == 8 || NV == 4) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = s1[j], b = s2[j];
      if (NV == 4) {
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x124, 0xf, 0xf, true));
        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x124, 0xf, 0xf, true));
      }
      a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xf, 0xf, true));
      b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x128, 0xf, 0xf, true));
      a = xor32_sum(xor16_sum(a)); b = xor32_sum(xor16_sum(b));
      if (l < NV) { red[(w * 2 + 0) * BN + l * 8 + j] = a; red[(w * 2 + 1) * BN + l * 8 + j] = b; }
    }
  } else {
    __syncthreads();                          // everyone has read its output vectors out of Os
    float* Rs = reinterpret_cast<float*>(Os);  // [2][VR][BN]
    if (t < NTH) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Rs[vrow0 * BN + vcol * 8 + j] = s1[j];
        Rs[VR * BN + vrow0 * BN + vcol * 8 + j] = s2[j];
      }
    }
  }
  __syncthreads();
  if (t < 2 * BN) {
    const int which = t / BN, c = t - which * BN;
    float s, sg;
    if constexpr (NV == 8 || NV == 4) {
      s = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] + red[(2 * 2 + which) * BN + c] + red[(3 * 2 + which) * BN + c];
      sg = red[0 * BN + c] + red[2 * BN + c] + red[4 * BN + c] + red[6 * BN + c];
    } else {
      const float* Rs = reinterpret_cast<const float*>(Os);
      float p[VR], q[VR];
#pragma unroll
      for (int r = 0; r < VR; ++r) { p[r] = Rs[which * VR * BN + r * BN + c]; q[r] = Rs[r * BN + c]; }
      s = 0.f; sg = 0.f;
#pragma unroll
      for (int r = 0; r < VR; ++r) { s += p[r]; sg += q[r]; }
    }
    if (n0 + c < N) {
      if (EPI == 2 && which == 1) {   // sum g*z -> sum g*xhat = invstd * (sum g*z - mean * sum g)
        const float mu = ecoef[2 * BN + c], is = ecoef[3 * BN + c];   // kept from the prologue
        s = is * (s - mu * sg);
      }
      atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + (size_t)which * N + n0 + c, s);
    }
  }
  SPB_TSR(4);
  // ---- output stores, last
  if (colok) {
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int r = vrow0 + s * VR;
      const int m = m0 + r;
      if (r < OBM && m < M) *reinterpret_cast<uint4*>(Yg + (size_t)m * ldc + nE) = outv[s];
    }
  }
  SPB_TSR(5);
  SPB_TS_FLUSH;
}

template <int PRO, int EPI, int NF>
int launch_os(const spb_gemm_args_t& g, hipStream_t stream) {
  constexpr int BN = 16 * NF, LDO = BN + 8, NA = PRO == 2 ? 2 : 1;
  constexpr int OSB = (OBM * LDO * 2 > 2 * 21 * BN * 4 ? OBM * LDO * 2 : 2 * 21 * BN * 4);
  const int KC = (g.K + 31) / 32, Kp = KC * 32;
  const int NT = (g.N + BN - 1) / BN, MT = (g.M + OBM - 1) / OBM;
  const size_t fixed = (size_t)(3 * Kp + 4 * BN + 8 * BN) * sizeof(float) + ((OSB + 1023) & ~1023);
  const size_t per_chunk = (size_t)NA * 4096 + (size_t)NF * 1024;
  const size_t cap = 160 * 1024;
  if (fixed + per_chunk > cap) return SPB_E_UNSUPPORTED;
  int KS = (int)((cap - fixed) / per_chunk);
  if (KS > KC) KS = KC;
  const int passes = (KC + KS - 1) / KS;
  KS = (KC + passes - 1) / passes;                           // even passes
  const size_t lds = fixed + (size_t)KS * per_chunk;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_os_kernel<PRO, EPI, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  int grid = NT * MT;
  if (grid >= 8 && (grid & 7)) grid = (grid + 7) & ~7;       // keeps the XCD remap a bijection; the extra workgroups own no rows
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_os_kernel<PRO, EPI, NF>), dim3(grid), dim3(256), (unsigned)lds, stream, nullptr, (hipEvent_t)g.stop_event, 0, g, KS);
  else
    hipLaunchKernelGGL((pw_os_kernel<PRO, EPI, NF>), dim3(grid), dim3(256), lds, stream, g, KS);
  SPB_CHECK_LAUNCH();
  return 0;
}

template <int PRO, int EPI>
int dispatch_os(const spb_gemm_args_t& g, hipStream_t stream) {
  if (g.N <= 32) return launch_os<PRO, EPI, 2>(g, stream);
  if (g.N <= 64) return launch_os<PRO, EPI, 4>(g, stream);
  return launch_os<PRO, EPI, 6>(g, stream);
}

int g_os_on = 1;          // spb_debug_set_gemm_os(0): these shapes back on the tiled / split-K kernels
int g_os_min_k = 160, g_os_max_k = 576, g_os_max_n = 96;
int g_os_min_m = 4096;    // the 7x7 maps (2352 rows at bs=48) would give 37 workgroups: they stay on the split-K kernel

}  // namespace

// bf16 only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the other kernels
int spb_gemm_os(const spb_gemm_args_t* a, hipStream_t stream) {
  if (!g_os_on || a->K < g_os_min_k || a->K > g_os_max_k || a->N > g_os_max_n || a->M < g_os_min_m || (a->K & 7) || (a->N & 7))
    return SPB_E_UNSUPPORTED;
  if (a->bias != nullptr || a->out_act != SPB_ACT_NONE || a->out_scale != 1.f) return SPB_E_UNSUPPORTED;
  if (a->pro_mode == 1 && a->epi_mode == 1) return dispatch_os<1, 1>(*a, stream);
  if (a->pro_mode == 2 && a->epi_mode == 2) return dispatch_os<2, 2>(*a, stream);
  return SPB_E_UNSUPPORTED;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_os(int on, int min_k, int max_n, int min_m) {
  g_os_on = on;
  if (min_k > 0) g_os_min_k = min_k;
  if (max_n > 0) g_os_max_n = max_n > 96 ? 96 : max_n;
  if (min_m > 0) g_os_min_m = min_m;
  return 0;
}
#endif
