// Pointwise (1x1) convolution GEMMs with a WIDE output behind a LONG reduction on the 7x7 maps: the three ConvDw extras of
// KeypointRegressionNet (reference park2019.py:32-58,114-117: Conv2d(320|1024|1280, 1024, 1) + BatchNorm2d + ReLU at M = B*49
// rows) and RevGrad's domain classifier (revgrad.py:76).  Same contract as spb_pwconv_gemm (gemm_pw.hip):
//
//   forward   Y[M,N]  = act(bn(A))[M,K] * W[N,K]^T        + sum(y), sum(y^2)                     (K = 320..1280, N = 1024 | 1280)
//   dgrad     dA[M,N] = bn_bwd(G,Z)[M,K] * Wt[N,K]^T      + mask, sum(g), sum(g*xhat)             (K = 1024, N = 320..1280)
//
// These six launches are the only matrix-core-bound GEMMs of the KRN step (5-6 GFLOP each).  The tiled kernel runs them in
// 64 x 64 tiles with a register prefetch two chunks deep and two barriers per 64-deep chunk: 25-43 us, 150-200 TFLOP/s, the
// K loop bound by one memory round trip per chunk.  Here:
//   * 128 x 128 tiles, eight waves as 2 x 4, a wave owns 64 x 32 = 4 x 2 MFMA tiles;
//   * operands go global -> LDS by LDS-DMA (no VGPRs) into a ring of three stages, counted s_waitcnt; the 16-byte slots of a
//     row are XOR-swizzled with (row & 7) on the SOURCE address, so the lane-linear DMA image is read back conflict-free;
//   * the BatchNorm(+activation) / BatchNorm-backward transform runs ONCE per element, in place in the landed A tile (each
//     wave the rows it fetched itself, so it waits for its own DMA only), not per fragment read: the MFMAs stay fed by plain
//     ds_read_b128; the transform of stage kt+1 is issued behind the matrix steps of stage kt and one barrier closes the stage;
//   * epilogue through LDS for 16-byte stores; batch sums per workgroup, one f32 atomic per channel and workgroup.
// bf16 only (the f32 parity mode keeps the tiled kernel).
#include "common.h"
#include <hip/hip_ext.h>

#ifndef SPB_TS_DECL       // register-held phase timestamps (scratch/ubench_gemm2.hip); compiled out in the product build
#define SPB_TS_DECL
#define SPB_TSR(i)
#define SPB_TS_FLUSH
#endif

namespace {

constexpr int GB = 128, GBK = 64;
constexpr int GB_TILE = GB * GBK * 2;   // bytes of one 128 x 64 bf16 operand tile

// NW = 8 waves (512 threads, 2 x 4, a wave owns 64 x 32): two waves per SIMD, so one wave's DMA issue / transform / barrier
// wait overlaps the other's matrix steps (with 4 waves every phase of a stage is serial on its SIMD)
constexpr int NW = 8, NTH = NW * 64, WCOLS = 4, WJ = GB / WCOLS / 16;   // waves across the columns, MFMA column tiles per wave
template <int PRO, int EPI>
__global__ __launch_bounds__(NTH) void pw_big_kernel(const spb_gemm_args_t g) {
  typedef bf16_t T;
  constexpr int NA = PRO == 2 ? 2 : 1;                 // A-side tensors (g and z for the BatchNorm-backward prologue)
  constexpr int STAGE = (NA + 1) * GB_TILE;
  constexpr int NST = 3;                               // 96 KB (forward) / 144 KB (BatchNorm-backward prologue: g and z tiles) of ring
  constexpr int DI = 16 / NW;                          // DMA instructions per tile and wave
  constexpr int IPS = (NA + 1) * DI;                   // DMA instructions per stage and wave
  constexpr int LDO = GB + 8, NV = GB / 8, VR = NTH / NV, VRI = GB / VR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SPB_TS_DECL;
  SPB_TSR(0);
  const int M = g.M, K = g.K, N = g.N;
  const int lda = g.lda > 0 ? g.lda : K, ldc = g.ldc > 0 ? g.ldc : N;
  const int Kp = (K + GBK - 1) / GBK * GBK, KT = Kp / GBK;
  float* coef = reinterpret_cast<float*>(smem);                       // [3][Kp]
  float* ecoef = coef + 3 * Kp;                                        // [2][128]
  char* ring = reinterpret_cast<char*>(ecoef + 2 * GB);
  T* Os = reinterpret_cast<T*>(ring);                                  // [128][LDO], after the K loop
  float* Rs = reinterpret_cast<float*>(ring + 36864);                  // [2][VR][128] statistics scratch (32 KB)

  const int t = threadIdx.x, l = t & 63, li = l & 15, lq = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WCOLS, wn = w % WCOLS;
  const int NT = (N + GB - 1) / GB;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (lid % NT) * GB, m0 = (lid / NT) * GB;

  const T* Ag = reinterpret_cast<const T*>(g.A);
  const T* A2g = (PRO == 2 && g.A2) ? reinterpret_cast<const T*>(g.A2) : Ag;   // identity prologue: p1 == 0
  const T* Bg = reinterpret_cast<const T*>(g.Bw);
  T* Yg = reinterpret_cast<T*>(g.Y);
  const T* Rg = reinterpret_cast<const T*>(g.res);
  const T* Zg = reinterpret_cast<const T*>(g.Zout);

  // DMA lane roles: instruction i of wave w fills rows w*32 + i*8 .. +7 of a tile (8 lanes = the 8 slots of a 128-byte row);
  // slot (l & 7) holds k-vector (l & 7) ^ (row & 7), and row & 7 == l >> 3
  const int dkv = (l & 7) ^ (l >> 3);
  size_t arow[DI], brow[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int r = w * (DI * 8) + i * 8 + (l >> 3);
    const int m = m0 + r, n = n0 + r;
    arow[i] = (size_t)(m < M ? m : M - 1) * lda;
    brow[i] = (size_t)(n < N ? n : N - 1) * K;
  }
  const unsigned ring_lds = lds_addr(ring);
  const unsigned wave_ro = __builtin_amdgcn_readfirstlane((unsigned)(w * DI * 1024));
  auto issue = [&](int kt) {
    const unsigned sb = ring_lds + (unsigned)((kt % NST) * STAGE) + wave_ro;
    const int k = kt * GBK + dkv * 8;
    const int kc = k < K ? k : K - 8;
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      dma16(Ag + arow[i] + kc, sb + (unsigned)(i << 10));
      if (PRO == 2) dma16(A2g + arow[i] + kc, sb + (unsigned)(GB_TILE + (i << 10)));
      dma16(Bg + brow[i] + kc, sb + (unsigned)(NA * GB_TILE + (i << 10)));
    }
  };
  for (int s = 0; s < NST - 1 && s < KT; ++s) issue(s);

  // ---- coefficient tables and the output-side operands of the dgrad epilogue share the round trip of the first stages
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < N;
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
  if (t < 256) bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);   // the table routine is written for 256 threads
  if (EPI == 2) {
    if (t < GB) {
      float sc = 1.f, sh = 0.f;
      if (n0 + t < N && g.epi.gamma != nullptr) {
        float mu, is;
        bn_moments(g.epi, n0 + t, mu, is);
        sc = g.epi.gamma[n0 + t] * is;
        sh = g.epi.beta[n0 + t] - mu * sc;
      }
      ecoef[t] = sc; ecoef[GB + t] = sh;
    }
  }
  float e_bias[8];
  if (EPI == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e_bias[j] = (colok && g.bias) ? g.bias[nE + j] : 0.f;
  }
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
  // In-place transform of the A tile of stage kt: a wave transforms exactly the rows it fetched itself (rows w*DI*8 .. +DI*8-1:
  // lane l, instruction i -> row w*DI*8 + i*8 + (l >> 3), slot l & 7, the slot its own DMA lane wrote), so it only has to wait
  // for its OWN DMA (vmcnt), not for a workgroup barrier, before it starts.
  auto transform = [&](int kt) {
    char* sb = ring + (size_t)(kt % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int row = w * (DI * 8) + i * 8 + (l >> 3), ps = l & 7;
      const int kb = kt * GBK + (dkv << 3);
      char* p = sb + row * (GBK * 2) + (ps << 4);
      Raw8<T> ar; ar.u = *reinterpret_cast<const uint4*>(p);
      float a[8], x[8];
      cvt8(ar, a);
      const float4 c0a = *reinterpret_cast<const float4*>(coef + kb), c0b = *reinterpret_cast<const float4*>(coef + kb + 4);
      const float4 c1a = *reinterpret_cast<const float4*>(coef + Kp + kb), c1b = *reinterpret_cast<const float4*>(coef + Kp + kb + 4);
      const float c0[8] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y, c0b.z, c0b.w};
      const float c1[8] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y, c1b.z, c1b.w};
      if (PRO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = a[j] * c0[j] + c1[j];
          x[j] = __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f);
        }
      } else {
        Raw8<T> a2r; a2r.u = *reinterpret_cast<const uint4*>(p + GB_TILE);
        float a2[8];
        cvt8(a2r, a2);
        const float4 c2a = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb), c2b = *reinterpret_cast<const float4*>(coef + 2 * Kp + kb + 4);
        const float c2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[j] * c0[j] + a2[j] * c1[j] + c2[j];
      }
      uint4 pa;
      pa.x = pack_bf16x2(x[0], x[1]); pa.y = pack_bf16x2(x[2], x[3]); pa.z = pack_bf16x2(x[4], x[5]); pa.w = pack_bf16x2(x[6], x[7]);
      if (kb >= K) pa = make_uint4(0, 0, 0, 0);    // reduction padding: explicit zeros against clamped (finite) weights
      *reinterpret_cast<uint4*>(p) = pa;
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stages 0 .. NST-2, the tables and the epilogue operands have landed
  __syncthreads();                                    // coefficient tables visible
  transform(0);
  __syncthreads();                                    // stage 0 transformed and visible
  SPB_TSR(1);

  f32x4_t acc[4][WJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // Software pipeline, ONE barrier per stage.  Entering iteration kt: stage kt is transformed and visible to everyone, stage
  // kt+1 is in flight or landed (raw), buffer (kt+2) % 3 == (kt-1) % 3 is free (the barrier closing iteration kt-1).
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + NST - 1 < KT;
    if (more) issue(kt + NST - 1);
    const char* at = ring + (size_t)(kt % NST) * STAGE;
    const char* bt = at + NA * GB_TILE;
#pragma unroll
    for (int ks = 0; ks < GBK / 32; ++ks) {
      const int v = ks * 4 + lq;
      bf16x8_t af[4], bfv[WJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + li;
        af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(at + r * (GBK * 2) + ((v ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int r = wn * (WJ * 16) + j * 16 + li;
        bfv[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bt + r * (GBK * 2) + ((v ^ (r & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = SPB_MFMA16(af[i], bfv[j], acc[i][j]);
    }
    if (kt + 1 < KT) {
      // this wave's own share of stage kt+1 has landed once only the stage issued above is still in flight; its transform
      // (LDS + vector ALU) runs while the matrix cores work through the steps issued above
      if (more) wait_vmcnt<IPS>(); else wait_vmcnt<0>();
      transform(kt + 1);
    }
    __builtin_amdgcn_s_barrier();   // stage kt+1 transformed and visible; everyone is done reading stage kt
    asm volatile("" ::: "memory");
  }
  __syncthreads();   // the ring is idle (the last stages were waited with vmcnt(0)): reuse it
  SPB_TSR(2);

  // ---- accumulators -> LDS (C layout: col = lane & 15, row = (lane >> 4) * 4 + r), then the coalesced 16-byte epilogue
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Os[(wm * 64 + i * 16 + lq * 4 + r) * LDO + wn * (WJ * 16) + j * 16 + li] = f2bf(acc[i][j][r]);
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (colok) {
    float e_sc[8], e_sh[8];
    if (EPI == 2) {
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        *reinterpret_cast<float4*>(e_sc + j) = *reinterpret_cast<const float4*>(ecoef + vcol * 8 + j);
        *reinterpret_cast<float4*>(e_sh + j) = *reinterpret_cast<const float4*>(ecoef + GB + vcol * 8 + j);
      }
    }
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
          Raw8<T> zq; zq.u = zr[EPI == 2 ? s : 0];
          float z[8];
          cvt8(zq, z);
          if (Rg) {
            Raw8<T> rq; rq.u = rr[EPI == 2 ? s : 0];
            float rv[8];
            cvt8(rq, rv);
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
  SPB_TSR(3);
  if (EPI != 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Rs[vrow0 * GB + vcol * 8 + j] = s1[j];
      Rs[VR * GB + vrow0 * GB + vcol * 8 + j] = s2[j];
    }
    __syncthreads();
    const int which = t / GB, c = t % GB;    // the first 256 threads: 2 x 128 channels
    float s = 0.f, sg = 0.f;
    if (t < 2 * GB) {
#pragma unroll
      for (int r = 0; r < VR; ++r) { s += Rs[which * VR * GB + r * GB + c]; sg += Rs[r * GB + c]; }
    }
    if (t < 2 * GB && n0 + c < N) {
      if (EPI == 2 && which == 1) {   // sum g*z -> sum g*xhat = invstd * (sum g*z - mean * sum g)
        float mu = 0.f, is = 0.f;
        if (g.epi.gamma != nullptr) bn_moments(g.epi, n0 + c, mu, is);
        s = is * (s - mu * sg);
      }
      atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + (size_t)which * N + n0 + c, s);
    }
  }
  SPB_TSR(4);
  SPB_TS_FLUSH;
}

template <int PRO, int EPI>
int launch_big(const spb_gemm_args_t& g, hipStream_t stream) {
  constexpr int NA = PRO == 2 ? 2 : 1, NST = 3;
  const int NT = (g.N + GB - 1) / GB, MT = (g.M + GB - 1) / GB;
  const int Kp = (g.K + GBK - 1) / GBK * GBK;
  const size_t lds = (size_t)(3 * Kp + 2 * GB) * sizeof(float) + (size_t)NST * (NA + 1) * GB_TILE;
  if (lds > 160 * 1024) return SPB_E_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_big_kernel<PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  int grid = NT * MT;
  if (grid >= 8 && (grid & 7)) grid = (grid + 7) & ~7;     // keeps the XCD remap a bijection; the extra workgroups own no rows
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_big_kernel<PRO, EPI>), dim3(grid), dim3(NTH), (unsigned)lds, stream, nullptr, (hipEvent_t)g.stop_event, 0, g);
  else
    hipLaunchKernelGGL((pw_big_kernel<PRO, EPI>), dim3(grid), dim3(NTH), lds, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

int g_big_on = 1, g_big_min_n = 512, g_big_min_k = 256, g_big_max_m = 16384;

}  // namespace

// bf16 only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the other kernels
int spb_gemm_big(const spb_gemm_args_t* a, hipStream_t stream) {
  if (!g_big_on || a->N < g_big_min_n || a->K < g_big_min_k || a->M > g_big_max_m || (a->K & 7) || (a->N & 7)) return SPB_E_UNSUPPORTED;
  if (a->pro_mode == 1 && a->epi_mode == 1) return launch_big<1, 1>(*a, stream);
  if (a->pro_mode == 1 && a->epi_mode == 0) return launch_big<1, 0>(*a, stream);
  if (a->pro_mode == 2 && a->epi_mode == 2) return launch_big<2, 2>(*a, stream);
  return SPB_E_UNSUPPORTED;
}

extern "C" int spb_debug_set_gemm_big(int on, int min_n, int min_k) {
  g_big_on = on;
  if (min_n > 0) g_big_min_n = min_n;
  if (min_k > 0) g_big_min_k = min_k;
  return 0;
}
