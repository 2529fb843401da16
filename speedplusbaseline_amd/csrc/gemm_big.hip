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
#include <type_traits>

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
// ACTK: prologue activation of the forward transform -- 0 none (affine only), 1 clamp (ReLU / ReLU6), 2 generic (leaky)
template <int PRO, int EPI, int ACTK>
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
  float* ecoef = coef + 3 * Kp;                                        // [4][128] scale, shift, mean, inverse std of the output columns
  char* ring = reinterpret_cast<char*>(ecoef + 4 * GB);
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
  unsigned aoff[DI], boff[DI];          // byte offsets of this lane's 16 bytes inside stage 0 (K % 64 == 0: no clamp along k)
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int r = w * (DI * 8) + i * 8 + (l >> 3);
    const int m = m0 + r, n = n0 + r;
    aoff[i] = (unsigned)(((size_t)(m < M ? m : M - 1) * lda + dkv * 8) * 2);
    boff[i] = (unsigned)(((size_t)(n < N ? n : N - 1) * K + dkv * 8) * 2);
  }
  const unsigned ring_lds = lds_addr(ring);
  const unsigned wave_ro = __builtin_amdgcn_readfirstlane((unsigned)(w * DI * 1024));
  // stage kt -> ring buffer `buf` (compile-time in the main loop).  The sources advance by 128 bytes per stage on the SCALAR unit.
  auto issue = [&](int kt, int buf) {
    const unsigned sb = ring_lds + (unsigned)(buf * STAGE) + wave_ro;
    const char* sa = reinterpret_cast<const char*>(Ag) + (size_t)kt * (GBK * 2);
    const char* sa2 = reinterpret_cast<const char*>(A2g) + (size_t)kt * (GBK * 2);
    const char* sw = reinterpret_cast<const char*>(Bg) + (size_t)kt * (GBK * 2);
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      dma16s(sa, aoff[i], sb + (unsigned)(i << 10));
      if (PRO == 2) dma16s(sa2, aoff[i], sb + (unsigned)(GB_TILE + (i << 10)));
      dma16s(sw, boff[i], sb + (unsigned)(NA * GB_TILE + (i << 10)));
    }
  };
  for (int s = 0; s < NST && s < KT; ++s) issue(s, s);

  // ---- coefficient tables and the output-side operands of the dgrad epilogue share the round trip of the first stages
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < N;
  if (t < 256) bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, Kp, coef, t);   // the table routine is written for 256 threads
  if (EPI == 2) {
    if (t >= 256 && t < 256 + GB) {      // waves 4-5: in parallel with the table of waves 0-3 (the same threads did one after the other)
      const int c = t - 256;
      float sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
      if (n0 + c < N && g.epi.gamma != nullptr) {
        const float gm = g.epi.gamma[n0 + c], bt = g.epi.beta[n0 + c];
        bn_moments(g.epi, n0 + c, mu, is);
        sc = gm * is;
        sh = bt - mu * sc;
      }
      ecoef[c] = sc; ecoef[GB + c] = sh; ecoef[2 * GB + c] = mu; ecoef[3 * GB + c] = is;   // mean / inverse std: for the sums at the very end
    }
  }
  float e_bias[8];
  if (EPI == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) e_bias[j] = (colok && g.bias) ? g.bias[nE + j] : 0.f;
  }
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
  // In-place transform of the A tile of a stage: a wave transforms exactly the rows it fetched itself (rows w*DI*8 .. +DI*8-1: lane l,
  // instruction i -> row w*DI*8 + i*8 + (l >> 3), slot l & 7, the slot its own DMA lane wrote), so it only has to wait for its OWN DMA
  // (vmcnt), not for a workgroup barrier, before it starts.  Split in two: every LDS read first (raw operands + coefficients), arithmetic
  // and stores later -- the row-by-row form finishes row 0's store before row 1's loads (they may alias).
  const unsigned xrow = (unsigned)((w * (DI * 8) + (l >> 3)) * (GBK * 2) + ((l & 7) << 4));      // byte offset of this lane's slot, row i = 0
  struct XfRegs { uint4 raw[DI], raw2[DI]; float4 cf[6]; };
  auto xf_load = [&](int kt, int buf, XfRegs& x) {
    const char* sb = ring + (size_t)buf * STAGE + xrow;
    const float* cp = coef + kt * GBK + (dkv << 3);
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      x.raw[i] = *reinterpret_cast<const uint4*>(sb + i * 8 * (GBK * 2));
      if (PRO == 2) x.raw2[i] = *reinterpret_cast<const uint4*>(sb + GB_TILE + i * 8 * (GBK * 2));
    }
#pragma unroll
    for (int c = 0; c < (PRO == 2 ? 3 : 2); ++c) {
      x.cf[2 * c] = *reinterpret_cast<const float4*>(cp + c * Kp);
      x.cf[2 * c + 1] = *reinterpret_cast<const float4*>(cp + c * Kp + 4);
    }
  };
  auto xf_store = [&](int buf, const XfRegs& x) {
    char* sb = ring + (size_t)buf * STAGE + xrow;
    const float c0[8] = {x.cf[0].x, x.cf[0].y, x.cf[0].z, x.cf[0].w, x.cf[1].x, x.cf[1].y, x.cf[1].z, x.cf[1].w};
    const float c1[8] = {x.cf[2].x, x.cf[2].y, x.cf[2].z, x.cf[2].w, x.cf[3].x, x.cf[3].y, x.cf[3].z, x.cf[3].w};
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      Raw8<T> ar; ar.u = x.raw[i];
      float a[8], v[8];
      cvt8(ar, a);
      if (PRO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = a[j] * c0[j] + c1[j];
          v[j] = ACTK == 0 ? u : (ACTK == 1 ? __builtin_amdgcn_fmed3f(u, 0.f, act_h) : __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f));
        }
      } else {
        Raw8<T> a2r; a2r.u = x.raw2[i];
        float a2[8];
        cvt8(a2r, a2);
        const float c2[8] = {x.cf[4].x, x.cf[4].y, x.cf[4].z, x.cf[4].w, x.cf[5].x, x.cf[5].y, x.cf[5].z, x.cf[5].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a[j] * c0[j] + a2[j] * c1[j] + c2[j];
      }
      uint4 pa;
      pa.x = pack_bf16x2(v[0], v[1]); pa.y = pack_bf16x2(v[2], v[3]); pa.z = pack_bf16x2(v[4], v[5]); pa.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(sb + i * 8 * (GBK * 2)) = pa;
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stages 0 .. NST-1, the tables and the epilogue operands have landed
  __syncthreads();                                    // coefficient tables visible
  {
    XfRegs x0, x1;
    xf_load(0, 0, x0);
    if (KT > 1) xf_load(1, 1, x1);
    xf_store(0, x0);
    if (KT > 1) xf_store(1, x1);
  }
  __syncthreads();                                    // stages 0 and 1 transformed and visible
  SPB_TSR(1);

  f32x4_t acc[4][WJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // MFMA operands of a stage: 2 reduction steps x (4 row tiles + WJ column tiles), held in registers for a whole iteration
  struct Frags { bf16x8_t a[2][4], b[2][WJ]; };
  unsigned fa[2][4], fb[2][WJ];        // byte offsets of this lane's operand vectors inside a stage
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int v = ks * 4 + lq;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 64 + i * 16 + li; fa[ks][i] = (unsigned)(r * (GBK * 2) + ((v ^ (r & 7)) << 4)); }
#pragma unroll
    for (int j = 0; j < WJ; ++j) { const int r = wn * (WJ * 16) + j * 16 + li; fb[ks][j] = (unsigned)(NA * GB_TILE + r * (GBK * 2) + ((v ^ (r & 7)) << 4)); }
  }
  auto fetch = [&](int buf, Frags& f) {
    const char* sb = ring + (size_t)buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f.a[ks][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + fa[ks][i]));
#pragma unroll
      for (int j = 0; j < WJ; ++j) f.b[ks][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + fb[ks][j]));
    }
  };
  // Software pipeline over THREE stages, one barrier per iteration.  Entering iteration kt: the operands of stage kt sit in registers (fetched
  // during iteration kt-1), stage kt+1 is transformed and visible, stage kt+2 is raw (landed or in flight), and the buffer of stage kt is
  // free -- every wave finished reading it before the barrier that closed iteration kt-1 -- so the DMA of stage kt+3 goes there.
  // What bounds the loop is the VECTOR-INSTRUCTION COUNT of an iteration, not its order (round 4: 240 instructions per wave and 64-deep
  // stage, 150 of them vector ALU, at 2 waves per SIMD and ~4 cycles each = the 0.93 us per stage measured whatever the order: transform
  // before / after / between the matrix steps, waves of a SIMD in opposite phase, operands prefetched into registers -- all within 5 %).
  // Hence: ring buffers are compile-time (the loop is unrolled by three: no address arithmetic for 20 LDS accesses), DMA sources advance
  // on the scalar unit, the activation is a template parameter, K % 64 == 0 (no padding test).
  Frags cur, nxt;
  fetch(0, cur);
  lds_barrier();                                      // every wave holds the operands of stage 0: its buffer is free
  auto iteration = [&](int kt, auto b_, auto f_dma, auto f_fetch, auto f_xf) {
    constexpr int BUF = decltype(b_)::value;          // == kt % 3
    constexpr bool DMA = decltype(f_dma)::value, FETCH = decltype(f_fetch)::value, XF = decltype(f_xf)::value;
    XfRegs xr;
    if (DMA) issue(kt + NST, BUF);
    if (XF) {
      // this wave's own share of stage kt+2 (requested one iteration ago) has landed once only the stage issued above is still in flight
      if (DMA) wait_vmcnt<IPS>(); else wait_vmcnt<0>();
      xf_load(kt + 2, (BUF + 2) % NST, xr);
    }
    if (FETCH) fetch((BUF + 1) % NST, nxt);
#pragma unroll
    for (int ks = 0; ks < GBK / 32; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j] = SPB_MFMA16(cur.a[ks][i], cur.b[ks][j], acc[i][j]);
    if (XF) xf_store((BUF + 2) % NST, xr);
    lds_barrier();   // stage kt+2 transformed and visible; every wave holds the operands of stage kt+1
    if (FETCH) cur = nxt;
  };
  typedef std::true_type Y_;
  typedef std::false_type N_;
  typedef std::integral_constant<int, 0> B0;
  typedef std::integral_constant<int, 1> B1;
  typedef std::integral_constant<int, 2> B2;
  // the last three iterations (no DMA; no fetch / transform beyond the last stage), starting at ring phase P
  auto tail = [&](int kt, auto p_) {
    constexpr int P = decltype(p_)::value;
    iteration(kt, std::integral_constant<int, P>(), N_(), Y_(), Y_());
    iteration(kt + 1, std::integral_constant<int, (P + 1) % 3>(), N_(), Y_(), N_());
    iteration(kt + 2, std::integral_constant<int, (P + 2) % 3>(), N_(), N_(), N_());
  };
  int kt = 0;
  for (; kt + 3 <= KT - NST; kt += 3) {               // KT >= 4 (K >= 256)
    iteration(kt, B0(), Y_(), Y_(), Y_());
    iteration(kt + 1, B1(), Y_(), Y_(), Y_());
    iteration(kt + 2, B2(), Y_(), Y_(), Y_());
  }
  const int left = KT - NST - kt;                     // 0..2 steady iterations left; kt is a multiple of 3
  if (left >= 1) iteration(kt, B0(), Y_(), Y_(), Y_());
  if (left == 2) iteration(kt + 1, B1(), Y_(), Y_(), Y_());
  kt += left;
  if (left == 0) tail(kt, B0()); else if (left == 1) tail(kt, B1()); else tail(kt, B2());
  // the output-side operands of the dgrad epilogue: requested here, used after the accumulators went through LDS (held across the K loop
  // they were 32 registers of a kernel that has none to spare: the loop spilled)
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
  lds_barrier();   // the ring is idle (every DMA was waited for inside the loop): reuse it
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
        const float mu = ecoef[2 * GB + c], is = ecoef[3 * GB + c];   // kept from the prologue (loading the sums again here put a memory
        s = is * (s - mu * sg);                                       // round trip between the last store and the end of the kernel)
      }
      atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + (size_t)which * N + n0 + c, s);
    }
  }
  SPB_TSR(4);
  SPB_TS_FLUSH;
}

template <int PRO, int EPI, int ACTK>
int launch_big(const spb_gemm_args_t& g, hipStream_t stream) {
  constexpr int NA = PRO == 2 ? 2 : 1, NST = 3;
  const int NT = (g.N + GB - 1) / GB, MT = (g.M + GB - 1) / GB;
  const int Kp = (g.K + GBK - 1) / GBK * GBK;
  const size_t lds = (size_t)(3 * Kp + 4 * GB) * sizeof(float) + (size_t)NST * (NA + 1) * GB_TILE;
  if (lds > 160 * 1024) return SPB_E_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_big_kernel<PRO, EPI, ACTK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  int grid = NT * MT;
  if (grid >= 8 && (grid & 7)) grid = (grid + 7) & ~7;     // keeps the XCD remap a bijection; the extra workgroups own no rows
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_big_kernel<PRO, EPI, ACTK>), dim3(grid), dim3(NTH), (unsigned)lds, stream, nullptr, (hipEvent_t)g.stop_event, 0, g);
  else
    hipLaunchKernelGGL((pw_big_kernel<PRO, EPI, ACTK>), dim3(grid), dim3(NTH), lds, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

int g_big_on = 1, g_big_min_n = 512, g_big_min_k = 256, g_big_max_m = 16384;

}  // namespace

// bf16 only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the other kernels
int spb_gemm_big(const spb_gemm_args_t* a, hipStream_t stream) {
  if (!g_big_on || a->N < g_big_min_n || a->K < g_big_min_k || a->M > g_big_max_m || (a->K & 7) || (a->N & 7)) return SPB_E_UNSUPPORTED;
  if ((a->K & 63) || a->K < 256) return SPB_E_UNSUPPORTED;         // whole 64-deep stages, at least four of them
  const int actk = (a->pro.act == SPB_ACT_RELU || a->pro.act == SPB_ACT_RELU6) ? 1 : (a->pro.act == SPB_ACT_NONE ? 0 : 2);
  if (a->pro_mode == 1 && a->epi_mode == 1)
    return actk == 1 ? launch_big<1, 1, 1>(*a, stream) : (actk == 0 ? launch_big<1, 1, 0>(*a, stream) : launch_big<1, 1, 2>(*a, stream));
  if (a->pro_mode == 1 && a->epi_mode == 0)
    return actk == 1 ? launch_big<1, 0, 1>(*a, stream) : (actk == 0 ? launch_big<1, 0, 0>(*a, stream) : launch_big<1, 0, 2>(*a, stream));
  if (a->pro_mode == 2 && a->epi_mode == 2) return launch_big<2, 2, 0>(*a, stream);
  return SPB_E_UNSUPPORTED;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_big(int on, int min_n, int min_k) {
  g_big_on = on;
  if (min_n > 0) g_big_min_n = min_n;
  if (min_k > 0) g_big_min_k = min_k;
  return 0;
}
#endif
