// Pointwise (1x1) convolution FORWARD on the large maps (112x112, 56x56: M = B*H*W = 150k .. 600k rows, N*K a few thousand weights):
// "streaming" kernel.  Same contract as spb_pwconv_gemm (gemm_pw.hip):  Y[M,N] = act(bn(A))[M,K] * W[N,K]^T  (or the residual join
// bn(A) + bn2(A2), written out as Ymat) + per-channel sum(y), sum(y^2).  Reference: torchvision MobileNetV2 expand / project convolutions of
// blocks 1-4 (park2019.py:107-108).
//
// Why (round 5).  These six launches are pure streams -- 16->96 at 112x112 writes 116 MB for 19 MB read and 0.2 GFLOP -- and the tiled kernel
// (64*RF x BN tiles, operands and results staged through LDS, a barrier per stage) moved them at 2.0-2.6 TB/s: 170 us of the 615 us of the
// forward pointwise family, the family bench.py's roofline line is quoted on.  The fused backward kernel of the same layers
// (pw_bwd_fused.hip) runs at 2.5-5.3 TB/s with waves that own their rows end to end; this is that design for the forward direction:
//   * a WAVE owns 16-row chunks end to end: no LDS tile, no barrier in the loop.  Its operand fragments come straight from global memory in
//     the matrix-core layout (lane (li, lq): row li, 8 consecutive input channels at lq*8 of each 32-deep step -- one 16-byte load), two
//     chunks ahead in registers; BatchNorm + activation (or the join) run on the fragment in registers;
//   * the product is formed transposed, W * a^T, with the weight rows permuted so that a lane ends up with 8 CONSECUTIVE output channels
//     of its row per fragment pair (gemm_rs.hip): results leave as 16-byte stores, the four lq lanes of a row cover 64 contiguous bytes;
//   * the weights of the wave's NW = 16*NJ output channels stay in registers for the whole launch.  Wide outputs are split over
//     WORKGROUPS (N = 96: two column splits of 48, N = 144: three): the rows are read once per split -- they are 1/6 of the bytes;
//   * the per-channel sums live in registers across all chunks of a wave and leave once: DPP row reduction, LDS across the four waves,
//     one f32 atomic per channel and workgroup.
// Measured (bs=48): 32->16 at 112x112 21.6 -> 17.0 us, 16->96 52.2 -> 35.2 us (3.8 TB/s), 24->144 at 56x56 23.8 -> 20.0 us, the same with the
// residual join 33.3 -> 23.5 us; the KRN step 2.665 -> 2.62 ms.  With the column splits of a row range spread over the XCDs (the first version)
// 16->96 took 41.9 us: every L2 fetched the rows again.
// 16-bit storage only (bf16, or IEEE half in the -DSPB_F16 twin); the f32 parity mode keeps the tiled kernel.
#include "common.h"
#include <hip/hip_ext.h>

namespace {

// PRO 1: a = act(bn(A));  3: a = bn(A) + bn2(A2), the first column split writes Ymat.  NJ = 16-channel fragments per wave, KS = 32-deep steps
// SO (round 6): STATISTICS ONLY -- g.Y == NULL: nothing is stored, the per-channel sums see the f32 accumulators.  This is the pass that
// is left of an expand convolution whose output is never written (spb_dw_args_t::Xe: the depthwise kernels recompute it per pixel):
// it reads the 16 / 24-channel block input (once from HBM; the column splits of a row range share an XCD) and costs a sixth of the
// bytes.  The first column split also writes the convolution's OPERAND round16(act(bn(A))) to Ymat when one is given (PRO 3: the residual
// join, always): the recomputing kernels then read it as it is instead of redoing the BatchNorm per use (measured: with the affine in
// their staging loops the depthwise forward kernels were 10 us slower than unfused; each input element is staged 8-10 times there).
// The pass is bound by vector-instruction issue, not by bytes in flight: ~90 wave instructions per 16-row chunk and column split (a wave64
// instruction holds a SIMD for 4 cycles; with 16 input channels half of the lanes of every operand instruction carry padding).  Measured:
// a six-chunk register ring instead of two chunks in flight made the 112x112 launch SLOWER (23 -> 30 us: 154 registers, two waves per
// SIMD); one column split of 96 channels instead of two of 48 halves the operand work of that launch.
template <int PRO, int NJ, int KS, bool SO = false>
__global__ __launch_bounds__(256) void pw_st_kernel(const spb_gemm_args_t g, int nsplit, long long nchunks) {
  constexpr int KP = 32 * KS, NW = 16 * NJ, NP = NJ / 2;
  __shared__ float coef[3 * KP];
  __shared__ float sred[4][2 * NW];
  const int t = threadIdx.x, l = t & 63, li = l & 15, lq = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = g.M, K = g.K, N = g.N;
  // workgroup b runs on XCD b % 8: the nsplit column splits of one row range sit on ONE XCD, 8 apart in b (dispatched back to back), so the
  // rows they all read cross HBM once and are L2 hits for the other splits
  const int xcd = blockIdx.x & 7, bq = blockIdx.x >> 3;
  const int split = bq % nsplit, wgi = xcd + 8 * (bq / nsplit), nwg = gridDim.x / nsplit;
  const int c_base = split * NW;
  const bf16_t* Ag = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* A2g = reinterpret_cast<const bf16_t*>(PRO == 3 ? g.A2 : g.A);
  const bf16_t* Bg = reinterpret_cast<const bf16_t*>(g.Bw);
  bf16_t* Yg = reinterpret_cast<bf16_t*>(g.Y);
  bf16_t* Ymat = reinterpret_cast<bf16_t*>(g.Ymat);

  // the wave's weight fragments (A operand of W * a^T), rows permuted as in gemm_rs.hip: row p' of fragments 2p, 2p+1 is output channel
  // c_base + p*32 + (p' >> 2)*8 + {0, 4} + (p' & 3); an odd last fragment holds 16 channels on its own (4 per lane)
  auto chan = [&](int j, int pr) {
    if ((NJ & 1) && j == NJ - 1) return c_base + (NJ - 1) * 16 + pr;
    return c_base + (j >> 1) * 32 + (pr >> 2) * 8 + (j & 1) * 4 + (pr & 3);
  };
  uint4 wf[NJ][KS];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = chan(j, li);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + lq * 8;
      uint4 u = *reinterpret_cast<const uint4*>(Bg + (size_t)(n < N ? n : N - 1) * K + (k < K ? k : K - 8));
      if (n >= N || k >= K) u = make_uint4(0, 0, 0, 0);
      wf[j][ks] = u;
    }
  }
  if constexpr (PRO == 3) bn_join_table(g.pro, g.pro2, K, KP, coef, t);
  else bn_coef_table<1>(g.pro, K, KP, coef, t);

  // chunks of this wave: (wgi*4 + w), + nwg*4, ...
  const long long cstep = (long long)nwg * 4;
  long long c = (long long)wgi * 4 + w;
  uint4 ra0[KS], ra1[KS], rb0[PRO == 3 ? KS : 1], rb1[PRO == 3 ? KS : 1];
  auto load = [&](long long cc, uint4 (&ra)[KS], uint4 (&rb)[PRO == 3 ? KS : 1]) __attribute__((always_inline)) {
    long long m = cc * 16 + li;
    m = m < M ? m : M - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + lq * 8;
      const size_t o = (size_t)m * K + (k < K ? k : K - 8);
      ra[ks] = *reinterpret_cast<const uint4*>(Ag + o);
      if constexpr (PRO == 3) rb[ks] = *reinterpret_cast<const uint4*>(A2g + o);
    }
  };
  load(c < nchunks ? c : 0, ra0, rb0);
  if constexpr (KS < 3) load(c + cstep < nchunks ? c + cstep : 0, ra1, rb1);
  __syncthreads();      // the coefficient table (the only barrier of the launch besides the final reduction)

  float s1[NJ][4], s2[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
  const bool act_none = g.pro.act == SPB_ACT_NONE;

  // the B fragment of reduction step ks of a chunk from the raw 16-byte vector(s): BatchNorm + activation (or the join) in registers
  auto frag = [&](int ks, long long m, bool rowok, uint4 va, uint4 vb) __attribute__((always_inline)) {
    const int k = ks * 32 + lq * 8;
    Raw8<bf16_t> r1; r1.u = va;
    float a[8], x[8];
    cvt8(r1, a);
    float c0[8], c1[8];
    *reinterpret_cast<float4*>(c0) = *reinterpret_cast<const float4*>(coef + k);
    *reinterpret_cast<float4*>(c0 + 4) = *reinterpret_cast<const float4*>(coef + k + 4);
    *reinterpret_cast<float4*>(c1) = *reinterpret_cast<const float4*>(coef + KP + k);
    *reinterpret_cast<float4*>(c1 + 4) = *reinterpret_cast<const float4*>(coef + KP + k + 4);
    if constexpr (PRO == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = a[j] * c0[j] + c1[j];
        x[j] = (SO && act_none) ? u : __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f);   // (uniform; block inputs are linear)
      }
    } else {
      Raw8<bf16_t> r2; r2.u = vb;
      float a2[8], c2[8];
      cvt8(r2, a2);
      *reinterpret_cast<float4*>(c2) = *reinterpret_cast<const float4*>(coef + 2 * KP + k);
      *reinterpret_cast<float4*>(c2 + 4) = *reinterpret_cast<const float4*>(coef + 2 * KP + k + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = a[j] * c0[j] + a2[j] * c1[j] + c2[j];
    }
    const bool ok = rowok && k < K;
    uint4 pa;
    pa.x = pack_bf16x2(x[0], x[1]); pa.y = pack_bf16x2(x[2], x[3]); pa.z = pack_bf16x2(x[4], x[5]); pa.w = pack_bf16x2(x[6], x[7]);
    if (!ok) pa = make_uint4(0, 0, 0, 0);
    if constexpr (PRO == 3) { if (ok && split == 0) *reinterpret_cast<uint4*>(Ymat + (size_t)m * K + k) = pa; }
    else if constexpr (SO) { if (ok && split == 0 && Ymat != nullptr) *reinterpret_cast<uint4*>(Ymat + (size_t)m * K + k) = pa; }
    return __builtin_bit_cast(bf16x8_t, pa);
  };
  // rounded 16-byte stores and the sums of a chunk's accumulators.  Rows past M and channels past N carry exact zeros (zeroed operands /
  // weight rows): the sums need no mask
  auto finish = [&](f32x4_t (&acc)[NJ], long long m, bool rowok) __attribute__((always_inline)) {
    if constexpr (SO) {       // rows past M / channels past N carry exact zeros: no mask
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[j][r] += acc[j][r]; s2[j][r] += acc[j][r] * acc[j][r]; }
      return;
    }
    bf16_t* yp = Yg + (size_t)(rowok ? m : M - 1) * N;
    auto fin4 = [&](const f32x4_t& av, float (&t1)[4], float (&t2)[4]) {
      uint2 o;
      o.x = pack_bf16x2(av[0], av[1]); o.y = pack_bf16x2(av[2], av[3]);
      float q[4];
      spb_unpack2(o.x, q[0], q[1]); spb_unpack2(o.y, q[2], q[3]);            // the sums see the rounded values, as every other producer's
#pragma unroll
      for (int r = 0; r < 4; ++r) { t1[r] += q[r]; t2[r] += q[r] * q[r]; }
      return o;
    };
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint2 lo = fin4(acc[2 * p], s1[2 * p], s2[2 * p]);
      const uint2 hi = fin4(acc[2 * p + 1], s1[2 * p + 1], s2[2 * p + 1]);
      const int ch = c_base + p * 32 + lq * 8;
      if (rowok && ch < N) *reinterpret_cast<uint4*>(yp + ch) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    if (NJ & 1) {
      const uint2 o = fin4(acc[NJ - 1], s1[NJ - 1], s2[NJ - 1]);
      const int ch = c_base + (NJ - 1) * 16 + lq * 4;
      if (rowok && ch < N) *reinterpret_cast<uint2*>(yp + ch) = o;
    }
  };
  // one chunk from a register buffer: 1..KS matrix steps per output fragment
  auto chunk = [&](long long cc, uint4 (&ra)[KS], uint4 (&rb)[PRO == 3 ? KS : 1]) __attribute__((always_inline)) {
    const long long m = cc * 16 + li;
    const bool rowok = m < M;
    bf16x8_t bfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfr[ks] = frag(ks, m, rowok, ra[ks], rb[PRO == 3 ? ks : 0]);
    f32x4_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[j] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wf[j][ks]), bfr[ks], acc[j]);
    }
    finish(acc, m, rowok);
  };

  if constexpr (KS >= 3) {
    // long reduction (the project convolutions 96 -> 24, 144 -> 24 at 56x56): ONE register buffer, refilled in place -- step ks of the next
    // chunk is requested the moment step ks of this one has been taken out, so KS loads stay in flight with KS registers (two whole
    // buffers + the copy cost 154 / 222 registers: 2-3 waves per SIMD, slower than the tiled kernel)
    while (c < nchunks) {
      // the 16 BatchNorm coefficients a lane needs per step are loop invariant: left alone the compiler keeps all 16 * KS of them in
      // registers (80 at KS = 5); re-read from LDS per chunk they are 4 ds_read_b128 per step
      asm volatile("" ::: "memory");
      const long long cn = c + cstep, cl = cn < nchunks ? cn : c;
      const long long m = c * 16 + li;
      const bool rowok = m < M;
      long long mn = cl * 16 + li;
      mn = mn < M ? mn : M - 1;
      f32x4_t acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 va = ra0[ks];
        const int k = ks * 32 + lq * 8;
        ra0[ks] = *reinterpret_cast<const uint4*>(Ag + (size_t)mn * K + (k < K ? k : K - 8));
        const bf16x8_t bf = frag(ks, m, rowok, va, va);
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wf[j][ks]), bf, acc[j]);
      }
      finish(acc, m, rowok);
      c = cn;
    }
  } else {
  // two chunks in flight per wave: the buffer just consumed is refilled with the chunk two steps ahead before the other one is worked on
  while (c < nchunks) {
    {
      uint4 xa[KS], xb[PRO == 3 ? KS : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { xa[ks] = ra0[ks]; if (PRO == 3) xb[PRO == 3 ? ks : 0] = rb0[PRO == 3 ? ks : 0]; }
      const long long cn = c + 2 * cstep;
      load(cn < nchunks ? cn : c, ra0, rb0);
      chunk(c, xa, xb);
    }
    c += cstep;
    if (c >= nchunks) break;
    {
      uint4 xa[KS], xb[PRO == 3 ? KS : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { xa[ks] = ra1[ks]; if (PRO == 3) xb[PRO == 3 ? ks : 0] = rb1[PRO == 3 ? ks : 0]; }
      const long long cn = c + 2 * cstep;
      load(cn < nchunks ? cn : c, ra1, rb1);
      chunk(c, xa, xb);
    }
    c += cstep;
  }
  }

  // ---- per-channel sums: over the 16 rows of a DPP row -> LDS per wave -> one f32 atomic per channel, sum and workgroup
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = row16_sum(s1[j][r]), b = row16_sum(s2[j][r]);
      if (li == ((j * 4 + r) & 15)) {
        const int cl = ((NJ & 1) && j == NJ - 1) ? (NJ - 1) * 16 + lq * 4 + r : (j >> 1) * 32 + lq * 8 + (j & 1) * 4 + r;
        sred[w][cl] = a; sred[w][NW + cl] = b;
      }
    }
  __syncthreads();
  for (int i = t; i < 2 * NW; i += 256) {
    const int which = i >= NW, cl = i - which * NW, ch = c_base + cl;
    if (ch < N) {
      const float v = (sred[0][i] + sred[1][i]) + (sred[2][i] + sred[3][i]);
      atomicAdd(g.osums + (size_t)(wgi % g.oR) * 2 * N + (size_t)which * N + ch, v);
    }
  }
}

// persistent workgroups: 384 -> 2.627 ms per KRN step, 512 -> 2.625, 640 -> 2.635, 768 -> 2.620, 1024 -> 2.634, 1280 -> 2.636 (without the kernel: 2.665)
int g_st_on = 1, g_st_min_m = 100000, g_st_wgs = 768, g_st_long_k = 1;

template <int PRO, int NJ, int KS, bool SO = false>
int launch_st(const spb_gemm_args_t& g, hipStream_t stream) {
  const int NW = 16 * NJ;
  const int nsplit = (g.N + NW - 1) / NW;
  const long long nchunks = ((long long)g.M + 15) / 16;
  long long nwg = (nchunks + 3) / 4;                    // workgroups per column split: a multiple of 8 (one share per XCD)
  const long long cap = g_st_wgs / nsplit > 8 ? g_st_wgs / nsplit : 8;
  if (nwg > cap) nwg = cap;
  nwg = (nwg + 7) & ~7LL;
  const unsigned grid = (unsigned)(nwg * nsplit);
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_st_kernel<PRO, NJ, KS, SO>), dim3(grid), dim3(256), 0, stream, nullptr, (hipEvent_t)g.stop_event, 0, g, nsplit, nchunks);
  else
    hipLaunchKernelGGL((pw_st_kernel<PRO, NJ, KS, SO>), dim3(grid), dim3(256), 0, stream, g, nsplit, nchunks);
  SPB_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// 16-bit storage only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the other kernels
int spb_gemm_st(const spb_gemm_args_t* a, hipStream_t stream) {
  if (a->Y == nullptr) {   // statistics only (any M).  Column splits of 48 as the storing form: wider splits need 160-250 registers (one or two
                           // waves per SIMD: 20-25 us per launch, measured) for rows that are L2 hits anyway
    if ((a->N % 48) || a->N > 192 || (a->K & 7) || a->K > 32 || a->epi_mode != 1 || (a->pro_mode != 1 && a->pro_mode != 3)) return SPB_E_UNSUPPORTED;
    if (a->bias != nullptr || a->res != nullptr || (a->lda > 0 && a->lda != a->K)) return SPB_E_UNSUPPORTED;
    if (a->N == 96 && a->pro_mode == 1) return launch_st<1, 6, 1, true>(*a, stream);     // 16 -> 96 at 112x112: one split
    return a->pro_mode == 3 ? launch_st<3, 3, 1, true>(*a, stream) : launch_st<1, 3, 1, true>(*a, stream);
  }
  if (!g_st_on || a->M < g_st_min_m || (a->N & 7) || (a->K & 7)) return SPB_E_UNSUPPORTED;
  if (a->epi_mode != 1 || (a->pro_mode != 1 && a->pro_mode != 3)) return SPB_E_UNSUPPORTED;
  if (a->bias != nullptr || a->out_act != SPB_ACT_NONE || a->out_scale != 1.f || a->res != nullptr) return SPB_E_UNSUPPORTED;
  if ((a->lda > 0 && a->lda != a->K) || (a->ldc > 0 && a->ldc != a->N)) return SPB_E_UNSUPPORTED;
  const int N = a->N, K = a->K;
  const bool join = a->pro_mode == 3;
  if (K <= 32) {                                            // expand convolutions (and 32 -> 16)
    if (N == 16 && !join) return launch_st<1, 1, 1>(*a, stream);
    if (N % 48 == 0 && N <= 192) return join ? launch_st<3, 3, 1>(*a, stream) : launch_st<1, 3, 1>(*a, stream);   // 96, 144: splits of 48
    if (N <= 32 && !join) return launch_st<1, 2, 1>(*a, stream);
    return SPB_E_UNSUPPORTED;
  }
  // project convolutions with a longer reduction (96 -> 24, 144 -> 24 | 32 at 56x56).  The first form (two whole register buffers, 154 / 222
  // registers) was 1-4 us SLOWER than the tiled kernel; with ONE buffer refilled in place and the coefficients re-read from LDS per chunk
  // (89 / 119 registers): 17.1 -> 14.2 us and 21.5 -> 18.0 us
  if (g_st_long_k && !join && N <= 32) {
    if (K <= 96) return launch_st<1, 2, 3>(*a, stream);
    if (K <= 160) return launch_st<1, 2, 5>(*a, stream);
  }
  return SPB_E_UNSUPPORTED;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_st(int on, int min_m, int wgs) {
  g_st_on = on & 1;
  g_st_long_k = (on >> 1) & 1;       // on = 3: also the long-reduction project instances
  if (min_m > 0) g_st_min_m = min_m;
  if (wgs > 0) g_st_wgs = wgs;
  return 0;
}
#endif
