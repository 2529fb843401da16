// Pointwise (1x1) convolution GEMMs with a SHORT reduction and a WIDE output on the 28x28 / 14x14 maps: "row slab".
//
//   forward   Y[M,N]  = act(bn(A))[M,K] * W[N,K]^T  (or the residual join bn(A) + bn2(A2))   + sum(y), sum(y^2)     expand convs:  K = 32..96, N = 6K
//   dgrad     dA[M,N] = bn_bwd(G,Z)[M,K] * Wt[N,K]^T                                         + mask, sum(g), sum(g*xhat)   project convs: K = 32..96, N = 192..576
// Same contract as spb_pwconv_gemm (gemm_pw.hip): reference park2019.py:107-108 (torchvision MobileNetV2 expand / project convolutions).
//
// Why another GEMM.  These launches are all epilogue: the reduction is one to three MFMA steps, the bytes are the [M, N] output (and, for the
// input gradient, the equally large raw tensor of the output side for the activation mask).  The tiled kernel does them in 64 x 64 tiles:
// three dependent memory round trips per tile (operands, output-side operand, stores + statistics) and, beyond 1024 workgroups, two tiles
// per workgroup back to back -- 17-34 us for 17-34 MB (round-3 trace).  Here a launch is ONE round trip:
//   * a workgroup owns 16*RF rows and ALL N columns; wave w owns columns [w*N/4, (w+1)*N/4);
//   * every global load of the launch is issued before anything is waited for: the raw A rows (whole workgroup), the wave's weight fragments
//     straight into registers in the matrix-core layout (W is L2 resident), the lane's own output-side operands, the BatchNorm sums;
//   * the product is formed transposed, W * a^T, with the weight rows permuted so that a lane ends up with 8 CONSECUTIVE channels of one
//     row per fragment pair and the four lanes of a row with 64 contiguous bytes (the trick of pw_bwd_fused.hip, taken one step further):
//     the output-side operand is read and the result is written by the lane itself in 16-byte accesses, no LDS staging;
//   * a wave's columns are its own: the per-channel sums reduce over the 16 lanes of a DPP row, meet in 2*N floats of LDS and leave as
//     coalesced atomics, one per channel, sum and workgroup.
// LDS holds the transformed A tile and the coefficient tables only (< 16 KB): the launch is bound by how fast the chip starts workgroups.
// bf16 only (the f32 parity mode keeps the tiled kernel).
#include "common.h"
#include <hip/hip_ext.h>

#ifndef SPB_TS_DECL       // register-held phase timestamps (scratch/ubench_gemm2.hip); compiled out in the product build
#define SPB_TS_DECL
#define SPB_TSR(i)
#define SPB_TS_FLUSH
#endif

namespace {

// PRO 1: a = act(bn(A));  2: a = bn_backward(g = A, z = A2);  3: a = bn(A) + bn2(A2), first... (every workgroup owns whole rows: it writes Ymat)
// EPI 1: y = acc, sums of y, y^2;  2: g = acc * act'(bn(Zout)), sums of g, g*xhat
// NJ = 16-column fragments per wave (N = 64 * NJ), RF = 16-row fragments per workgroup, KS = 32-deep reduction steps (K <= 32 * KS)
template <int PRO, int EPI, int NJ, int RF, int KS>
__global__ __launch_bounds__(256, NJ >= 9 ? 2 : (NJ >= 6 ? 2 : 4)) void pw_rs_kernel(const spb_gemm_args_t g) {   // M / 32 workgroups must be resident at once
  constexpr int BM = 16 * RF, KP = 32 * KS, LDA = KP + 8;      // padded A row: conflict-free 16-byte fragment reads
  constexpr int NW = 16 * NJ;                                   // columns per wave
  __shared__ __attribute__((aligned(16))) bf16_t As[BM * LDA];
  __shared__ float coef[3 * KP];
  __shared__ float ecoef[EPI == 2 ? 4 * 64 * NJ : 1];           // [4][N]: scale, shift, mean, invstd of the output-side BatchNorm
  __shared__ float sred[2 * 64 * NJ];                           // [2][N]: the workgroup's per-channel sums
  SPB_TS_DECL;
  SPB_TSR(0);
  const int M = g.M, K = g.K, N = 64 * NJ;
  const int t = threadIdx.x, l = t & 63, li = l & 15, lq = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * BM;
  const bf16_t* Ag = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* A2g = (PRO >= 2 && g.A2) ? reinterpret_cast<const bf16_t*>(g.A2) : Ag;
  const bf16_t* Bg = reinterpret_cast<const bf16_t*>(g.Bw);
  bf16_t* Yg = reinterpret_cast<bf16_t*>(g.Y);
  const bf16_t* Zg = reinterpret_cast<const bf16_t*>(g.Zout);
  bf16_t* Ymat = reinterpret_cast<bf16_t*>(g.Ymat);

  // ---- every load of the launch
  // (1) raw A rows: BM * KP / 8 16-byte vectors over the 256 threads
  constexpr int KV = KP / 8, NVEC = BM * KV, NL = (NVEC + 255) / 256;
  uint4 ra[NL], ra2[PRO >= 2 ? NL : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int e = t + 256 * i, row = e / KV, kv = e - row * KV;
    const int m = m0 + row, k = kv * 8;
    const size_t o = (size_t)(m < M ? m : M - 1) * K + (k < K ? k : K - 8);
    ra[i] = *reinterpret_cast<const uint4*>(Ag + o);
    if (PRO >= 2) ra2[PRO >= 2 ? i : 0] = *reinterpret_cast<const uint4*>(A2g + o);
  }
  // (2) the wave's weight fragments (A operand of W * a^T).  Fragments are PAIRED: row p' of fragments 2p and 2p+1 is output channel
  //     w*NW + p*32 + (p' >> 2)*8 + {0, 4} + (p' & 3), so that lane (li, lq) ends up with the 8 consecutive channels
  //     w*NW + p*32 + lq*8 .. +7 of its row in acc[2p], acc[2p+1] -- one 16-byte access, and the four lq lanes of a row cover 64
  //     contiguous bytes in ONE instruction (8-byte pieces scattered over 64 addresses per store instruction were measured at 9 G
  //     pieces/s: 100 us for a 7 MB output).  An odd last fragment holds 16 channels on its own: 4 per lane, one 8-byte access.
  auto chan = [&](int j, int pr) {   // output channel of row pr (0..15) of fragment j
    if ((NJ & 1) && j == NJ - 1) return w * NW + (NJ - 1) * 16 + pr;
    return w * NW + (j >> 1) * 32 + (pr >> 2) * 8 + (j & 1) * 4 + (pr & 3);
  };
  uint4 wf[NJ][KS];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = chan(j, li);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + lq * 8;
      wf[j][ks] = *reinterpret_cast<const uint4*>(Bg + (size_t)n * K + (k < K ? k : K - 8));
    }
  }
  // (3) the lane's output-side operands: row m0 + f*16 + li; pair p: channels w*NW + p*32 + lq*8 .. +7 (16 bytes), tail: 4 channels (8 bytes)
  constexpr int NP = NJ / 2;                       // fragment pairs
  const int cpair = w * NW + lq * 8, ctail = w * NW + (NJ - 1) * 16 + lq * 4;
  uint4 zr[EPI == 2 ? RF : 1][EPI == 2 && NP > 0 ? NP : 1];
  uint2 zt[EPI == 2 ? RF : 1];
  if (EPI == 2) {
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int m = m0 + f * 16 + li;
      const bf16_t* zp = Zg + (size_t)(m < M ? m : M - 1) * N;
#pragma unroll
      for (int p = 0; p < NP; ++p) zr[EPI == 2 ? f : 0][p] = *reinterpret_cast<const uint4*>(zp + cpair + p * 32);
      if (NJ & 1) zt[EPI == 2 ? f : 0] = *reinterpret_cast<const uint2*>(zp + ctail);
    }
  }
  // (4) coefficient tables
  if constexpr (PRO == 3) bn_join_table(g.pro, g.pro2, K, KP, coef, t);
  else bn_coef_table<PRO == 1 ? 1 : 2>(g.pro, K, KP, coef, t);
  if (EPI == 2) {     // (its own round trip: the table above is built by all 256 threads; EPI == 2 instances of this kernel are not in the KRN plan)
    BNEpiPre epre;
    bn_epi_issue(g.epi, 0, N, N, t, epre);
    bn_epi_finish<true>(g.epi, 0, N, N, N, ecoef, t, epre);
  }
  __syncthreads();
  SPB_TSR(1);
  // ---- transform the A rows into LDS (and, for the residual join, write the joined block output)
  const float act_h = act_hi(g.pro.act), act_n = act_ns(g.pro.act, g.pro.slope);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int e = t + 256 * i, row = e / KV, kv = e - row * KV;
    if (e < NVEC) {
      const int k = kv * 8, m = m0 + row;
      Raw8<bf16_t> r1; r1.u = ra[i];
      float a[8], x[8];
      cvt8(r1, a);
      if (PRO == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = a[j] * coef[k + j] + coef[KP + k + j];
          x[j] = __builtin_amdgcn_fmed3f(u, 0.f, act_h) + act_n * fminf(u, 0.f);
        }
      } else {
        Raw8<bf16_t> r2; r2.u = ra2[PRO >= 2 ? i : 0];
        float a2[8];
        cvt8(r2, a2);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[j] * coef[k + j] + a2[j] * coef[KP + k + j] + coef[2 * KP + k + j];
      }
      const bool ok = m < M && k < K;
      uint4 pa;
      pa.x = pack_bf16x2(x[0], x[1]); pa.y = pack_bf16x2(x[2], x[3]); pa.z = pack_bf16x2(x[4], x[5]); pa.w = pack_bf16x2(x[6], x[7]);
      if (!ok) pa = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(As + row * LDA + k) = pa;
      if (PRO == 3) { if (ok) *reinterpret_cast<uint4*>(Ymat + (size_t)m * K + k) = pa; }
    }
  }
  __syncthreads();
  SPB_TSR(2);
  // ---- W * a^T: C[row = channel (lq*4 + r of fragment j)][col = li = row of the slab]
  f32x4_t acc[RF][NJ];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[f][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8_t bfr[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) bfr[f] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(As + (f * 16 + li) * LDA + ks * 32 + lq * 8));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, wf[j][ks]);
#pragma unroll
      for (int f = 0; f < RF; ++f) acc[f][j] = SPB_MFMA16(af, bfr[f], acc[f][j]);
    }
  }
  SPB_TSR(3);
  // ---- epilogue in registers
  float s1[NJ][4], s2[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
  // one group of 4 channels: accumulator -> (mask) -> rounded output word pair, sums
  auto fin4 = [&](const f32x4_t& av, uint2 zz, int c0, bool ok, float (&t1)[4], float (&t2)[4]) {
    float v[4] = {av[0], av[1], av[2], av[3]}, z[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == 2) {
      spb_unpack2(zz.x, z[0], z[1]); spb_unpack2(zz.y, z[2], z[3]);
      const float4 sc = *reinterpret_cast<const float4*>(ecoef + c0), sh = *reinterpret_cast<const float4*>(ecoef + N + c0);
      const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= act_grad(z[r] * scv[r] + shv[r], g.epi.act, g.epi.slope);
    }
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    float q[4];
    spb_unpack2(o.x, q[0], q[1]); spb_unpack2(o.y, q[2], q[3]);            // the sums see the rounded values, as every other producer's
    if (ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { t1[r] += q[r]; t2[r] += EPI == 2 ? q[r] * z[r] : q[r] * q[r]; }
    }
    return o;
  };
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    const int m = m0 + f * 16 + li;
    const bool ok = m < M;
    bf16_t* yp = Yg + (size_t)(ok ? m : M - 1) * N;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint4 zz = EPI == 2 ? zr[EPI == 2 ? f : 0][p] : make_uint4(0, 0, 0, 0);
      const uint2 lo = fin4(acc[f][2 * p], make_uint2(zz.x, zz.y), cpair + p * 32, ok, s1[2 * p], s2[2 * p]);
      const uint2 hi = fin4(acc[f][2 * p + 1], make_uint2(zz.z, zz.w), cpair + p * 32 + 4, ok, s1[2 * p + 1], s2[2 * p + 1]);
      if (ok) *reinterpret_cast<uint4*>(yp + cpair + p * 32) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    if (NJ & 1) {
      const uint2 o = fin4(acc[f][NJ - 1], EPI == 2 ? zt[EPI == 2 ? f : 0] : make_uint2(0, 0), ctail, ok, s1[NJ - 1], s2[NJ - 1]);
      if (ok) *reinterpret_cast<uint2*>(yp + ctail) = o;
    }
  }
  SPB_TSR(4);
  // ---- per-channel sums: over the 16 lanes of a DPP row (the rows of the slab) -> LDS [2][N] (the A tile is idle; a wave's channels are
  // its own) -> one coalesced f32 atomic per channel, sum and workgroup
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = row16_sum(s1[j][r]), b = row16_sum(s2[j][r]);
      if (li == ((j * 4 + r) & 15)) {
        const int c = ((NJ & 1) && j == NJ - 1) ? ctail + r : cpair + (j >> 1) * 32 + (j & 1) * 4 + r;
        sred[c] = a; sred[N + c] = b;
      }
    }
  __syncthreads();
  for (int i = t; i < 2 * N; i += 256) {
    const int which = i >= N, c = i - which * N;
    float v = sred[i];
    if (EPI == 2 && which) v = ecoef[3 * N + c] * (v - ecoef[2 * N + c] * sred[c]);   // sum g*z -> sum g*xhat
    atomicAdd(g.osums + (size_t)(blockIdx.x % g.oR) * 2 * N + i, v);
  }
  SPB_TSR(5);
  SPB_TS_FLUSH;
}

template <int PRO, int EPI, int NJ, int RF, int KS>
int launch_rs(const spb_gemm_args_t& g, hipStream_t stream) {
  constexpr int BM = 16 * RF;
  const int grid = (g.M + BM - 1) / BM;
  if (g.stop_event)
    hipExtLaunchKernelGGL((pw_rs_kernel<PRO, EPI, NJ, RF, KS>), dim3(grid), dim3(256), 0, stream, nullptr, (hipEvent_t)g.stop_event, 0, g);
  else
    hipLaunchKernelGGL((pw_rs_kernel<PRO, EPI, NJ, RF, KS>), dim3(grid), dim3(256), 0, stream, g);
  SPB_CHECK_LAUNCH();
  return 0;
}

template <int PRO, int EPI>
int dispatch_rs(const spb_gemm_args_t& g, hipStream_t stream) {
  const int nj = g.N / 64, ks = (g.K + 31) / 32;
  if (nj == 3 && ks == 1) return launch_rs<PRO, EPI, 3, 2, 1>(g, stream);      // 28x28: 32 <-> 192
  if (nj == 3 && ks == 2) return launch_rs<PRO, EPI, 3, 2, 2>(g, stream);      // 14x14: 64 <-  192 (block 7's project conv)
  if (nj == 6 && ks == 2) return launch_rs<PRO, EPI, 6, 2, 2>(g, stream);      // 14x14: 64 <-> 384
  if (nj == 6 && ks == 3) return launch_rs<PRO, EPI, 6, 2, 3>(g, stream);      // 14x14: 96 <-  384
  if (nj == 9 && ks == 3) return launch_rs<PRO, EPI, 9, 2, 3>(g, stream);      // 14x14: 96 <-> 576
  return SPB_E_UNSUPPORTED;
}

int g_rs_on = 1, g_rs_min_m = 4096;

}  // namespace

// bf16 only; SPB_E_UNSUPPORTED tells spb_pwconv_gemm to use the other kernels
int spb_gemm_rs(const spb_gemm_args_t* a, hipStream_t stream) {
  if (!g_rs_on || a->M < g_rs_min_m || (a->N & 63) || (a->K & 7) || a->K > 96) return SPB_E_UNSUPPORTED;
  if (a->bias != nullptr || a->out_act != SPB_ACT_NONE || a->out_scale != 1.f || a->res != nullptr) return SPB_E_UNSUPPORTED;
  if ((a->lda > 0 && a->lda != a->K) || (a->ldc > 0 && a->ldc != a->N)) return SPB_E_UNSUPPORTED;
  if (a->pro_mode == 1 && a->epi_mode == 1) return dispatch_rs<1, 1>(*a, stream);
  if (a->pro_mode == 3 && a->epi_mode == 1) return dispatch_rs<3, 1>(*a, stream);
  if (a->pro_mode == 2 && a->epi_mode == 2) return dispatch_rs<2, 2>(*a, stream);
  return SPB_E_UNSUPPORTED;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gemm_rs(int on, int min_m) {
  g_rs_on = on;
  if (min_m > 0) g_rs_min_m = min_m;
  return 0;
}
#endif
