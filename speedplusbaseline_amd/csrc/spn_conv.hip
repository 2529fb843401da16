// Spacecraft Pose Network: the AlexNet trunk's convolutions as implicit GEMMs on the matrix cores (reference: spn.py:60-101,
// nn.Conv2d 5x5 / 3x3, groups 1 or 2, + ReLU).  No column matrix in HBM: the reduction runs over (tap, channel), and the
// operand row of output pixel m for tap (ky, kx) is the NHWC input pixel (b, oy*s + ky - pad, ox*s + kx - pad) -- 2*C
// contiguous bytes -- or a page of zeros outside the image.
//
//   forward    Y[m][n]  = relu(bias[n] + sum_{tap,c} X[pix(m,tap)][g*Cg + c] * Wp[n][tap*Cg + c])
//   input grad dX[p][c] = (Y_below[p][c] > 0) * sum_{tap',n} G[pix'(p,tap')][g*Ng + n] * WpD[c][tap'*Ng + n]
//              the same kernel on the output gradient with the taps mirrored (stride-1 layers: conv2..conv5), the ReLU mask of
//              the layer below folded into the store
//   weight grad dWp[n][tap*Cg + c] = sum_m G[m][n] * X[pix(m,tap)][g*Cg + c]         (spn_conv_wgrad_kernel)
//
// Data movement: global -> LDS with `global_load_lds_dwordx4` (dma16), a ring of DS stages of 64 reduction elements, counted
// s_waitcnt + one s_barrier per stage; each lane keeps the (tap, channel) position of ITS 16-byte slot and steps it by 64
// elements per stage, so the gather costs a handful of integer instructions per stage.  HBM traffic per launch is the input
// map (read about once per column tile through L2), the weights and the output -- the column matrix the round-1 path wrote
// and read back was 6x (3x3) to 25x (5x5) the input.
#include "common.h"
#include <cstring>

namespace {

constexpr int CBN = 64, CBK = 64, CDS = 4;

struct ConvGeo {
  int B, H, W, Cx;          // input map (NHWC, Cx channels in total)
  int OH, OW, KH, KW, stride, pad;
  int groups, Cg, Ng;       // input / output channels per group
  int Kp;                   // packed weight row length (>= KH*KW*Cg, multiple of 8)
  int M;                    // B*OH*OW output pixels
  int kw_inv;               // ceil(65536 / KW): tap / KW without a division
  int plain;                // weight gradient only: X is an explicit [M][Cx] column matrix (k = column), no gather
};

// RW: 16-row fragments per wave (workgroup tile = 64*RW rows x 64 columns)
template <int RW>
__global__ __launch_bounds__(256) void spn_conv_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, const float* __restrict__ bias,
                                                       const bf16_t* __restrict__ mask, bf16_t* __restrict__ Y, const bf16_t* __restrict__ zero,
                                                       const ConvGeo g, int relu) {
  constexpr int BM = 64 * RW;
  constexpr int A_BYTES = BM * CBK * 2, B_BYTES = CBN * CBK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int IPS = 2 * RW + 2;                  // DMA instructions per stage per wave
  constexpr int LDO = CBN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int K = g.KH * g.KW * g.Cg;
  const int KT = (K + CBK - 1) / CBK;
  const int NTg = (g.Ng + CBN - 1) / CBN, NT = NTg * g.groups;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lid % NT, m0 = (lid / NT) * BM;
  const int gi = ntile / NTg, n0 = (ntile % NTg) * CBN;
  const int Ntot = g.Ng * g.groups;

  // ---- per-lane DMA sources.  A: rows w*16*RW + i*8 + (l>>3), i < 2*RW; B: rows w*16 + i*8 + (l>>3), i < 2.
  // 16-byte slot l&7 of a row holds reduction vector slot ^ (row & 7); (row & 7) == (l >> 3) & 7 for every row of this lane.
  const int dkv = (l & 7) ^ ((l >> 3) & 7);
  long long pix[2 * RW];
  int iy0[2 * RW], ix0[2 * RW];
#pragma unroll
  for (int i = 0; i < 2 * RW; ++i) {
    const int m = m0 + w * 16 * RW + i * 8 + (l >> 3);
    const int mc = m < g.M ? m : g.M - 1;
    const int ox = mc % g.OW, oy = (mc / g.OW) % g.OH, b = mc / (g.OW * g.OH);
    iy0[i] = m < g.M ? oy * g.stride - g.pad : -(1 << 20);     // rows past the end read the zero page
    ix0[i] = ox * g.stride - g.pad;
    pix[i] = ((long long)(b * g.H + iy0[i]) * g.W + ix0[i]) * g.Cx + gi * g.Cg;
  }
  size_t brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + w * 16 + i * 8 + (l >> 3);
    brow[i] = (size_t)(gi * g.Ng + (n < g.Ng ? n : g.Ng - 1)) * g.Kp;
  }
  // position of this lane's slot in the reduction: element k = tap*Cg + c
  int kk = dkv * 8, tap = 0, c = dkv * 8;
  while (c >= g.Cg) { c -= g.Cg; ++tap; }
  const unsigned stages_lds = lds_addr(smem);
  const unsigned wave_a = __builtin_amdgcn_readfirstlane((unsigned)(w * 16 * RW * CBK * 2));
  const unsigned wave_b = __builtin_amdgcn_readfirstlane((unsigned)(A_BYTES + w * 16 * CBK * 2));
#define CONV_STAGE(kt_)                                                                              \
  {                                                                                                  \
    const unsigned sb = stages_lds + (unsigned)(((kt_) % CDS) * STAGE);                              \
    const int ky = (tap * g.kw_inv) >> 16, kx = tap - ky * g.KW;                                     \
    const long long koff = (long long)(ky * g.W + kx) * g.Cx + c;                                    \
    const bool kok = kk < K;                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2 * RW; ++i) {                                             \
      const bool ok = kok && (unsigned)(iy0[i] + ky) < (unsigned)g.H && (unsigned)(ix0[i] + kx) < (unsigned)g.W; \
      dma16(ok ? X + (pix[i] + koff) : zero, sb + wave_a + (unsigned)(i * 8 * CBK * 2));             \
    }                                                                                                \
    const int kc = kk < g.Kp ? kk : g.Kp - 8;                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) dma16(Wp + brow[i] + kc, sb + wave_b + (unsigned)(i * 8 * CBK * 2)); \
    kk += CBK; c += CBK;                                                                             \
    while (c >= g.Cg) { c -= g.Cg; ++tap; }                                                          \
  }
  for (int s = 0; s < CDS - 1 && s < KT; ++s) CONV_STAGE(s);

  // epilogue operands that do not depend on the reduction
  constexpr int NV = CBN / 8, VR = 256 / NV, VRI = BM / VR;
  const int vcol = t % NV, vrow0 = t / NV;
  const int nE = n0 + vcol * 8;
  const bool colok = nE < g.Ng;
  float e_bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e_bias[j] = (colok && bias) ? bias[gi * g.Ng + nE + j] : 0.f;

  f32x4_t acc[RW][CBN / 16];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < CBN / 16; ++j) acc[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < KT; ++kt) {
    // stage kt has landed once at most min(CDS-2, KT-1-kt) younger stages are still in flight
    const int rem = KT - 1 - kt;
    if (rem >= CDS - 2) wait_vmcnt<(CDS - 2) * IPS>();
    else if (rem == 1) wait_vmcnt<IPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();      // every wave's share of stage kt is visible; everyone is done reading stage kt-1
    if (kt + CDS - 1 < KT) CONV_STAGE(kt + CDS - 1);   // into the buffer stage kt-1 just vacated
    const char* sb = smem + (size_t)(kt % CDS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < CBK / 32; ++ks) {
      const int v = ks * 4 + lq;
      bf16x8_t af[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int frow = w * 16 * RW + r * 16 + li;
        af[r] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + frow * (CBK * 2) + ((v ^ (frow & 7)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < CBN / 16; ++j) {
        const int br = j * 16 + li;
        const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + A_BYTES + br * (CBK * 2) + ((v ^ (br & 7)) << 4)));
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r][j] = SPB_MFMA16(af[r], bf, acc[r][j]);
      }
    }
  }
#undef CONV_STAGE
  __syncthreads();    // all DMA consumed (the last stage was waited with vmcnt(0)): the ring becomes the output tile

  bf16_t* Os = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < CBN / 16; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) Os[(w * 16 * RW + r * 16 + lq * 4 + e) * LDO + j * 16 + li] = f2bf(acc[r][j][e]);
  __syncthreads();
  if (colok) {
#pragma unroll
    for (int s = 0; s < VRI; ++s) {
      const int r = vrow0 + s * VR;
      const int m = m0 + r;
      if (m < g.M) {
        float v[8];
        ld8<bf16_t>(Os + r * LDO + vcol * 8, v);
        const size_t o = (size_t)m * Ntot + gi * g.Ng + nE;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] += e_bias[j];
          if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        if (mask) {
          float mk[8];
          ld8<bf16_t>(mask + o, mk);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = mk[j] > 0.f ? v[j] : 0.f;
        }
        if (nE + 8 <= g.Ng) st8<bf16_t>(Y + o, v);
        else
          for (int j = 0; j < 8; ++j) if (nE + j < g.Ng) Y[o + j] = f2bf(v[j]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ RGB stem
// conv1 (3 -> 96, 11x11, stride 4, no padding) straight from the float32 NCHW image, no column matrix.  A workgroup owns a band
// of STEM_R output rows of one image: the 4*STEM_R + 7 input rows are staged ONCE in LDS (bf16, planar per colour, coalesced
// row loads), and every matrix operand is then two aligned 8-byte LDS reads: the reduction index is k' = (ci*11 + ky)*16 + kx
// with the kernel row padded from 11 to 16 columns (zero weights), so a lane's 8 consecutive k' are 8 consecutive pixels of one
// image row starting at 4*ox + {0, 8}.  y^T = W * patch^T as in the style decoder (rows of W permuted so a lane stores 24
// consecutive channels).  The first version of this kernel gathered its operands from global memory with 8 scalar loads per
// 16-byte slot: 82 us alone and 350 us beside the HBM-bound parameter update; the column-matrix path is 67 + 41 us.
constexpr int STEM_R = 4, STEM_KH = 11, STEM_KW = 11, STEM_ST = 4, STEM_NB = 6, STEM_STEPS = (3 * STEM_KH + 1) / 2;
constexpr int STEM_OW = 55, STEM_WP = (STEM_ST * (STEM_OW - 1) + 16 + 3) / 4 * 4;   // 227-wide images: 55 output columns, 232 staged
__global__ __launch_bounds__(256) void spn_stem_kernel(const float* __restrict__ x, const bf16_t* __restrict__ Wb, const float* __restrict__ bias,
                                                       bf16_t* __restrict__ Y, int B, int H, int W, int OH, int OW, int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BR = STEM_ST * (STEM_R - 1) + STEM_KH;            // band rows
  constexpr int N = STEM_NB * 16, KB = STEM_STEPS * 32;
  constexpr int WP = STEM_WP, TOTAL = 3 * BR * WP, NLD = (TOTAL + 255) / 256;   // band elements, loads per thread
  constexpr int WGRAN = STEM_STEPS * STEM_NB * 64, NWD = (WGRAN + 255) / 256;    // weight granules of 16 bytes, DMA instructions per thread
  bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                   // [STEPS][NB][16 rows][32]: fragment order, 1 KB per fragment
  bf16_t* band = wl + STEM_STEPS * STEM_NB * 512;                 // [3][BR][WP]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int nbands = (OH + STEM_R - 1) / STEM_R, total_bands = B * nbands;
  // Everything this workgroup reads from memory is issued in as few dependent round trips as possible: beside the HBM-bound
  // parameter update a round trip costs tens of microseconds (the memory system is saturated by ~300 k queued workgroups), and
  // the versions that streamed the weights step by step (17 round trips) or staged the band 16 loads at a time took 250-500 us
  // there against 40-55 us alone.  Weights (104 KB, fragment order) go global -> LDS by DMA, 26 instructions per thread back to
  // back; the band's 63 loads per thread are all in flight at once; the next band's loads are issued before this band's
  // matrix-core loop and consumed after it.
  {
    const unsigned wl_lds = lds_addr(wl) + __builtin_amdgcn_readfirstlane((unsigned)(wave * 64 * 16));   // wave-uniform for the DMA base
#pragma unroll
    for (int jj = 0; jj < NWD; ++jj) {
      const int i = t + 256 * jj;
      const int ic = i < WGRAN ? i : WGRAN - 1;                   // the surplus lanes of the last instruction land in the band area,
      const int q = ic & 3, r = (ic >> 2) & 15, nb = (ic >> 6) % STEM_NB, st_ = ic / (64 * STEM_NB);   // which is written afterwards
      const int co = (r >> 2) * 4 * STEM_NB + nb * 4 + (r & 3);   // channel permutation: a lane ends up with 4*NB consecutive channels
      dma16(Wb + (size_t)co * KB + st_ * 32 + q * 8, wl_lds + (unsigned)(jj * 256 * 16));
    }
  }
  float v[NLD];
  auto band_issue = [&](int bandi_) {                             // element e = (ci, row, x); outside the image: zeros
    const int bb = bandi_ / nbands, iy0_ = (bandi_ % nbands) * STEM_R * STEM_ST;
    const float* xb = x + (size_t)bb * 3 * H * W;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = t + 256 * u;
      const int ec = e < TOTAL ? e : TOTAL - 1;
      const int xx = ec % WP, rr = (ec / WP) % BR, ci = ec / (WP * BR);
      const int iy = iy0_ + rr;
      const bool ok = xx < W && iy < H;
      const float val = xb[((size_t)ci * H + (ok ? iy : 0)) * W + (ok ? xx : 0)];
      v[u] = ok ? val : 0.f;
    }
  };
  auto band_commit = [&]() {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = t + 256 * u;
      if (e < TOTAL) band[e] = f2bf(v[u]);
    }
  };
  band_issue(blockIdx.x < total_bands ? blockIdx.x : 0);
  wait_vmcnt<0>();                                                // weights (DMA) and the first band's loads
  __syncthreads();
  band_commit();
  for (int bandi = blockIdx.x; bandi < total_bands; bandi += gridDim.x) {
  const int b = bandi / nbands, oy0 = (bandi % nbands) * STEM_R;
  __syncthreads();                                                // band (and, first pass, weights) complete in LDS
  const int nxt = bandi + gridDim.x;
  band_issue(nxt < total_bands ? nxt : bandi);                    // in flight during the matrix-core loop (last pass: redundant)
  const bf16_t* wfrag = wl + (li * 4 + lq) * 8;
  // wave = output row of the band; four 16-pixel groups cover the 55 columns (pixel index clamped for the addresses)
  f32x4_t acc[4][STEM_NB];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int nb = 0; nb < STEM_NB; ++nb) acc[g][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int pxo[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pxo[g] = STEM_ST * min(g * 16 + li, OW - 1) + (lq & 1) * 8;
  for (int s = 0; s < STEM_STEPS; ++s) {
    uint4 a[STEM_NB];
#pragma unroll
    for (int nb = 0; nb < STEM_NB; ++nb) a[nb] = *reinterpret_cast<const uint4*>(wfrag + (s * STEM_NB + nb) * 512);
    const int r = min(2 * s + (lq >> 1), 3 * STEM_KH - 1);        // (ci, ky) row of this lane's 8 reduction elements (pad row: zero weights)
    const int ci = r / STEM_KH, ky = r - ci * STEM_KH;
    const bf16_t* brow = band + ((size_t)(ci * BR + wave * STEM_ST + ky)) * WP;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint2 lo = *reinterpret_cast<const uint2*>(brow + pxo[g]), hi = *reinterpret_cast<const uint2*>(brow + pxo[g] + 4);
      const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
      for (int nb = 0; nb < STEM_NB; ++nb)
        acc[g][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, a[nb]), bf, acc[g][nb]);
    }
  }
  // ---- epilogue: lane (li = pixel, lq): channels lq*24 + nb*4 + e, 48 contiguous bytes
  const int oy = oy0 + wave;
  const int co0 = lq * 4 * STEM_NB;
  if (oy < OH) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ox = g * 16 + li;
      if (ox >= OW) continue;
      bf16_t* dst = Y + ((size_t)(b * OH + oy) * OW + ox) * N + co0;
      uint2 o[STEM_NB];
#pragma unroll
      for (int nb = 0; nb < STEM_NB; ++nb) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[g][nb][e] + (bias ? bias[co0 + nb * 4 + e] : 0.f);
          if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        o[nb].x = pack_bf16x2(v[0], v[1]); o[nb].y = pack_bf16x2(v[2], v[3]);
      }
#pragma unroll
      for (int nb = 0; nb < STEM_NB; nb += 2) *reinterpret_cast<uint4*>(dst + nb * 4) = make_uint4(o[nb].x, o[nb].y, o[nb + 1].x, o[nb + 1].y);
    }
  }
  __syncthreads();                                                // every wave is done reading this band
  band_commit();                                                  // the next one (prefetched above)
  }   // bands of this workgroup
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// dWp[n][k] += sum_m G[m][n] * X[pix(m, tap(k))][c(k)].  A workgroup owns a 64 x 64 tile of (n, k) and a range of output
// pixels m; the reduction axis m is the slow axis of both operands, so the LDS tiles ([64 m][64] bf16, filled by LDS-DMA, CDS
// stages deep) are read with the transpose load ds_read_b64_tr_b16.  The DMA image is lane-linear (rows of exactly 128
// bytes, no padding), so the 16-byte slot pairs of a row are XOR-swizzled with (row & 3): the 16 rows a transpose load
// touches then spread over the four 32-byte bank groups (4 cycles for 512 bytes -- the LDS floor).  The tap of a lane's
// 16-byte slot is fixed for the whole launch (the k tile is); only the pixel advances.
__device__ __forceinline__ int wg_sw(int row) { return (row & 3) << 1; }

__global__ __launch_bounds__(256) void spn_conv_wgrad_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X, float* __restrict__ dWp,
                                                             const bf16_t* __restrict__ zero, const ConvGeo g, int rows_per_split,
                                                             unsigned ow_magic, unsigned oh_magic) {
  constexpr int TILE = 64 * 64 * 2, STAGE = 2 * TILE, IPS = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int wn = w >> 1, wk = w & 1;
  const int K = g.plain ? g.Cg : g.KH * g.KW * g.Cg;
  const int NTg = (g.Ng + 63) / 64, KTk = (K + 63) / 64;
  const int tiles = g.groups * NTg * KTk;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);     // the tiles of one pixel range share G and X rows: one XCD, back to back
  const int tile = lid % tiles, split = lid / tiles;
  const int gi = tile / (NTg * KTk), n0 = ((tile / KTk) % NTg) * 64, k0 = (tile % KTk) * 64;
  const int mbeg = split * rows_per_split, mend = min(g.M, mbeg + rows_per_split);
  const int Ntot = g.Ng * g.groups;

  // this lane's DMA slots: rows w*16 + i*8 + (l>>3) (i < 2) of either tile, logical 16-byte vector (l & 7) ^ sw(row)
  const int drow = w * 16 + (l >> 3);
  const int vec = (l & 7) ^ wg_sw(drow);               // (drow + 8) & 3 == drow & 3
  const int nv = n0 + vec * 8, kv = k0 + vec * 8;
  const int tap = kv / g.Cg, c = kv - tap * g.Cg;
  const int ky = tap / g.KW, kx = tap - ky * g.KW;
  const bool kok = kv < K;
  const size_t gcol = (size_t)gi * g.Ng + (nv < g.Ng ? nv : g.Ng - 8);
  const long long xcol = (long long)gi * g.Cg + c;
  const unsigned lds0 = lds_addr(smem);
  const unsigned wave_o = __builtin_amdgcn_readfirstlane((unsigned)(w * 16 * 128));
#define WG_STAGE(st_, mb_)                                                                              \
  {                                                                                                     \
    const unsigned sb = lds0 + (unsigned)(((st_) % CDS) * STAGE) + wave_o;                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
      const int m = (mb_) + drow + 8 * i;                                                               \
      const int mc = m < mend ? m : mend - 1;                                                           \
      dma16(G + (size_t)mc * Ntot + gcol, sb + (unsigned)(i * 8 * 128));                                \
      if (g.plain) {                                                                                    \
        dma16((kok && m < mend) ? X + ((long long)mc * g.Cx + kv) : zero, sb + TILE + (unsigned)(i * 8 * 128)); \
      } else {                                                                                          \
      const int q1 = (int)__umulhi((unsigned)mc, ow_magic);            /* mc / OW */                    \
      const int ox = mc - q1 * g.OW;                                                                    \
      const int b = (int)__umulhi((unsigned)q1, oh_magic);             /* q1 / OH */                    \
      const int oy = q1 - b * g.OH;                                                                     \
      const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;                       \
      const bool ok = kok && m < mend && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;  \
      dma16(ok ? X + (((long long)(b * g.H + iy) * g.W + ix) * g.Cx + xcol) : zero, sb + TILE + (unsigned)(i * 8 * 128)); \
      }                                                                                                 \
    }                                                                                                   \
  }
  const int nst = (mend - mbeg + 63) / 64;
  for (int s = 0; s < CDS - 1 && s < nst; ++s) WG_STAGE(s, mbeg + s * 64);

  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  // transpose-load addresses within a stage: row = mc*32 + lq*8 + (li>>2) (+4), columns (frag*16 + (li&3)*4 .. +3)
  const int rsw = wg_sw(li >> 2);
  int offp[2], offq[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int cp = (wn * 2 + f) * 16 + (li & 3) * 4, cq = (wk * 2 + f) * 16 + (li & 3) * 4;
    offp[f] = (lq * 8 + (li >> 2)) * 128 + (((cp >> 3) ^ rsw) << 4) + (cp & 7) * 2;
    offq[f] = TILE + (lq * 8 + (li >> 2)) * 128 + (((cq >> 3) ^ rsw) << 4) + (cq & 7) * 2;
  }
  for (int st = 0; st < nst; ++st) {
    const int rem = nst - 1 - st;
    if (rem >= CDS - 2) wait_vmcnt<(CDS - 2) * IPS>();
    else if (rem == 1) wait_vmcnt<IPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (st + CDS - 1 < nst) WG_STAGE(st + CDS - 1, mbeg + (st + CDS - 1) * 64);
    const char* sb = smem + (size_t)(st % CDS) * STAGE;
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      bf16x8_t pf[2], qf[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        union { struct { s16x4_t lo, hi; } s; bf16x8_t v; } up, uq;
        up.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + mc * 32 * 128 + offp[f]));
        up.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + mc * 32 * 128 + 4 * 128 + offp[f]));
        uq.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + mc * 32 * 128 + offq[f]));
        uq.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(sb + mc * 32 * 128 + 4 * 128 + offq[f]));
        pf[f] = up.v; qf[f] = uq.v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = SPB_MFMA16(pf[a], qf[b], acc[a][b]);
    }
  }
#undef WG_STAGE
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * 2 + a) * 16 + lq * 4 + r;
        const int k = k0 + (wk * 2 + b) * 16 + li;
        if (n < g.Ng && k < K) atomicAdd(dWp + (size_t)(gi * g.Ng + n) * g.Kp + k, acc[a][b][r]);
      }
}

// weights for the input-gradient pass: WpD[g*Cg + ci][tap'*Ng + n] = W[g*Ng + n][ci][KH-1-ky'][KW-1-kx'], tap' = ky'*KW + kx'
__global__ void pack_conv_dgrad_kernel(const float* __restrict__ W, bf16_t* __restrict__ WpD, int Cout, int Cin, int G, int KH, int KW, int KpD) {
  const int cog = Cout / G, cig = Cin / G;
  const long long total = (long long)Cin * KpD;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / KpD), k = (int)(i % KpD);
    float v = 0.f;
    if (k < KH * KW * cog) {
      const int tp = k / cog, n = k % cog;
      const int ky = KH - 1 - tp / KW, kx = KW - 1 - tp % KW;
      const int gi = row / cig, ci = row % cig;
      v = W[(((size_t)(gi * cog + n) * cig + ci) * KH + ky) * KW + kx];
    }
    WpD[i] = f2bf(v);
  }
}

// every repack the next forward / backward needs after an optimizer step, in one launch (blockIdx.y = job): nine dependent
// 5-20 us launches sat between the optimizer and the next step's first convolution
struct PackJobs { spb_spn_pack_job_t j[SPB_SPN_MAX_PACK_JOBS]; };
__global__ void pack_jobs_kernel(const PackJobs jobs, int bf16) {
  const spb_spn_pack_job_t& q = jobs.j[blockIdx.y];
  const int cog = q.Cout / q.groups, cig = q.Cin / q.groups;
  const int rows = q.mode == 1 ? q.Cin : q.Cout;
  const long long total = (long long)rows * q.Kp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / q.Kp), k = (int)(i % q.Kp);
    float v = 0.f;
    if (q.mode == 2) {          // RGB stem band layout: k' = (ci*KH + ky)*16 + kx, kernel rows padded to 16 columns with zeros
      const int rr = k >> 4, kx = k & 15;
      if (kx < q.KW && rr < q.Cin * q.KH) v = q.W[((size_t)row * q.Cin * q.KH + rr) * q.KW + kx];
    } else if (q.mode == 1) {   // mirrored taps for the input gradient: [g*cig + ci][(tap', n)]
      if (k < q.KH * q.KW * cog) {
        const int tp = k / cog, n = k % cog;
        const int ky = q.KH - 1 - tp / q.KW, kx = q.KW - 1 - tp % q.KW;
        const int gi = row / cig, ci = row % cig;
        v = q.W[(((size_t)(gi * cog + n) * cig + ci) * q.KH + ky) * q.KW + kx];
      }
    } else if (k < q.KH * q.KW * cig) {   // forward: [co][(tap, ci)], or [co][(ci, tap)] for the RGB stem's column order
      v = q.chw ? q.W[(size_t)row * cig * q.KH * q.KW + k] : q.W[((size_t)row * cig + (k % cig)) * q.KH * q.KW + k / cig];
    }
    if (bf16) reinterpret_cast<bf16_t*>(q.out)[i] = f2bf(v); else reinterpret_cast<float*>(q.out)[i] = v;
    if (q.mode == 0 && q.outT) {          // [g][k][co in group]: the explicit-GEMM input gradient's operand
      const int gi = row / cog;
      const size_t o = ((size_t)gi * q.Kp + k) * cog + (row - gi * cog);
      if (bf16) reinterpret_cast<bf16_t*>(q.outT)[o] = f2bf(v); else reinterpret_cast<float*>(q.outT)[o] = v;
    }
  }
}

// One zeroed page per device (out-of-image taps read it).  Created on the first convolution call of a device with a blocking
// allocation + fill: make that first call outside any stream capture (every caller here warms up before capturing).
bf16_t* g_zero_page[64] = {};
bf16_t* zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!g_zero_page[dev]) {
    bf16_t* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_zero_page[dev] = p;
  }
  return g_zero_page[dev];
}

}  // namespace

extern "C" int spb_spn_conv(const spb_spn_conv_args_t* a, spb_stream_t stream) {
  if (!a || !a->X || !a->Wp || !a->Y) return SPB_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->groups <= 0 || a->Cg <= 0 || a->Ng <= 0 || a->KH <= 0 || a->KW <= 0 || a->stride <= 0)
    return SPB_E_ARG;
  // 16-byte operand vectors must not straddle taps or groups; rows of the packed weights hold whole vectors
  if ((a->Cg & 7) || (a->Cx & 7) || (a->Kp & 7) || a->Kp < a->KH * a->KW * a->Cg || a->Cx < a->groups * a->Cg) return SPB_E_UNSUPPORTED;
  if (a->mask && ((a->Ng * a->groups) & 7)) return SPB_E_UNSUPPORTED;
  ConvGeo g;
  g.B = a->B; g.H = a->H; g.W = a->W; g.Cx = a->Cx; g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad;
  g.OH = (a->H + 2 * a->pad - a->KH) / a->stride + 1; g.OW = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
  if (g.OH <= 0 || g.OW <= 0) return SPB_E_SHAPE;
  g.groups = a->groups; g.Cg = a->Cg; g.Ng = a->Ng; g.Kp = a->Kp;
  g.M = a->B * g.OH * g.OW;
  g.plain = 0;
  g.kw_inv = (65536 + a->KW - 1) / a->KW;
  if (a->KH * a->KW > 4096) return SPB_E_SHAPE;     // kw_inv is exact for tap < 2^16 / KW
  bf16_t* zp = zero_page();
  if (!zp) return SPB_E_STATE;
  const int NT = ((a->Ng + CBN - 1) / CBN) * a->groups;
  hipStream_t s = (hipStream_t)stream;
  // 128-row tiles once they still fill the chip twice over
  const bool big = (long long)((g.M + 127) / 128) * NT >= 512;
#define CONV_LAUNCH(RW_)                                                                                                 \
  {                                                                                                                      \
    constexpr int BM_ = 64 * RW_;                                                                                        \
    size_t lds = (size_t)CDS * (BM_ + CBN) * CBK * 2;                                                                    \
    const size_t os = (size_t)BM_ * (CBN + 8) * 2;                                                                       \
    if (os > lds) lds = os;                                                                                              \
    static bool once = false;                                                                                            \
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&spn_conv_kernel<RW_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; } \
    hipLaunchKernelGGL((spn_conv_kernel<RW_>), dim3(NT * ((g.M + BM_ - 1) / BM_)), dim3(256), lds, s, (const bf16_t*)a->X,   \
                       (const bf16_t*)a->Wp, a->bias, (const bf16_t*)a->mask, (bf16_t*)a->Y, zp, g, a->relu);           \
  }
  if (big) CONV_LAUNCH(2) else CONV_LAUNCH(1)
#undef CONV_LAUNCH
  SPB_CHECK_LAUNCH();
  return 0;
}

static unsigned magic_div(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }   // exact for m*d < 2^32... m < 2^32/d

extern "C" int spb_spn_conv_wgrad(const spb_spn_conv_args_t* a, const void* G, float* dWp, spb_stream_t stream) {
  if (!a || !a->X || !G || !dWp) return SPB_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->groups <= 0 || a->Cg <= 0 || a->Ng <= 0 || a->KH <= 0 || a->KW <= 0 || a->stride <= 0)
    return SPB_E_ARG;
  if ((a->Cg & 7) || (a->Cx & 7) || (a->Ng & 7) || (a->Kp & 7) || a->Kp < a->KH * a->KW * a->Cg || a->Cx < a->groups * a->Cg) return SPB_E_UNSUPPORTED;
  ConvGeo g;
  g.B = a->B; g.H = a->H; g.W = a->W; g.Cx = a->Cx; g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad;
  g.OH = (a->H + 2 * a->pad - a->KH) / a->stride + 1; g.OW = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
  if (g.OH <= 0 || g.OW <= 0) return SPB_E_SHAPE;
  g.groups = a->groups; g.Cg = a->Cg; g.Ng = a->Ng; g.Kp = a->Kp;
  g.M = a->B * g.OH * g.OW;
  g.plain = 0;
  g.kw_inv = 0;
  if ((long long)g.M * (g.OW > g.OH ? g.OW : g.OH) >= (1ll << 31)) return SPB_E_SHAPE;     // magic_div's exact range
  bf16_t* zp = zero_page();
  if (!zp) return SPB_E_STATE;
  const int K = a->KH * a->KW * a->Cg;
  const int tiles = a->groups * ((a->Ng + 63) / 64) * ((K + 63) / 64);
  int S = (1024 + tiles - 1) / tiles;
  const int maxS = (g.M + 255) / 256;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rps = ((g.M + S - 1) / S + 63) / 64 * 64;
  S = (g.M + rps - 1) / rps;
  const size_t lds = (size_t)CDS * 2 * 64 * 64 * 2;
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&spn_conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL(spn_conv_wgrad_kernel, dim3(tiles * S), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)G, (const bf16_t*)a->X, dWp, zp, g,
                     rps, magic_div(g.OW), magic_div(g.OH));
  SPB_CHECK_LAUNCH();
  return 0;
}

// the same kernel on an explicit column matrix (conv1: 3-channel taps do not form 16-byte vectors, so its columns are built by
// spb_im2col_rgb): dW[n][k] += sum_m G[m][n] * col[m][k]
extern "C" int spb_spn_col_wgrad(const void* G, const void* col, float* dW, int M, int N, int K, int ldcol, int lddw, spb_stream_t stream) {
  if (!G || !col || !dW || M <= 0 || N <= 0 || K <= 0) return SPB_E_ARG;
  if ((N & 7) || (ldcol & 7) || ldcol < K || lddw < K || (long long)M >= (1ll << 30)) return SPB_E_UNSUPPORTED;
  ConvGeo g;
  std::memset(&g, 0, sizeof(g));
  g.plain = 1; g.groups = 1; g.Cg = K; g.Ng = N; g.Cx = ldcol; g.Kp = lddw; g.M = M;
  g.KH = g.KW = g.OH = g.OW = g.H = g.W = g.stride = 1;
  bf16_t* zp = zero_page();
  if (!zp) return SPB_E_STATE;
  const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
  int S = (1024 + tiles - 1) / tiles;
  const int maxS = (M + 255) / 256;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int rps = ((M + S - 1) / S + 63) / 64 * 64;
  S = (M + rps - 1) / rps;
  const size_t lds = (size_t)CDS * 2 * 64 * 64 * 2;
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&spn_conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL(spn_conv_wgrad_kernel, dim3(tiles * S), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)G, (const bf16_t*)col, dW, zp, g,
                     rps, 1u, 1u);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_stem(const float* x, const void* Wb, const float* bias, void* Y, int B, int H, int W, int KH, int KW, int stride,
                            int N, int Kb, int relu, spb_stream_t stream) {
  if (!x || !Wb || !Y || B <= 0 || H < KH || W < KW || KH <= 0 || KW <= 0 || stride <= 0 || N <= 0) return SPB_E_ARG;
  if (KH != STEM_KH || KW != STEM_KW || stride != STEM_ST || N != STEM_NB * 16 || Kb != STEM_STEPS * 32) return SPB_E_UNSUPPORTED;
  const int OH = (H - KH) / stride + 1, OW = (W - KW) / stride + 1;
  if (OW != STEM_OW) return SPB_E_UNSUPPORTED;                 // the staged row width is a compile-time constant (227-wide images)
  const size_t lds = (size_t)STEM_STEPS * STEM_NB * 512 * sizeof(bf16_t) + (size_t)3 * (STEM_ST * (STEM_R - 1) + STEM_KH) * STEM_WP * sizeof(bf16_t);
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  const int nbands = (OH + STEM_R - 1) / STEM_R;
  const int total = B * nbands;
  // one workgroup per CU (136 KB of LDS); each walks ceil(total / grid) bands -- grid chosen so that they all walk the same number
  const int per = (total + 255) / 256;
  const int grid = (total + per - 1) / per;
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(&spn_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL(spn_stem_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, x, (const bf16_t*)Wb, bias, (bf16_t*)Y, B,
                     H, W, OH, OW, relu);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_pack_jobs(int dtype, const spb_spn_pack_job_t* jobs, int njobs, spb_stream_t stream) {
  if (!jobs || njobs <= 0 || njobs > SPB_SPN_MAX_PACK_JOBS || (dtype != SPB_BF16 && dtype != SPB_F32)) return SPB_E_ARG;
  PackJobs pj;
  long long biggest = 0;
  for (int i = 0; i < njobs; ++i) {
    const spb_spn_pack_job_t& q = jobs[i];
    if (!q.W || !q.out || q.Cout <= 0 || q.Cin <= 0 || q.groups <= 0 || (q.Cout % q.groups) || (q.Cin % q.groups)) return SPB_E_ARG;
    const int need = q.mode == 2 ? q.Cin * q.KH * 16 : q.KH * q.KW * (q.mode == 1 ? q.Cout / q.groups : q.Cin / q.groups);
    if (q.Kp < need || q.mode < 0 || q.mode > 2 || (q.mode == 2 && (q.groups != 1 || q.KW > 16))) return SPB_E_ARG;
    pj.j[i] = q;
    const long long total = (long long)(q.mode == 1 ? q.Cin : q.Cout) * q.Kp;
    if (total > biggest) biggest = total;
  }
  long long gx = (biggest + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3((unsigned)gx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, pj, dtype == SPB_BF16 ? 1 : 0);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_spn_pack_conv_dgrad(const float* W, void* WpD, int Cout, int Cin, int groups, int KH, int KW, int KpD, spb_stream_t stream) {
  if (!W || !WpD || Cout <= 0 || Cin <= 0 || groups <= 0 || (Cout % groups) || (Cin % groups) || KpD < KH * KW * (Cout / groups)) return SPB_E_ARG;
  long long total = (long long)Cin * KpD, gsz = (total + 255) / 256;
  if (gsz > 65535 * 4) gsz = 65535 * 4;
  hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, W, (bf16_t*)WpD, Cout, Cin, groups, KH, KW, KpD);
  SPB_CHECK_LAUNCH();
  return 0;
}
