// Style-transfer decoder of the reference's style augmentation (Ghiasi et al.), inference only.
// Reference: src/styleaug/ghiasi.py (ConvInRelu :6-23, UpsampleConvInRelu :26-59, ResidualBlock :62-104, Ghiasi :107-135),
// called from StyleAugmentor.forward (styleAugmentor.py:48-68) on half of the training batches (trainer.py:68-69).
//
// 15.4 GFLOP per 224x224 image, almost all of it in 3x3 convolutions with 32..128 channels: MFMA-bound implicit GEMMs.
// Data design (the same idea as the KRN BatchNorm layers): an instance-normalised tensor exists only as the RAW
// convolution output (NHWC bf16) plus per-(image, channel) sums; its consumer applies
//     a = act(x * scale[b,c] + shift[b,c]),   scale = gamma[b,c] * invstd[b,c],  shift = beta[b,c] - mean[b,c] * scale
// while it stages its input tile, with gamma/beta coming from the style embedding (or 1/0 for the first three layers).
//
//   gconv_kernel<NB>     KxK convolution, reflection padding, stride 1|2, optional nearest x2 upsampling of the input:
//                        an 8x8 output tile per workgroup; the (upsampled, reflected, normalised, activated) input halo is
//                        staged ONCE in LDS, then every tap is an MFMA step whose B operand is a plain 16-byte LDS read of
//                        32 input channels of the tap's pixel.  y^T = W * patch^T, so results leave the matrix core as
//                        consecutive output channels per lane (rows of W permuted at load: 4*NB consecutive channels per
//                        lane).  Weights [Cout][K*K][Cin] stream from L2 (double-buffered in registers).  The per-(image,
//                        channel) sums of the stored output are reduced with DPP butterflies and one atomic per channel.
//   conv9_rgb_kernel     first layer, 3 -> 32, 9x9 on the fp32 NCHW image: K = 243 taps gathered per pixel.
//   in_coef / style_fc / in_apply / final_sigmoid: the small per-(image, channel) and elementwise pieces.
// bf16 storage and MFMA operands, f32 accumulation.
#include "common.h"

// ablation switches for scratch/ubench_gconv.hip (0 in the product build): 1 no K loop, 2 no halo staging, 4 no weight
// slab traffic, 8 no MFMA (fragment reads only), 16 no barrier in the K loop
#ifndef GABL
#define GABL 0
#endif

#ifdef GC_TS
__device__ unsigned long long* g_gc_ts;     // scratch/ubench_gconv.hip: [workgroup][tile 0..3][4] wall clock (100 MHz)
#define GC_STAMP(i) do { if (threadIdx.x == 0 && g_gc_ts && ti < 4 && blockIdx.x < 1024) g_gc_ts[(blockIdx.x * 4 + ti) * 4 + (i)] = wall_clock64(); } while (0)
#define GC_STAMP2(i) do { if (threadIdx.x == 0 && g_gc_ts && blockIdx.y == 0 && blockIdx.x < 512) g_gc_ts[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define GC_STAMP(i) do { } while (0)
#define GC_STAMP2(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ int reflecti(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * n - 2 - i : i; }

// scale / shift of input channel c of image b: from the producer's sums when the caller passes them (spb_gconv_args_t.in_stats:
// what spb_in_coef would have written, without the launch), else from the coefficient table, else identity
__device__ __forceinline__ void in_coef_of(const float* stats, const float* gamma, const float* beta, int ld, float inv_n, float eps,
                                           int b, int C, int c, float& sc, float& sh) {
  const size_t i = (size_t)b * C + c;
  const float mean = stats[i * 2] * inv_n;
  const float var = fmaxf(stats[i * 2 + 1] * inv_n - mean * mean, 0.f);
  const float is = rsqrtf(var + eps);
  const float ga = gamma ? gamma[(size_t)b * ld + c] : 1.f, be = beta ? beta[(size_t)b * ld + c] : 0.f;
  sc = ga * is;
  sh = be - mean * ga * is;
}
__device__ __forceinline__ void gconv_coef(const spb_gconv_args_t& g, int b, int Cin, int c, float& sc, float& sh) {
  sc = 1.f; sh = 0.f;
  if (g.in_stats) in_coef_of(g.in_stats, g.in_gamma, g.in_beta, g.in_ld, g.in_inv_n, g.in_eps, b, Cin, c, sc, sh);
  else if (g.coef) { sc = g.coef[((size_t)b * Cin + c) * 2]; sh = g.coef[((size_t)b * Cin + c) * 2 + 1]; }
}

// Halo vector index -> (tile p, halo pixel hp, row hy, column hx, channel chunk cv) without integer division by run-time values
// (each one is ~35 instructions on this ISA; four of them per 16-byte vector were most of the staging cost of the narrow
// layers): Cin / 8 is a power of two, p comes from <= 3 compares, the row from a float multiply (exact for hp < 2^20 / HT).
struct HaloIdx { int p, hp, hy, hx, cv; };
template <int PXG>
__device__ __forceinline__ HaloIdx halo_split(int ic, int cvs, int HT) {
  HaloIdx h;
  const int hh = HT * HT;
  const int hpp = ic >> cvs;
  h.cv = ic & ((1 << cvs) - 1);
  h.p = 0;
#pragma unroll
  for (int q = 1; q < PXG; ++q) h.p += hpp >= q * hh ? 1 : 0;
  h.hp = hpp - h.p * hh;
  h.hy = (int)(((float)h.hp + 0.5f) * (1.0f / (float)HT));
  h.hx = h.hp - h.hy * HT;
  return h;
}

// Stage `count` halo vectors (8 channels each) of PXG tiles: upsample (nearest), reflect, normalise + activate, bf16.
// Loads are issued SU at a time before any of them is consumed (a load -> transform -> store loop exposed one L2 round
// trip per iteration: 25 iterations x ~1.5 us per workgroup in the 128-channel layers).
template <int PXG, int SU>
__device__ __forceinline__ void stage_halo(const spb_gconv_args_t& g, const bf16_t* X, bf16_t* halo, const float* cf, int b,
                                           const int* oy0, const int* ox0, int HT, int LDP, int Hu, int Wu, int st, int up,
                                           int pad, int t) {
  const int Cin = g.Cin, CV = Cin >> 3, cvsh = __ffs(CV) - 1;
  const int per = HT * HT * CV, total = PXG * per;
  for (int i0 = t; i0 < total; i0 += 256 * SU) {
    Raw8<bf16_t> r[SU];
    int dst[SU], cvs[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u;
      const int ic = i < total ? i : total - 1;
      const HaloIdx hi = halo_split<PXG>(ic, cvsh, HT);
      const int p = hi.p, hp = hi.hp, cv = hi.cv, hy = hi.hy, hx = hi.hx;
      int oyp = oy0[0], oxp = ox0[0];       // select, not oy0[p]: a dynamically indexed array lives in scratch memory
#pragma unroll
      for (int q = 1; q < PXG; ++q) { oyp = p == q ? oy0[q] : oyp; oxp = p == q ? ox0[q] : oxp; }
      const int sy = reflecti(oyp * st - pad + hy, Hu) >> (up - 1), sx = reflecti(oxp * st - pad + hx, Wu) >> (up - 1);   // up = 1 | 2
      r[u] = ldraw<bf16_t>(X + ((size_t)(b * g.Hin + sy) * g.Win + sx) * Cin + cv * 8);
      dst[u] = i < total ? (p * HT * HT + hp) * LDP + cv * 8 : -1;
      cvs[u] = cv;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      if (dst[u] < 0) continue;
      float v[8], sc[8], sh[8];
      cvt8(r[u], v);
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(cf + cvs[u] * 8 + j);
        *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(cf + Cin + cvs[u] * 8 + j);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = v[j] * sc[j] + sh[j];
        v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
      }
      st8<bf16_t>(halo + dst[u], v);
    }
  }
}

// The same in two halves for workgroups that walk several tiles: `halo_issue` puts the raw loads of ONE tile group in flight
// (at most 256 * SU vectors), `halo_commit` transforms them into the LDS tile.  Between the two the workgroup runs the previous
// tile's matrix-core loop and epilogue, so the L2 / HBM round trip of the halo is hidden (it was exposed once per tile: 28 tiles
// per workgroup in the 64->32 layer at 224x224).
template <int SU> struct HaloRegs { Raw8<bf16_t> r[SU]; int dst[SU]; int cvs[SU]; };
template <int PXG, int SU>
__device__ __forceinline__ void halo_issue(const spb_gconv_args_t& g, const bf16_t* X, HaloRegs<SU>& h, int b, const int* oy0,
                                           const int* ox0, int HT, int LDP, int Hu, int Wu, int st, int up, int pad, int t) {
  const int Cin = g.Cin, CV = Cin >> 3, cvsh = __ffs(CV) - 1;
  const int per = HT * HT * CV, total = PXG * per;
#pragma unroll
  for (int u = 0; u < SU; ++u) {
    const int i = t + 256 * u;
    const int ic = i < total ? i : total - 1;
    const HaloIdx hi = halo_split<PXG>(ic, cvsh, HT);
    const int p = hi.p, hp = hi.hp, cv = hi.cv, hy = hi.hy, hx = hi.hx;
    int oyp = oy0[0], oxp = ox0[0];
#pragma unroll
    for (int q = 1; q < PXG; ++q) { oyp = p == q ? oy0[q] : oyp; oxp = p == q ? ox0[q] : oxp; }
    const int sy = reflecti(oyp * st - pad + hy, Hu) >> (up - 1), sx = reflecti(oxp * st - pad + hx, Wu) >> (up - 1);   // up = 1 | 2
    h.r[u] = ldraw<bf16_t>(X + ((size_t)(b * g.Hin + sy) * g.Win + sx) * Cin + cv * 8);
    h.dst[u] = i < total ? (p * HT * HT + hp) * LDP + cv * 8 : -1;
    h.cvs[u] = cv;
  }
}
template <int SU>
__device__ __forceinline__ void halo_commit(const spb_gconv_args_t& g, const HaloRegs<SU>& h, bf16_t* halo, const float* cf) {
  const int Cin = g.Cin;
#pragma unroll
  for (int u = 0; u < SU; ++u) {
    if (h.dst[u] < 0) continue;
    float v[8], sc[8], sh[8];
    cvt8(h.r[u], v);
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
      *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(cf + h.cvs[u] * 8 + j);
      *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(cf + Cin + h.cvs[u] * 8 + j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float uu = v[j] * sc[j] + sh[j];
      v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
    }
    st8<bf16_t>(halo + h.dst[u], v);
  }
}

// ---------------------------------------------------------------------------------------------- implicit-GEMM conv
// WLDS: the whole weight tensor sits in LDS (<= 64 KB: the 32/64-channel layers); otherwise weights stream from L2 with a
// PD-step register prefetch.  A workgroup is persistent over `tpw` consecutive 8x8 tiles of ONE image: weights are staged
// once, and the per-(image, channel) sums stay in registers until the end (first version: one tile per workgroup, one
// prefetch step, per-wave atomics -- 9.6 M atomics and an exposed L2 round trip per tap: 5.5 ms for the 64->32 layer).
// PXG = 8x8 tiles a workgroup computes side by side (the streaming variant uses 2: every weight fragment fetched from
// L2 then feeds two MFMAs -- with one, the 128->128 layers were bound by 2.8 GB of weight re-reads per launch).
template <int NB, bool WLDS, int PXG, bool PRE = false>
__global__ __launch_bounds__(256) void gconv_kernel(const spb_gconv_args_t g, int tpw) {
  constexpr int PSU = 5;                                       // PRE: halo vectors per thread (launcher checks PXG*HT*HT*Cin/8 <= 1280)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PD = NB >= 8 ? 2 : 3;                          // weight prefetch depth (steps) when streaming
  const int Cin = g.Cin, Cout = g.Cout, KH = g.KH, st = g.stride, up = g.upsample;
  const int Hu = g.Hin * up, Wu = g.Win * up;                 // input size after upsampling
  const int Hout = Hu / st, Wout = Wu / st;
  const int pad = KH / 2;
  const int HT = 7 * st + KH;                                 // halo tile edge (upsampled coordinates)
  const int LDP = Cin + 8;                                    // bf16 per halo pixel (+16 B skew)
  const int KK = KH * KH;
  const int LDW = KK * Cin + 8;                               // bf16 per weight row in LDS (+16 B skew)
  float* cf = reinterpret_cast<float*>(smem);                 // [Cin][2]
  float* red = cf + Cin * 2;                                  // [4 waves][NB*16][2]
  bf16_t* halo = reinterpret_cast<bf16_t*>(red + 4 * NB * 16 * 2);
  bf16_t* wl = halo + PXG * HT * HT * LDP;                    // [Cout][LDW] (WLDS)
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int tiles_x = Wout >> 3, tiles_y = Hout >> 3;
  const int tpi = tiles_x * tiles_y;
  const int gpi = (tpi + PXG - 1) / PXG;                      // tile groups per image
  const int wpi = gpi / tpw;                                  // workgroups per image
  const int b = blockIdx.x / wpi;
  const int grp0 = (blockIdx.x % wpi) * tpw;
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);

  for (int c = t; c < Cin; c += 256) {
    float sc, sh;
    gconv_coef(g, b, Cin, c, sc, sh);
    cf[c] = sc; cf[Cin + c] = sh;
  }
  if (WLDS) {   // Cout rows; lanes whose (permuted) row is >= Cout use a zero fragment
    // LDS rows in fragment order [nb][li]: conflict-free fragment reads.  Eight loads in flight per thread and no integer division
    // by run-time values (the one-granule-at-a-time loop with `i / RV, i % RV` was 27 us of every workgroup of the phase kernel:
    // a third of the 64 -> 32 layer, round-3 timestamps)
    const int RV = (KK * Cin) >> 3, total = Cout * RV;
    const float inv_rv = 1.0f / (float)RV;
    for (int i0 = t; i0 < total; i0 += 256 * 8) {
      uint4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        wv[u] = *reinterpret_cast<const uint4*>(Wg + (size_t)(i < total ? i : total - 1) * 8);     // rows are contiguous in memory
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        if (i >= total) continue;
        const int r = (int)(((float)i + 0.5f) * inv_rv), v = i - r * RV;                             // exact for i < 2^20
        const int slot = ((r % (4 * NB)) >> 2) * 16 + (r / (4 * NB)) * 4 + (r & 3);
        *reinterpret_cast<uint4*>(wl + slot * LDW + v * 8) = wv[u];
      }
    }
  }
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  const int CV = Cin >> 3;
  const int prow = wave * 2 + (li >> 3), pcol = li & 7;
  const bf16_t* hbase = halo + ((prow * st) * HT + pcol * st) * LDP + lq * 8;
  const int nch = Cin >> 5;
  const int nsteps = KK * nch;
  int wco[NB];
  bool wok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = (li >> 2) * 4 * NB + nb * 4 + (li & 3);     // row permutation: see header
    wok[nb] = co < Cout;
    wco[nb] = wok[nb] ? co : 0;
  }
  const int co0 = lq * 4 * NB;
  float s1[NB][4], s2[NB][4];   // sums of the stored values over all tiles of this workgroup, reduced once at the end
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[nb][e] = 0.f; s2[nb][e] = 0.f; }

  HaloRegs<PRE ? PSU : 1> hregs;
  auto tile_origin = [&](int ti_, int* oy0_, int* ox0_, bool* tv_) {
#pragma unroll
    for (int p = 0; p < PXG; ++p) {
      int tr = (grp0 + ti_) * PXG + p;
      tv_[p] = tr < tpi;                                       // odd tile counts: the last group repeats its first tile
      tr = tv_[p] ? tr : tpi - 1;
      oy0_[p] = (tr / tiles_x) * 8; ox0_[p] = (tr % tiles_x) * 8;
    }
  };
  if constexpr (PRE) {
    int oy0[PXG], ox0[PXG]; bool tv[PXG];
    tile_origin(0, oy0, ox0, tv);
    halo_issue<PXG, PSU>(g, X, hregs, b, oy0, ox0, HT, LDP, Hu, Wu, st, up, pad, t);
  }
  for (int ti = 0; ti < tpw; ++ti) {
    int oy0[PXG], ox0[PXG];
    bool tvalid[PXG];
    tile_origin(ti, oy0, ox0, tvalid);
    GC_STAMP(0);
    lds_barrier();   // previous tile's taps are done with the halo (first pass: coefficient / weight tables written)
    // ---- stage the input halo(s)
    if constexpr (PRE) {
      if (!(GABL & 2)) halo_commit<PSU>(g, hregs, halo, cf);
      lds_barrier();
      GC_STAMP(1);
      int oyn[PXG], oxn[PXG]; bool tvn[PXG];                    // next tile's halo: in flight during this tile's taps and stores
      tile_origin(ti + 1 < tpw ? ti + 1 : ti, oyn, oxn, tvn);   // (clamped, no branch around the loads: the last one is redundant)
      halo_issue<PXG, PSU>(g, X, hregs, b, oyn, oxn, HT, LDP, Hu, Wu, st, up, pad, t);
    } else {
      stage_halo<PXG, 4>(g, X, halo, cf, b, oy0, ox0, HT, LDP, Hu, Wu, st, up, pad, t);
      lds_barrier();
    }
    // ---- K loop: taps x 32-channel chunks.  lane (li, lq): pixel li of this wave's 2x8 strip, channels lq*8..+7
    f32x4_t acc[PXG][NB];
#pragma unroll
    for (int p = 0; p < PXG; ++p)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[p][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (WLDS) {
      // Two-stage software pipeline over the taps: the fragment reads of step s+1 are issued before the MFMAs of step s.  The
      // loop is not unrollable (runtime tap count), and without this every step exposed the LDS latency (~130 cycles) in front of
      // NB * PXG MFMAs (32 cycles for the 64->32 layer): 146 of that layer's 333 us (round-2 ablation).
      const int ns = (GABL & 1) ? 0 : nsteps;
      int ky = 0, kx = 0, cc = 0;
      bf16x8_t bfc[PXG], bfn[PXG];
      uint4 auc[NB], aun[NB];
      auto frag_load = [&](bf16x8_t* bfv, uint4* auv) {
#pragma unroll
        for (int p = 0; p < PXG; ++p)
          bfv[p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + ky * HT + kx) * LDP + cc * 32);
        const int ko = (ky * KH + kx) * Cin + cc * 32 + lq * 8;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) auv[nb] = *reinterpret_cast<const uint4*>(wl + (wok[nb] ? nb * 16 + li : 0) * LDW + ko);
        if (++cc == nch) { cc = 0; if (++kx == KH) { kx = 0; ++ky; } }
      };
      auto frag_mma = [&](const bf16x8_t* bfv, const uint4* auv) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint4 au = wok[nb] ? auv[nb] : make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int p = 0; p < PXG; ++p)
            acc[p][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, au), bfv[p], acc[p][nb]);
        }
      };
      if (ns > 0) frag_load(bfc, auc);
      int s = 0;
      for (; s + 2 < ns; s += 2) {
        frag_load(bfn, aun);
        frag_mma(bfc, auc);
        frag_load(bfc, auc);
        frag_mma(bfn, aun);
      }
      if (s + 1 < ns) {            // two steps left
        frag_load(bfn, aun);
        frag_mma(bfc, auc);
        frag_mma(bfn, aun);
      } else if (s < ns) {
        frag_mma(bfc, auc);
      }
    } else {
      uint4 an[PD][NB];
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          an[d][nb] = *reinterpret_cast<const uint4*>(Wg + (size_t)wco[nb] * KK * Cin + lq * 8 + (size_t)(d < nsteps ? d : 0) * 32);
      int ky = 0, kx = 0, cc = 0;
      for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const int s = s0 + d;
          if (s < nsteps) {
            uint4 a[NB];   // streaming path: Cout == 16*NB (checked by the launcher), every row is real
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = an[d][nb];
            bf16x8_t bf[PXG];
#pragma unroll
            for (int p = 0; p < PXG; ++p)
              bf[p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + ky * HT + kx) * LDP + cc * 32);
            const int sn = s + PD;   // weights are [co][tap][ci]: step index * 32 is the offset inside a row
            if (sn < nsteps) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                an[d][nb] = *reinterpret_cast<const uint4*>(Wg + (size_t)wco[nb] * KK * Cin + lq * 8 + (size_t)sn * 32);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
              for (int p = 0; p < PXG; ++p)
                acc[p][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, a[nb]), bf[p], acc[p][nb]);
            if (++cc == nch) { cc = 0; if (++kx == KH) { kx = 0; ++ky; } }
          }
        }
      }
    }
    GC_STAMP(2);
    // ---- epilogue: lane (li = pixel, lq): channels lq*4*NB + nb*4 + e -- one contiguous run, 16-byte stores
#pragma unroll
    for (int p = 0; p < PXG; ++p) {
      if (!tvalid[p] || (GABL & 64)) continue;
      const int oy = oy0[p] + prow, ox = ox0[p] + pcol;
      bf16_t* dst = Y + ((size_t)(b * Hout + oy) * Wout + ox) * g.ldc + co0;
      uint2 o[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + nb * 4 + e;
          v[e] = acc[p][nb][e] + ((g.bias && co < Cout) ? g.bias[co] : 0.f);
        }
        o[nb].x = pack_bf16x2(v[0], v[1]); o[nb].y = pack_bf16x2(v[2], v[3]);
        float r[4];
        spb_unpack2(o[nb].x, r[0], r[1]); spb_unpack2(o[nb].y, r[2], r[3]);   // (16-bit storage format: common.h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[nb][e] += r[e]; s2[nb][e] += r[e] * r[e]; }
      }
      if constexpr ((NB & 1) == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; nb += 2)
          if (co0 + nb * 4 < g.ldc) *reinterpret_cast<uint4*>(dst + nb * 4) = make_uint4(o[nb].x, o[nb].y, o[nb + 1].x, o[nb + 1].y);
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (co0 + nb * 4 < g.ldc) *reinterpret_cast<uint2*>(dst + nb * 4) = o[nb];
      }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = row16_sum(s1[nb][e]), a2 = row16_sum(s2[nb][e]);   // DPP butterflies, not ds_bpermute
        if (li == 0) {
          red[(wave * NB * 16 + co0 + nb * 4 + e) * 2] = a1;
          red[(wave * NB * 16 + co0 + nb * 4 + e) * 2 + 1] = a2;
        }
      }
  }
  if (g.stats) {   // 4 waves -> one atomic per (image, channel, moment) and workgroup
    lds_barrier();
    for (int i = t; i < NB * 16 * 2; i += 256) {
      const int co = i >> 1;
      if (co < Cout)
        atomicAdd(g.stats + ((size_t)b * Cout + co) * 2 + (i & 1),
                  red[i] + red[NB * 32 + i] + red[2 * NB * 32 + i] + red[3 * NB * 32 + i]);
    }
  }
}

// ------------------------------------------------------------------- nearest x2 upsampling + 3x3 convolution, by phase
// UpsampleConvInRelu (ghiasi.py:46-59): Upsample(scale 2, nearest) -> ReflectionPad2d(1) -> Conv2d 3x3.  Output pixel
// (2i + py, 2j + px) sees only a 2x2 block of LOW-RESOLUTION pixels: along an axis the taps (o-1, o, o+1) of o = 2i + p land on
// rows (i-1, i, i) for p = 0 and (i, i, i+1) for p = 1, so the 3x3 kernel collapses to a 2x2 one per output phase with summed
// weights (w0, w1+w2) / (w0+w1, w2) -- and the reflection of the upsampled image at its border (-1 -> 1, i.e. low-res row 0)
// is exactly CLAMPING the low-resolution index.  9 taps become 4 (2.25x fewer matrix-core steps), the halo of an 8x8 output tile
// is 6x6 low-resolution pixels instead of 10x10 upsampled ones, and no pixel is staged twice.  A wave owns one phase: its 16
// pixels are the 4x4 low-resolution positions of the tile, its A operand that phase's weights (packed by the host side:
// [phase][Cout][tap 2x2][Cin], sums taken in float32).
// NV > 0: the halo vectors of a tile group are exactly <= NV per thread and the NEXT group's are loaded into registers right
// after this group's halo is committed (in flight during the matrix-core loop and the stores).  Without it every group exposed
// one L2 / HBM round trip: 5.2 us per 128-pixel group of the 64 -> 32 layer for 0.43 us of matrix-core work (round 3).
// WREG (Cin == 64, NB == 2: the 64 -> 32 layer, the largest launch of the decoder after the residual blocks): a wave's phase weights
// are 8 reduction steps x 2 fragments = 64 registers, loaded ONCE per workgroup and kept for all its tile groups.  The tap loop
// then has no weight traffic at all and is fully unrolled: 16 pixel-fragment reads in two batches, 32 MFMAs (timestamps, round 3:
// the LDS-resident version spent 1.55 us per 128-pixel group in its 8-step loop for 0.43 us of matrix-core work -- one LDS round
// trip per step -- and its 67 KB of weights per workgroup held the CU at two workgroups; without them LDS is 11 KB).
// The same for the 128 -> 64 layer (NB == 4, Cin == 128): 16 steps x 4 fragments = 256 registers of weights per wave, one
// workgroup per CU -- four waves that never wait for a weight: the streamed version spent 12.9 us per 256-pixel group in its tap
// loop (two steps of weights in flight per wave) for 1.7 us of matrix-core work.
template <int NB, bool WLDS, int PXG, int NV = 0, bool WREG = false, int WSTEPS = 8>
__global__ __launch_bounds__(256, (WREG && NB == 2) ? 3 : 1) void gconv_up2_kernel(const spb_gconv_args_t g, int tpw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PD = 2, HT = 6, KT = 4, SU = 4;
  const int Cin = g.Cin, Cout = g.Cout;
  const int Hout = 2 * g.Hin, Wout = 2 * g.Win;
  const int LDP = Cin + 8;
  const int LDW = KT * Cin + 8;
  float* cf = reinterpret_cast<float*>(smem);                 // [Cin][2]
  float* red = cf + Cin * 2;                                  // [4 waves][NB*16][2]
  bf16_t* halo = reinterpret_cast<bf16_t*>(red + 4 * NB * 16 * 2);   // [PXG][36][LDP]
  bf16_t* wl = halo + PXG * HT * HT * LDP;                    // [4 phases][NB*16 slots][LDW] (WLDS)
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int tiles_x = Wout >> 3, tiles_y = Hout >> 3;
  const int tpi = tiles_x * tiles_y;
  const int gpi = (tpi + PXG - 1) / PXG;
  const int wpi = gpi / tpw;
  const int b = blockIdx.x / wpi;
  const int grp0 = (blockIdx.x % wpi) * tpw;
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);
  const int ph = wave, py = wave >> 1, px = wave & 1;

  for (int c = t; c < Cin; c += 256) {
    float sc, sh;
    gconv_coef(g, b, Cin, c, sc, sh);
    cf[c] = sc; cf[Cin + c] = sh;
  }
  if (WLDS) {   // rows in fragment order [nb][li] per phase: conflict-free fragment reads; staged as in gconv_kernel
    const int RV = (KT * Cin) >> 3, total = 4 * Cout * RV;
    const float inv_rv = 1.0f / (float)RV, inv_co = 1.0f / (float)Cout;
    for (int i0 = t; i0 < total; i0 += 256 * 8) {
      uint4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        wv[u] = *reinterpret_cast<const uint4*>(Wg + (size_t)(i < total ? i : total - 1) * 8);     // [phase][row][tap][Cin]: contiguous
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        if (i >= total) continue;
        const int ra = (int)(((float)i + 0.5f) * inv_rv), v = i - ra * RV;                           // row over all phases
        const int p = (int)(((float)ra + 0.5f) * inv_co), r = ra - p * Cout;
        const int slot = ((r % (4 * NB)) >> 2) * 16 + (r / (4 * NB)) * 4 + (r & 3);
        *reinterpret_cast<uint4*>(wl + (size_t)(p * NB * 16 + slot) * LDW + v * 8) = wv[u];
      }
    }
  }
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  const int CV = Cin >> 3, cvsh = __ffs(CV) - 1;
  const int iy = li >> 2, ix = li & 3;                        // this lane's low-resolution position in the 4x4 block
  const bf16_t* hbase = halo + ((iy + py) * HT + (ix + px)) * LDP + lq * 8;
  const int nch = Cin >> 5;
  const int nsteps = KT * nch;
  int wco[NB];
  bool wok[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = (li >> 2) * 4 * NB + nb * 4 + (li & 3);
    wok[nb] = co < Cout;
    wco[nb] = wok[nb] ? co : 0;
  }
  const int co0 = lq * 4 * NB;
  float s1[NB][4], s2[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[nb][e] = 0.f; s2[nb][e] = 0.f; }

  auto tile_origin = [&](int ti_, int* oy0_, int* ox0_, bool* tv_) {
#pragma unroll
    for (int p = 0; p < PXG; ++p) {
      int tr = (grp0 + ti_) * PXG + p;
      tv_[p] = tr < tpi;
      tr = tv_[p] ? tr : tpi - 1;
      oy0_[p] = (tr / tiles_x) * 8; ox0_[p] = (tr % tiles_x) * 8;
    }
  };
  Raw8<bf16_t> pre[NV > 0 ? NV : 1];
  auto pre_issue = [&](int ti_) {
    int oyq[PXG], oxq[PXG]; bool tvq[PXG];
    tile_origin(ti_, oyq, oxq, tvq);
    const int per = HT * HT * CV, total = PXG * per;
#pragma unroll
    for (int u = 0; u < (NV > 0 ? NV : 1); ++u) {
      const int i = t + 256 * u;
      const int ic = i < total ? i : total - 1;
      const int hpp = ic >> cvsh, cv = ic & (CV - 1);
      const int p = hpp / (HT * HT), hp = hpp - p * (HT * HT);      // HT is a compile-time 6 here
      const int hy = hp / HT, hx = hp % HT;
      int oyp = oyq[0], oxp = oxq[0];
#pragma unroll
      for (int q = 1; q < PXG; ++q) { oyp = p == q ? oyq[q] : oyp; oxp = p == q ? oxq[q] : oxp; }
      const int sy = min(max((oyp >> 1) - 1 + hy, 0), g.Hin - 1), sx = min(max((oxp >> 1) - 1 + hx, 0), g.Win - 1);
      pre[u] = ldraw<bf16_t>(X + ((size_t)(b * g.Hin + sy) * g.Win + sx) * Cin + cv * 8);
    }
  };
  auto pre_commit = [&]() {
    const int per = HT * HT * CV, total = PXG * per;
#pragma unroll
    for (int u = 0; u < (NV > 0 ? NV : 1); ++u) {
      const int i = t + 256 * u;
      if (i >= total) continue;
      const int hpp = i >> cvsh, cv = i & (CV - 1);
      const int p = hpp / (HT * HT), hp = hpp - p * (HT * HT);
      float v[8], sc[8], sh[8];
      cvt8(pre[u], v);
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(cf + cv * 8 + j);
        *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(cf + Cin + cv * 8 + j);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = v[j] * sc[j] + sh[j];
        v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
      }
      st8<bf16_t>(halo + (p * HT * HT + hp) * LDP + cv * 8, v);
    }
  };
  if constexpr (NV > 0) pre_issue(0);
  constexpr int WNCH = WSTEPS / 4;                    // 32-channel chunks per tap (Cin / 32)
  uint4 areg[WREG ? WSTEPS : 1][NB];
  if constexpr (WREG) {     // step s = (ky * 2 + kx) * WNCH + cc, as the tap loops below walk it
#pragma unroll
    for (int s_ = 0; s_ < WSTEPS; ++s_) {
      const int ko = (s_ / WNCH) * Cin + (s_ % WNCH) * 32 + lq * 8;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint4 w = *reinterpret_cast<const uint4*>(Wg + ((size_t)(ph * Cout + wco[nb]) * KT) * Cin + ko);
        areg[s_][nb] = wok[nb] ? w : make_uint4(0, 0, 0, 0);
      }
    }
  }

  for (int ti = 0; ti < tpw; ++ti) {
    int oy0[PXG], ox0[PXG];
    bool tvalid[PXG];
    tile_origin(ti, oy0, ox0, tvalid);
    GC_STAMP(0);
    lds_barrier();   // previous tile's taps are done with the halo (first pass: coefficient / weight tables written)
    if constexpr (NV > 0) {
      pre_commit();
    } else {           // ---- low-resolution halo(s): 6x6 pixels per tile, index clamped, normalised + activated on the way in
      const int per = HT * HT * CV, total = PXG * per;
      for (int i0 = t; i0 < total; i0 += 256 * SU) {
        Raw8<bf16_t> r[SU];
        int dst[SU], cvs[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = i0 + 256 * u;
          const int ic = i < total ? i : total - 1;
          const int hpp = ic >> cvsh, cv = ic & (CV - 1);
          const int p = hpp / (HT * HT), hp = hpp - p * (HT * HT);
          const int hy = hp / HT, hx = hp % HT;
          int oyp = oy0[0], oxp = ox0[0];
#pragma unroll
          for (int q = 1; q < PXG; ++q) { oyp = p == q ? oy0[q] : oyp; oxp = p == q ? ox0[q] : oxp; }
          const int sy = min(max((oyp >> 1) - 1 + hy, 0), g.Hin - 1), sx = min(max((oxp >> 1) - 1 + hx, 0), g.Win - 1);
          r[u] = ldraw<bf16_t>(X + ((size_t)(b * g.Hin + sy) * g.Win + sx) * Cin + cv * 8);
          dst[u] = i < total ? (p * HT * HT + hp) * LDP + cv * 8 : -1;
          cvs[u] = cv;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (dst[u] < 0) continue;
          float v[8], sc[8], sh[8];
          cvt8(r[u], v);
#pragma unroll
          for (int j = 0; j < 8; j += 4) {
            *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(cf + cvs[u] * 8 + j);
            *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(cf + Cin + cvs[u] * 8 + j);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float uu = v[j] * sc[j] + sh[j];
            v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
          }
          st8<bf16_t>(halo + dst[u], v);
        }
      }
    }
    lds_barrier();
    GC_STAMP(1);
    if constexpr (NV > 0) pre_issue(ti + 1 < tpw ? ti + 1 : ti);   // clamped, no branch around the loads: the last one is redundant
    f32x4_t acc[PXG][NB];
#pragma unroll
    for (int p = 0; p < PXG; ++p)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[p][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (WREG) {
      constexpr int BATCH = PXG >= 4 ? 2 : 4;         // steps whose pixel fragments are read together (8 fragments)
#pragma unroll
      for (int s0 = 0; s0 < WSTEPS; s0 += BATCH) {
        bf16x8_t bf[BATCH][PXG];
#pragma unroll
        for (int sb = 0; sb < BATCH; ++sb) {
          const int s_ = s0 + sb, tap = s_ / WNCH;   // ky = tap / 2, kx = tap % 2
#pragma unroll
          for (int p = 0; p < PXG; ++p)
            bf[sb][p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + (tap >> 1) * HT + (tap & 1)) * LDP + (s_ % WNCH) * 32);
        }
#pragma unroll
        for (int sb = 0; sb < BATCH; ++sb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int p = 0; p < PXG; ++p)
              acc[p][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, areg[s0 + sb][nb]), bf[sb][p], acc[p][nb]);
      }
    } else if constexpr (WLDS) {
      int ky = 0, kx = 0, cc = 0;
      bf16x8_t bfc[PXG], bfn[PXG];
      uint4 auc[NB], aun[NB];
      const bf16_t* wph = wl + (size_t)ph * NB * 16 * LDW;
      auto frag_load = [&](bf16x8_t* bfv, uint4* auv) {
#pragma unroll
        for (int p = 0; p < PXG; ++p)
          bfv[p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + ky * HT + kx) * LDP + cc * 32);
        const int ko = (ky * 2 + kx) * Cin + cc * 32 + lq * 8;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) auv[nb] = *reinterpret_cast<const uint4*>(wph + (wok[nb] ? nb * 16 + li : 0) * LDW + ko);
        if (++cc == nch) { cc = 0; if (++kx == 2) { kx = 0; ++ky; } }
      };
      auto frag_mma = [&](const bf16x8_t* bfv, const uint4* auv) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint4 au = wok[nb] ? auv[nb] : make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int p = 0; p < PXG; ++p)
            acc[p][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, au), bfv[p], acc[p][nb]);
        }
      };
      frag_load(bfc, auc);
      int s = 0;
      for (; s + 2 < nsteps; s += 2) {
        frag_load(bfn, aun);
        frag_mma(bfc, auc);
        frag_load(bfc, auc);
        frag_mma(bfn, aun);
      }
      if (s + 1 < nsteps) {
        frag_load(bfn, aun);
        frag_mma(bfc, auc);
        frag_mma(bfn, aun);
      } else if (s < nsteps) {
        frag_mma(bfc, auc);
      }
    } else {
      const bf16_t* wrow[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wrow[nb] = Wg + ((size_t)(ph * Cout + wco[nb]) * KT) * Cin + lq * 8;
      uint4 an[PD][NB];
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) an[d][nb] = *reinterpret_cast<const uint4*>(wrow[nb] + (size_t)(d < nsteps ? d : 0) * 32);
      int ky = 0, kx = 0, cc = 0;
      for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const int s = s0 + d;
          if (s < nsteps) {
            uint4 a[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = an[d][nb];
            bf16x8_t bf[PXG];
#pragma unroll
            for (int p = 0; p < PXG; ++p)
              bf[p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + ky * HT + kx) * LDP + cc * 32);
            const int sn = s + PD;
            if (sn < nsteps) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) an[d][nb] = *reinterpret_cast<const uint4*>(wrow[nb] + (size_t)sn * 32);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
              for (int p = 0; p < PXG; ++p)
                acc[p][nb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, a[nb]), bf[p], acc[p][nb]);
            if (++cc == nch) { cc = 0; if (++kx == 2) { kx = 0; ++ky; } }
          }
        }
      }
    }
    GC_STAMP(2);
    // ---- epilogue: lane (li = low-resolution position, lq): channels lq*4*NB + nb*4 + e of output pixel (2 iy + py, 2 ix + px)
#pragma unroll
    for (int p = 0; p < PXG; ++p) {
      if (!tvalid[p]) continue;
      const int oy = oy0[p] + 2 * iy + py, ox = ox0[p] + 2 * ix + px;
      bf16_t* dst = Y + ((size_t)(b * Hout + oy) * Wout + ox) * g.ldc + co0;
      uint2 o[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + nb * 4 + e;
          v[e] = acc[p][nb][e] + ((g.bias && co < Cout) ? g.bias[co] : 0.f);
        }
        o[nb].x = pack_bf16x2(v[0], v[1]); o[nb].y = pack_bf16x2(v[2], v[3]);
        float r[4];
        spb_unpack2(o[nb].x, r[0], r[1]); spb_unpack2(o[nb].y, r[2], r[3]);   // (16-bit storage format: common.h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[nb][e] += r[e]; s2[nb][e] += r[e] * r[e]; }
      }
      if constexpr ((NB & 1) == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; nb += 2)
          if (co0 + nb * 4 < g.ldc) *reinterpret_cast<uint4*>(dst + nb * 4) = make_uint4(o[nb].x, o[nb].y, o[nb + 1].x, o[nb + 1].y);
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (co0 + nb * 4 < g.ldc) *reinterpret_cast<uint2*>(dst + nb * 4) = o[nb];
      }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = row16_sum(s1[nb][e]), a2 = row16_sum(s2[nb][e]);
        if (li == 0) {
          red[(wave * NB * 16 + co0 + nb * 4 + e) * 2] = a1;
          red[(wave * NB * 16 + co0 + nb * 4 + e) * 2 + 1] = a2;
        }
      }
    lds_barrier();
    for (int i = t; i < NB * 16 * 2; i += 256) {
      const int co = i >> 1;
      if (co < Cout)
        atomicAdd(g.stats + ((size_t)b * Cout + co) * 2 + (i & 1),
                  red[i] + red[NB * 32 + i] + red[2 * NB * 32 + i] + red[3 * NB * 32 + i]);
    }
  }
}

// ------------------------------------------------------------------------------------ implicit-GEMM conv, wide layers
// The 128-channel layers (weights 147..295 KB: not LDS resident).  Each wave computes FOUR 8x8 tiles' worth of its two
// rows (64 pixels x all output channels): every weight fragment read feeds 4 MFMAs and every pixel fragment NB of them,
// so per 32-channel step a wave issues 32 MFMAs (512 matrix-core cycles) for 12 KB of LDS reads.  The step's weight slab
// [Cout x 32] is fetched from L2 ONCE per workgroup (cooperatively, through registers, double-buffered in LDS; one barrier
// per step) instead of once per wave: the per-wave variant above was bound by L1 bandwidth at 149 TFLOP/s.
// PXG = 8x8 tiles per workgroup.  4: 130 KB of LDS, one workgroup per CU -- halo staging, the reduction loop and the epilogue
// of a workgroup overlap with nothing (round 1: 214 TFLOP/s over the whole decoder).  2: 75 KB and <= 256 registers, TWO
// workgroups per CU: one is in its matrix-core loop while the other stages its halo or stores its tile; a weight fragment then
// feeds 2 MFMAs per wave instead of 4 (10 instead of 12 LDS reads per 16 instead of 32 MFMAs: 147 B/clk/CU, under the 256 peak).
// PF = weight slabs in flight per workgroup (global -> registers -> LDS).  The ablation of round 2 (scratch/ubench_gconv.hip, 128->128
// at 56x56, B=48, PXG=2: 151 us) puts 73 us on this stream and 7 us on the matrix cores: with 3 slabs (24 KB) in flight per
// workgroup the L2 round trip (~1.5 us under load) bounds the stream at ~16 GB/s per workgroup, and every workgroup needs the
// whole 295 KB weight tensor.
#ifndef SLAB_SU
#define SLAB_SU 5      // halo vectors a thread has in flight per round trip
#endif
template <int NB, int PXG, int PF>
__global__ __launch_bounds__(256, PXG >= 4 ? 1 : 2) void gconv_slab_kernel(const spb_gconv_args_t g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NB * 16, SLD = 40;                      // slab row = 32 k + 8 pad (bf16)
  constexpr int SPT = (ROWS * 4 + 255) / 256;                  // 16-byte slab granules per thread
  const int Cin = g.Cin, Cout = g.Cout, KH = g.KH, st = g.stride, up = g.upsample;
  const int Hu = g.Hin * up, Wu = g.Win * up;
  const int Hout = Hu / st, Wout = Wu / st;
  const int pad = KH / 2;
  const int HT = 7 * st + KH;
  const int LDP = Cin + 8;
  const int KK = KH * KH;
  float* cf = reinterpret_cast<float*>(smem);                 // [Cin][2]
  float* red = cf + Cin * 2;                                  // [4 waves][ROWS][2]
  bf16_t* slab = reinterpret_cast<bf16_t*>(red + 4 * ROWS * 2);   // [2][ROWS][SLD]
  bf16_t* halo = slab + 2 * ROWS * SLD;                       // [PXG][HT*HT][LDP]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int tiles_x = Wout >> 3, tiles_y = Hout >> 3;
  const int tpi = tiles_x * tiles_y;
  const int gpi = (tpi + PXG - 1) / PXG;
  const int b = blockIdx.x / gpi;
  const int grp = blockIdx.x % gpi;
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);

  for (int c = t; c < Cin; c += 256) {
    float sc, sh;
    gconv_coef(g, b, Cin, c, sc, sh);
    cf[c] = sc; cf[Cin + c] = sh;
  }
  for (int i = t; i < 4 * ROWS * 2; i += 256) red[i] = 0.f;
  int oy0[PXG], ox0[PXG];
  bool tvalid[PXG];
#pragma unroll
  for (int p = 0; p < PXG; ++p) {
    int tr = grp * PXG + p;
    tvalid[p] = tr < tpi;
    tr = tvalid[p] ? tr : tpi - 1;
    oy0[p] = (tr / tiles_x) * 8; ox0[p] = (tr % tiles_x) * 8;
  }
  // slab granules of this thread: row, 16-byte part
  int srow[SPT], spart[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int gidx = t + 256 * j;
    srow[j] = (gidx >> 2) < ROWS ? (gidx >> 2) : ROWS - 1; spart[j] = gidx & 3;
  }
  // LDS slot of weight row r: fragment order [nb][li] (li = (r / 4NB) * 4 + r % 4, nb = (r % 4NB) / 4), so that the 16 rows
  // of one A fragment are contiguous (with the natural order they are 4NB rows apart = a multiple of 256 B: 4-way bank
  // conflicts on every fragment read)
  int sslot[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int r = srow[j];
    sslot[j] = (((r % (4 * NB)) >> 2) * 16 + (r / (4 * NB)) * 4 + (r & 3)) * SLD + spart[j] * 8;
  }
  const int nch = Cin >> 5;
  const int nsteps = (GABL & 1) ? 0 : KK * nch;
#ifdef GSLABMAJOR   // experiment: weights stored slab-major [step][row][32] (a slab = 8 KB contiguous)
#define SLAB_OFF(row, step, part) (((size_t)(step) * ROWS + (row)) * 32 + (part) * 8)
#else
#define SLAB_OFF(row, step, part) ((size_t)(row) * KK * Cin + (size_t)(step) * 32 + (part) * 8)
#endif
  // weight slabs travel global -> registers -> LDS three steps ahead (one step ahead left the L2 round trip exposed on
  // every step: 36 x ~1.5 us per workgroup, 10 % matrix-core utilisation); nsteps is a multiple of 3 (9 taps x Cin/32)
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t sreg[PF * SPT];   // flat, native vectors, constant indices only: stays in registers (uint4 [PF][SPT] lived in scratch)
#pragma unroll
  for (int d = 0; d < PF; ++d)
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      sreg[d * SPT + j] = *reinterpret_cast<const u32x4_t*>(Wg + SLAB_OFF(srow[j], d, spart[j]));
  lds_barrier();
  // ---- stage the four input halos
  if (!(GABL & 2)) stage_halo<PXG, SLAB_SU>(g, X, halo, cf, b, oy0, ox0, HT, LDP, Hu, Wu, st, up, pad, t);
#pragma unroll
  for (int j = 0; j < SPT; ++j) *reinterpret_cast<u32x4_t*>(slab + sslot[j]) = sreg[j];
  lds_barrier();

  const int prow = wave * 2 + (li >> 3), pcol = li & 7;
  const bf16_t* hbase = halo + ((prow * st) * HT + pcol * st) * LDP + lq * 8;
  int arow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) arow[nb] = (nb * 16 + li) * SLD + lq * 8;
  f32x4_t acc[PXG][NB];
#pragma unroll
  for (int p = 0; p < PXG; ++p)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[p][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int ky = 0, kx = 0, cc = 0;
  for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int s = s0 + d;
      const bf16_t* sl = slab + (s & 1) * ROWS * SLD;
      bf16x8_t bf[PXG];
#pragma unroll
      for (int p = 0; p < PXG; ++p)
        bf[p] = *reinterpret_cast<const bf16x8_t*>(hbase + (p * HT * HT + ky * HT + kx) * LDP + cc * 32);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(sl + arow[nb]);
#pragma unroll
        for (int p = 0; p < PXG; ++p) {
          if (GABL & 8) acc[p][nb][0] += (float)af[0] * (float)bf[p][0];
          else acc[p][nb] = SPB_MFMA16(af, bf[p], acc[p][nb]);
        }
      }
      // NO branch around these stores and loads (ROWS * 4 is a multiple of 256; the step index is clamped, the last steps
      // re-fetch the last slab and park it in the idle buffer): with `if (s + PF < nsteps)` / `if (t + 256 j < ROWS * 4)` around
      // them the compiler waited vmcnt(0) before every slab store -- the PF-deep prefetch collapsed to one exposed L2 round
      // trip per step (36 x ~2 us = 71 of the 151 us of a 128->128 layer, round-2 ablation)
      static_assert((ROWS * 4) % 256 == 0, "slab granules must tile the workgroup");
      if (!(GABL & 4)) {   // slab of step s+1 (fetched PF-1 steps ago) -> the other LDS buffer
        bf16_t* sn = slab + ((s + 1) & 1) * ROWS * SLD;
#pragma unroll
        for (int j = 0; j < SPT; ++j) *reinterpret_cast<u32x4_t*>(sn + sslot[j]) = sreg[((d + 1) % PF) * SPT + j];
        const int sf = s + PF < nsteps ? s + PF : nsteps - 1;   // the fetch of step s+PF takes the register set step s released
#pragma unroll
        for (int j = 0; j < SPT; ++j)
          sreg[d * SPT + j] = *reinterpret_cast<const u32x4_t*>(Wg + SLAB_OFF(srow[j], sf, spart[j]));
      }
      if (!(GABL & 16)) lds_barrier();
      if (++cc == nch) { cc = 0; if (++kx == KH) { kx = 0; ++ky; } }
    }
  }
  // ---- epilogue: each lane owns 4*NB consecutive channels of its pixel -> 16-byte stores (8-byte stores, one per
  // accumulator, cost 62 us of the 235 us launch: 64 scattered 8-byte pieces per instruction); the sums of the stored
  // values are accumulated over the four tiles first and reduced once (per-tile butterflies: 41 us)
  const int co0 = lq * 4 * NB;
  float s1[NB][4], s2[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[nb][e] = 0.f; s2[nb][e] = 0.f; }
#pragma unroll
  for (int p = 0; p < PXG; ++p) {
    if (!tvalid[p] || (GABL & 64)) continue;
    const int oy = oy0[p] + prow, ox = ox0[p] + pcol;
    bf16_t* dst = Y + ((size_t)(b * Hout + oy) * Wout + ox) * g.ldc + co0;
    uint2 o[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[p][nb][e] + (g.bias ? g.bias[co0 + nb * 4 + e] : 0.f);
      o[nb].x = pack_bf16x2(v[0], v[1]); o[nb].y = pack_bf16x2(v[2], v[3]);
      float r[4];
        spb_unpack2(o[nb].x, r[0], r[1]); spb_unpack2(o[nb].y, r[2], r[3]);   // (16-bit storage format: common.h)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[nb][e] += r[e]; s2[nb][e] += r[e] * r[e]; }
    }
#pragma unroll
    for (int nb = 0; nb < NB; nb += 2)
      *reinterpret_cast<uint4*>(dst + nb * 4) = make_uint4(o[nb].x, o[nb].y, o[nb + 1].x, o[nb + 1].y);
  }
  if (g.stats && !(GABL & 32)) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = row16_sum(s1[nb][e]), a2 = row16_sum(s2[nb][e]);   // DPP butterflies, not ds_bpermute
        if (li == 0) {
          red[(wave * ROWS + co0 + nb * 4 + e) * 2] = a1;
          red[(wave * ROWS + co0 + nb * 4 + e) * 2 + 1] = a2;
        }
      }
  }
  if (g.stats) {
    lds_barrier();
    for (int i = t; i < ROWS * 2; i += 256)
      atomicAdd(g.stats + ((size_t)b * Cout + (i >> 1)) * 2 + (i & 1),
                red[i] + red[ROWS * 2 + i] + red[2 * ROWS * 2 + i] + red[3 * ROWS * 2 + i]);
  }
}

// ------------------------------------------------------------------------------------------- last layer: 32 -> 3, 9x9
// (ghiasi.py:70, the convolution before the final instance norm + sigmoid).  The generic 8x8-tile kernel stages a 16x16 halo per
// 64 output pixels (4x the input) and reads a weight fragment per MFMA.  Here a workgroup owns a band of C9O_R rows x 32 columns:
// the halo is 2.5 staged pixels per output pixel, the <= 4 weight rows sit in LDS, and each wave keeps four 16-pixel groups in
// flight per weight-fragment read (81 taps x (1 + 4) 16-byte LDS reads x 4 MFMAs).
constexpr int C9O_R = 8, C9O_W = 32;
__global__ __launch_bounds__(256) void conv9_band_kernel(const spb_gconv_args_t g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HW_ = C9O_W + 8, HR = C9O_R + 8, LDP = 40, KK = 81, LDW = KK * 32 + 8;
  float* cf = reinterpret_cast<float*>(smem);                      // [32][2] scale | shift
  float* red = cf + 64;                                            // [4 waves][4 ch][2]
  bf16_t* halo = reinterpret_cast<bf16_t*>(red + 32);              // [HR][HW_][LDP]
  bf16_t* wl = halo + HR * HW_ * LDP;                              // [4][LDW]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int H = g.Hin, W = g.Win, Cout = g.Cout;
  const int bx = W / C9O_W;
  const int b = blockIdx.y, y0 = (blockIdx.x / bx) * C9O_R, x0 = (blockIdx.x % bx) * C9O_W;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  if (t < 32) {
    float sc, sh;
    gconv_coef(g, b, 32, t, sc, sh);
    cf[t] = sc; cf[32 + t] = sh;
  }
  for (int i = t; i < 4 * (KK * 4); i += 256) {                    // weight rows (zero rows past Cout), 16-byte granules
    const int r = i / (KK * 4), v = i % (KK * 4);
    uint4 u = make_uint4(0, 0, 0, 0);
    if (r < Cout) u = *reinterpret_cast<const uint4*>(Wg + (size_t)r * KK * 32 + v * 8);
    *reinterpret_cast<uint4*>(wl + r * LDW + v * 8) = u;
  }
  lds_barrier();
  // halo: HR x HW_ pixels x 4 channel vectors, reflection + instance norm + style affine + ReLU on the way in, 5 loads in flight
  constexpr int total = HR * HW_ * 4;
  for (int i0 = t; i0 < total; i0 += 256 * 5) {
    Raw8<bf16_t> r[5];
    int dst[5], cvs[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = i0 + 256 * u, ic = i < total ? i : total - 1;
      const int cv = ic & 3, hp = ic >> 2, hy = hp / HW_, hx = hp % HW_;
      const int sy = reflecti(y0 - 4 + hy, H), sx = reflecti(x0 - 4 + hx, W);
      r[u] = ldraw<bf16_t>(X + ((size_t)(b * H + sy) * W + sx) * 32 + cv * 8);
      dst[u] = i < total ? hp * LDP + cv * 8 : -1;
      cvs[u] = cv;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      if (dst[u] < 0) continue;
      float v[8];
      cvt8(r[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = v[j] * cf[cvs[u] * 8 + j] + cf[32 + cvs[u] * 8 + j];
        v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
      }
      st8<bf16_t>(halo + dst[u], v);
    }
  }
  lds_barrier();
  // wave w: rows 2w, 2w+1 x two 16-column groups
  f32x4_t acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16_t* hb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) hb[p] = halo + ((wave * 2 + (p >> 1)) * HW_ + (p & 1) * 16 + li) * LDP + lq * 8;
  const bf16_t* wrow = wl + (li < 4 ? li : 0) * LDW + lq * 8;
  const bool wok = li < 4;
  for (int ky = 0; ky < 9; ++ky)
#pragma unroll
    for (int kx = 0; kx < 9; ++kx) {
      uint4 au = make_uint4(0, 0, 0, 0);
      if (wok) au = *reinterpret_cast<const uint4*>(wrow + (ky * 9 + kx) * 32);
      const bf16x8_t af = __builtin_bit_cast(bf16x8_t, au);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(hb[p] + (ky * HW_ + kx) * LDP);
        acc[p] = SPB_MFMA16(af, bf, acc[p]);
      }
    }
  // epilogue: lanes lq == 0 hold channels 0..3 of their pixel
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (lq == 0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int oy = y0 + wave * 2 + (p >> 1), ox = x0 + (p & 1) * 16 + li;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[p][e] + ((g.bias && e < Cout) ? g.bias[e] : 0.f);
      uint2 o;
      o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(Y + ((size_t)(b * H + oy) * W + ox) * g.ldc) = o;
      float r[4];
        spb_unpack2(o.x, r[0], r[1]); spb_unpack2(o.y, r[2], r[3]);   // (16-bit storage format: common.h)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += r[e]; s2[e] += r[e] * r[e]; }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a1 = row16_sum(s1[e]), a2 = row16_sum(s2[e]);
      if (lane == 0) { red[(wave * 4 + e) * 2] = a1; red[(wave * 4 + e) * 2 + 1] = a2; }
    }
    lds_barrier();
    if (t < Cout * 2)
      atomicAdd(g.stats + (size_t)b * Cout * 2 + t, red[t] + red[8 + t] + red[16 + t] + red[24 + t]);
  }
}

// The same layer with the kernel COLUMN folded into the matrix rows: A rows are the 27 (kx, co) pairs (two 16-row fragments, 84 %
// real instead of 3 of 16), the reduction runs over (ky, channel) only, and the matrix cores produce per input column x'
//   P[(kx, co)][x'] = sum_{ky, c} w[co][ky][kx][c] * a[y + ky][x'][c];          y[x][co] = sum_kx P[(kx, co)][x + kx]
// -- 18 MFMAs per 16 input columns and output row instead of 81 per 16 output pixels (3.9x fewer per pixel); the shifted sum
// over kx (27 LDS reads per pixel) runs on the vector units from an LDS copy of P that reuses the halo.  Band: 8 rows x 56
// output columns = 64 input columns (4 column groups), C9K_W | W.
#ifndef C9K_ROWS
#define C9K_ROWS 4     // 4: 80 KB of LDS, two workgroups per CU (188 us at 224x224, B=48); 8: one per CU (206 us)
#endif
constexpr int C9K_R = C9K_ROWS, C9K_W = 56, C9K_RW = C9K_ROWS / 4;   // rows per wave
// Persistent (round 3 timestamps of the one-band-per-workgroup version, 10.7 us per band: 2.2 us weight staging + 3.3 us exposed halo
// round trip + 1.6 taps + 1.2 P / shifted sums + 2.4 statistics, 21 bands per CU slot): gridDim.x workgroups walk the bands
// blockIdx.x, blockIdx.x + gridDim.x, ...; the weights are staged ONCE, the next band's halo is in registers (and its image's
// coefficients in the other half of `cf`) before this band's taps start.
__global__ __launch_bounds__(256) void conv9_kxrows_kernel(const spb_gconv_args_t g, int nbands) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HW_ = C9K_W + 8, HR = C9K_R + 8, LDP = 40, LDWK = 32;   // weight rows unpadded: a fragment read is 1 KB contiguous
  float* cf = reinterpret_cast<float*>(smem);                      // [2 (band parity)][32][2] scale | shift
  float* red = cf + 128;                                           // [4 waves][4 ch][2]
  bf16_t* halo = reinterpret_cast<bf16_t*>(red + 32);              // [HR][HW_][LDP]; after the reduction loop: P, float [8][32][64]
  bf16_t* wl = halo + HR * HW_ * LDP;                              // [9 ky][32 rows = kx*3 + co][LDWK]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int H = g.Hin, W = g.Win, Cout = g.Cout;
  const int bx = W / C9K_W, bpi = bx * (H / C9K_R);                // bands per image
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);       // [Cout][9][9][32]
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  // XCD-local band ranges (round 5): workgroup k runs on XCD k % 8, and XCD x owns the CONTIGUOUS band range [base, base + cnt) -- whole
  // images, walked in order by its gridDim.x / 8 workgroups.  A band shares its 8 halo rows with its vertical neighbours; with bands dealt
  // round-robin over all workgroups those neighbours sat on different XCDs and every L2 fetched the rows again (PMC: 364 MB per launch for
  // 173 MB of algorithmic traffic).  Now the ~64 bands an XCD has in flight are 16 consecutive band rows of one image: ~1 MB, L2 resident.
  const int xcd = blockIdx.x & 7, nw8 = (int)gridDim.x >> 3;
  const int bq = nbands >> 3, br = nbands & 7;
  const int cnt = bq + (xcd < br ? 1 : 0), base = xcd * bq + (xcd < br ? xcd : br);
  int lb = blockIdx.x >> 3;
  if (lb >= cnt) return;
  int band = base + lb;
  constexpr int total = HR * HW_ * 4;
  constexpr int SUK = (total + 255) / 256;                         // halo vectors per thread (12 for 4-row bands)
  static_assert(SUK <= 16, "band halo must fit the register prefetch");
  Raw8<bf16_t> r[SUK];
  auto band_coef = [&](int bnd, int par) {      // threads 0..31: the band's image coefficients into cf[par]
    if (t < 32) {
      float sc, sh;
      gconv_coef(g, bnd / bpi, 32, t, sc, sh);
      cf[par * 64 + t] = sc; cf[par * 64 + 32 + t] = sh;
    }
  };
  auto halo_load = [&](int bnd) {
    const int b = bnd / bpi, bi = bnd - b * bpi;
    const int y0 = (bi / bx) * C9K_R, x0 = (bi % bx) * C9K_W;
#pragma unroll
    for (int u = 0; u < SUK; ++u) {
      const int i = t + 256 * u, ic = i < total ? i : total - 1;
      const int cv = ic & 3, hp = ic >> 2, hy = hp / HW_, hx = hp % HW_;
      const int sy = reflecti(y0 - 4 + hy, H), sx = reflecti(x0 - 4 + hx, W);
      r[u] = ldraw<bf16_t>(X + ((size_t)(b * H + sy) * W + sx) * 32 + cv * 8);
    }
  };
  GC_STAMP2(0);
  band_coef(band, 0);
  halo_load(band);
  for (int i = t; i < 9 * 32 * 4; i += 256) {                      // 16-byte granules of [ky][row][32 channels]
    const int v = i & 3, rr = (i >> 2) & 31, ky = i >> 7;
    const int kx = rr / 3, co = rr - kx * 3;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (rr < 27 && co < Cout) u = *reinterpret_cast<const uint4*>(Wg + ((size_t)(co * 9 + ky) * 9 + kx) * 32 + v * 8);
    *reinterpret_cast<uint4*>(wl + (ky * 32 + rr) * LDWK + v * 8) = u;
  }
  lds_barrier();
  GC_STAMP2(1);
  const bf16_t* hb = halo + ((wave * C9K_RW) * HW_ + li) * LDP + lq * 8;
  const bf16_t* wb = wl + li * LDWK + lq * 8;
  for (int k = 0;; ++k) {
    const int b = band / bpi, bi = band - b * bpi;
    const int y0 = (bi / bx) * C9K_R, x0 = (bi % bx) * C9K_W;
    // ---- commit this band's halo (loaded one band ago): a thread's channel chunk is fixed (256 % 4 == 0)
    {
      const float* cfk = cf + (k & 1) * 64;
      float scr[8], shr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { scr[j] = cfk[(t & 3) * 8 + j]; shr[j] = cfk[32 + (t & 3) * 8 + j]; }
#pragma unroll
      for (int u = 0; u < SUK; ++u) {
        const int i = t + 256 * u;
        if (i >= total) continue;
        float v[8];
        cvt8(r[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float uu = v[j] * scr[j] + shr[j];
          v[j] = g.relu ? fmaxf(uu, 0.f) : uu;
        }
        st8<bf16_t>(halo + (i >> 2) * LDP + (i & 3) * 8, v);
      }
    }
    lds_barrier();
    GC_STAMP2(2);
    const bool more = lb + nw8 < cnt;
    const int nband = base + lb + nw8;
    band_coef(more ? nband : band, (k + 1) & 1);      // next band: coefficients and halo in flight during the taps
    halo_load(more ? nband : band);                   // (clamped, not branched: the last band re-reads its own)
    // wave w: output rows w*C9K_RW ..; per row four 16-column groups of INPUT columns x two row fragments
    f32x4_t acc[C9K_RW][4][2];
#pragma unroll
    for (int rw = 0; rw < C9K_RW; ++rw)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[rw][q][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 9; ++ky) {
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(wb + (ky * 32) * LDWK);
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(wb + (ky * 32 + 16) * LDWK);
#pragma unroll
      for (int rw = 0; rw < C9K_RW; ++rw)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(hb + ((rw + ky) * HW_ + q * 16) * LDP);
          acc[rw][q][0] = SPB_MFMA16(a0, bf, acc[rw][q][0]);
          acc[rw][q][1] = SPB_MFMA16(a1, bf, acc[rw][q][1]);
        }
    }
    GC_STAMP2(3);
    lds_barrier();                                                 // everyone is done with the halo: it becomes P
    float* P = reinterpret_cast<float*>(halo);                       // [rows][32 (kx, co)][64 input columns]
#pragma unroll
    for (int rw = 0; rw < C9K_RW; ++rw)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int e = 0; e < 4; ++e) P[((wave * C9K_RW + rw) * 32 + f * 16 + lq * 4 + e) * 64 + q * 16 + li] = acc[rw][q][f][e];
    lds_barrier();
    GC_STAMP2(4);
    // lane = output column (56 of 64 lanes), every row of this wave: y[x][co] = sum_kx P[(kx, co)][x + kx]
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < C9K_W) {
#pragma unroll
      for (int rw = 0; rw < C9K_RW; ++rw) {
        const float* Pr = P + (size_t)(wave * C9K_RW + rw) * 32 * 64 + lane;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 9; ++kx)
#pragma unroll
          for (int co = 0; co < 3; ++co) v[co] += Pr[(kx * 3 + co) * 64 + kx];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (e < Cout ? v[e] : 0.f) + ((g.bias && e < Cout) ? g.bias[e] : 0.f);
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(Y + ((size_t)(b * H + y0 + wave * C9K_RW + rw) * W + x0 + lane) * g.ldc) = o;
        float rr[4];
        spb_unpack2(o.x, rr[0], rr[1]); spb_unpack2(o.y, rr[2], rr[3]);   // (16-bit storage format: common.h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += rr[e]; s2[e] += rr[e] * rr[e]; }
      }
    }
    GC_STAMP2(5);
    if (g.stats) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = xor32_sum(xor16_sum(row16_sum(s1[e]))), a2 = xor32_sum(xor16_sum(row16_sum(s2[e])));   // VALU only (no ds_bpermute)
        if (lane == 0) { red[(wave * 4 + e) * 2] = a1; red[(wave * 4 + e) * 2 + 1] = a2; }
      }
    }
    lds_barrier();     // P is read out (the next commit overwrites it); red is complete
    if (g.stats && t < Cout * 2)
      atomicAdd(g.stats + (size_t)b * Cout * 2 + t, red[t] + red[8 + t] + red[16 + t] + red[24 + t]);
    GC_STAMP2(6);
    if (!more) break;
    band = nband; lb += nw8;
  }
}

// ------------------------------------------------------------------------------------------- first layer: 3 -> 32, 9x9
// x fp32 NCHW [B,3,H,W]; w fp32 [32][3][9][9] (PyTorch layout); y bf16 NHWC [B,H,W,32] raw conv (+bias); stats [B][32][2].
// A workgroup owns a band of C9_R output rows of one image (blockIdx.y): the C9_R+8 input rows it needs are staged ONCE in LDS
// (bf16, planar per colour, reflection padding resolved at staging time), and the weights as MFMA A fragments per kernel row.
// Per 16-pixel group and kernel row ky one MFMA step covers k = ci*9 + kx (27 of 32 slots): the B fragment of lane (pixel, lq) is
// 8 two-byte LDS reads at  row(ci, y+ky) + x + kx  -- plain base + offset addressing.  (The first version gathered all 243 taps of
// a pixel from global memory with a table lookup and two reflections per tap: ~1200 instructions per group, 0.72 ms per batch.)
constexpr int C9_R = 8;
__global__ __launch_bounds__(256) void conv9_rgb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y, float* stats,
                                                        int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem9[];
  // Band in LDS as [row][column][4] bf16 (3 colours + a zero): 8 consecutive reduction elements k = kx*4 + ci of a pixel are the
  // 16 bytes of two neighbouring columns -- two aligned 8-byte reads instead of the 8 two-byte reads + packing of the planar
  // layout (72 LDS instructions per 16 pixels, the kernel's bound).  11 matrix steps per 16-pixel group: one per kernel row for
  // kx = 0..7, and kx = 8 of all nine rows folded into two more (k = ky*4 + ci).
  const int LDX = W + 8 + 1;                                   // staged columns: 4 reflected each side (+1: bank skew), x 4 channels
  uint4* wfrag = reinterpret_cast<uint4*>(smem9);              // [11 steps][2 cb][64 lanes]
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem9 + 11 * 2 * 64 * 16);   // [C9_R + 8][LDX][4]
  float* red = reinterpret_cast<float*>(xs + (C9_R + 8) * LDX * 4 + 8);  // [4][64]
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, y0 = blockIdx.x * C9_R;
  const int rows = min(C9_R, H - y0);
  // A operand: W[co][8 reduction elements of quarter lq], row li of fragment cb = channel (li / 4) * 8 + cb * 4 + li % 4: a lane of
  // the transposed product then holds 8 consecutive channels over its two fragments -- one 16-byte store per pixel and lane,
  // 1 KB contiguous per wave instruction (it was two 8-byte stores 32 B apart)
  for (int f = wave; f < 22; f += 4) {
    const int st_ = f >> 1, cb = f & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = lq * 8 + e, ci = k & 3, q = k >> 2;        // q: kx (steps 0..8) or ky (steps 9, 10)
      const int ky = st_ < 9 ? st_ : (st_ == 9 ? q : 8), kx = st_ < 9 ? q : 8;
      const bool ok = ci < 3 && (st_ < 10 || q == 0);
      v[e] = ok ? w[(((li >> 2) * 8 + cb * 4 + (li & 3)) * 3 + ci) * 81 + ky * 9 + kx] : 0.f;
    }
    wfrag[(st_ * 2 + cb) * 64 + lane] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
  // input band: rows y0-4 .. y0+rows+3, columns -4 .. W+3, reflected; one thread = one pixel (3 loads, one 8-byte store)
  const int nrow = rows + 8, ncol = W + 8;
  const int npix = nrow * ncol;
  const float inv_ncol = 1.0f / (float)ncol;
  for (int i0 = threadIdx.x; i0 < npix; i0 += 256 * 6) {       // 18 loads in flight per thread
    float v[6][3];
    int dst[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = i0 + 256 * u, ic = i < npix ? i : npix - 1;
      const int r = (int)(((float)ic + 0.5f) * inv_ncol), cx = ic - r * ncol;     // exact for ic < 2^20
      const size_t o = (size_t)reflecti(y0 - 4 + r, H) * W + reflecti(cx - 4, W);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) v[u][ci] = x[(size_t)(b * 3 + ci) * H * W + o];
      dst[u] = i < npix ? (r * LDX + cx) * 4 : -1;
    }
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (dst[u] >= 0) *reinterpret_cast<uint2*>(xs + dst[u]) = make_uint2(pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], 0.f));
  }
  lds_barrier();
  float s1[2][4], s2[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[cb][e] = 0.f; s2[cb][e] = 0.f; }
  const int gpr = W >> 4;
  int ry = 0, gx = wave;                       // group gi = ry * gpr + gx, walked without dividing
  while (gx >= gpr) { gx -= gpr; ++ry; }
  for (int gi = wave; gi < rows * gpr; gi += 4) {
    const int ox = gx * 16 + li, oy = y0 + ry;
    f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    const bf16_t* rp = xs + ((size_t)ry * LDX + ox) * 4;
#pragma unroll
    for (int st_ = 0; st_ < 11; ++st_) {
      uint2 lo, hi;
      if (st_ < 9) {            // kernel row st_, columns kx = 2 lq, 2 lq + 1
        lo = *reinterpret_cast<const uint2*>(rp + ((size_t)st_ * LDX + lq * 2) * 4);
        hi = *reinterpret_cast<const uint2*>(rp + ((size_t)st_ * LDX + lq * 2 + 1) * 4);
      } else if (st_ == 9) {    // column kx = 8 of kernel rows 2 lq, 2 lq + 1
        lo = *reinterpret_cast<const uint2*>(rp + ((size_t)(lq * 2) * LDX + 8) * 4);
        hi = *reinterpret_cast<const uint2*>(rp + ((size_t)(lq * 2 + 1) * LDX + 8) * 4);
      } else {                  // column 8 of kernel row 8 (quarter 0; the other quarters meet zero weights)
        lo = *reinterpret_cast<const uint2*>(rp + ((size_t)8 * LDX + 8) * 4);
        hi = lo;
      }
      const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[cb] = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wfrag[(st_ * 2 + cb) * 64 + lane]), bf, acc[cb]);
    }
    {
      uint2 o[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[cb][e] + (bias ? bias[lq * 8 + cb * 4 + e] : 0.f);
        o[cb].x = pack_bf16x2(v[0], v[1]); o[cb].y = pack_bf16x2(v[2], v[3]);
        float r[4];
        spb_unpack2(o[cb].x, r[0], r[1]); spb_unpack2(o[cb].y, r[2], r[3]);   // (16-bit storage format: common.h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[cb][e] += r[e]; s2[cb][e] += r[e] * r[e]; }
      }
      *reinterpret_cast<uint4*>(y + ((size_t)(b * H + oy) * W + ox) * 32 + lq * 8) = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
    }
    gx += 4;
    while (gx >= gpr) { gx -= gpr; ++ry; }
  }
  if (stats) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = row16_sum(s1[cb][e]), a2 = row16_sum(s2[cb][e]);
        if (li == 0) { red[wave * 64 + (lq * 8 + cb * 4 + e) * 2] = a1; red[wave * 64 + (lq * 8 + cb * 4 + e) * 2 + 1] = a2; }
      }
    lds_barrier();
    if (threadIdx.x < 64)
      atomicAdd(stats + (size_t)b * 64 + threadIdx.x,
                red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------------------- small pieces
// coef[b][c] = (gamma*invstd, beta - mean*gamma*invstd) from the raw sums of an instance-normalised tensor
__global__ void in_coef_kernel(const float* stats, const float* gamma, const float* beta, int ld, float* coef, int B, int C,
                               float inv_n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  const float mean = stats[i * 2] * inv_n;
  const float var = fmaxf(stats[i * 2 + 1] * inv_n - mean * mean, 0.f);
  const float is = rsqrtf(var + eps);
  const float ga = gamma ? gamma[(size_t)b * ld + c] : 1.f, be = beta ? beta[(size_t)b * ld + c] : 0.f;
  coef[i * 2] = ga * is;
  coef[i * 2 + 1] = be - mean * ga * is;
}

// out[b][j] = bias[j] + sum_i style[b][i] * W[j][i]   (all 26 nn.Linear(100, C) of the decoder as one [N,100] matrix)
__global__ void style_fc_kernel(const float* style, const float* W, const float* bias, float* out, int B, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N, j = i % N;
  float s = bias[j];
  for (int k = 0; k < 100; ++k) s += style[b * 100 + k] * W[(size_t)j * 100 + k];
  out[i] = s;
}

// y = [res +] act(x*scale + shift), NHWC bf16 (materialises the residual stream)
__global__ void in_apply_kernel(const bf16_t* X, const float* coef, const bf16_t* res, bf16_t* Y, long long hw, int C, int relu,
                                long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int CV = C >> 3;
  const int cv = (int)(i % CV);
  const long long b = i / (CV * hw);
  float v[8], r[8];
  ld8<bf16_t>(X + i * 8, v);
  if (res) ld8<bf16_t>(res + i * 8, r);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float* cf = coef + ((size_t)b * C + cv * 8 + j) * 2;
    float u = v[j] * cf[0] + cf[1];
    u = relu ? fmaxf(u, 0.f) : u;
    v[j] = res ? u + r[j] : u;
  }
  st8<bf16_t>(Y + i * 8, v);
}

// The same with the coefficients from the producer's sums: a thread keeps ONE channel chunk (8 scale / shift pairs, computed once)
// and walks IAS_PX pixels of one image with it.  grid (ceil(hw / (rows-per-block)), B), block 256 = 256 / CV pixel lanes x CV chunks.
constexpr int IAS_PX = 8;
__global__ void in_apply_stats_kernel(const bf16_t* X, const float* stats, const float* gamma, const float* beta, int ld, float inv_n,
                                      float eps, const bf16_t* res, bf16_t* Y, long long hw, int C, int relu) {
  const int CV = C >> 3, cvs = __ffs(CV) - 1;
  const int t = threadIdx.x, cv = t & (CV - 1), pl = t >> cvs, npl = 256 >> cvs;     // CV is a power of two <= 256
  const int b = blockIdx.y;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) in_coef_of(stats, gamma, beta, ld, inv_n, eps, b, C, cv * 8 + j, sc[j], sh[j]);
  const long long p0 = (long long)blockIdx.x * npl * IAS_PX + pl;
  const bf16_t* Xb = X + (size_t)b * hw * C;
  const bf16_t* Rb = res ? res + (size_t)b * hw * C : nullptr;
  bf16_t* Yb = Y + (size_t)b * hw * C;
#pragma unroll
  for (int u = 0; u < IAS_PX; ++u) {
    const long long p = p0 + (long long)u * npl;
    if (p >= hw) continue;
    float v[8], r[8];
    ld8<bf16_t>(Xb + p * C + cv * 8, v);
    if (Rb) ld8<bf16_t>(Rb + p * C + cv * 8, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float uu = v[j] * sc[j] + sh[j];
      uu = relu ? fmaxf(uu, 0.f) : uu;
      v[j] = Rb ? uu + r[j] : uu;
    }
    st8<bf16_t>(Yb + p * C + cv * 8, v);
  }
}
__global__ void final_sigmoid_stats_kernel(const bf16_t* Z, const float* stats, const float* gamma, const float* beta, int ld, float inv_n,
                                           float eps, float* out, long long hw, int ldc) {
  const int b = blockIdx.y;
  float sc[3], sh[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) in_coef_of(stats, gamma, beta, ld, inv_n, eps, b, 3, c, sc[c], sh[c]);
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float u = bf2f(Z[((size_t)b * hw + p) * ldc + c]) * sc[c] + sh[c];
    out[((size_t)b * 3 + c) * hw + p] = 1.f / (1.f + __expf(-u));
  }
}

// out[b][c][y][x] = sigmoid(z[b][y][x][c] * scale + shift), z NHWC bf16 with channel stride ldc, out fp32 NCHW (3 channels)
__global__ void final_sigmoid_kernel(const bf16_t* Z, const float* coef, float* out, int B, long long hw, int ldc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * hw) return;
  const long long b = i / hw, p = i % hw;
  for (int c = 0; c < 3; ++c) {
    const float z = bf2f(Z[i * ldc + c]);
    const float u = z * coef[((size_t)b * 3 + c) * 2] + coef[((size_t)b * 3 + c) * 2 + 1];
    out[((size_t)b * 3 + c) * hw + p] = 1.f / (1.f + __expf(-u));
  }
}

}  // namespace

static int g_gconv_slab = 2;   // 0: per-wave weight streaming; 1: slab kernel, 4 tiles / 1 workgroup per CU; 2: 2 tiles / 2 per CU
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_slab(int on) { g_gconv_slab = on; return 0; }
#endif

// 8x8 tiles side by side per workgroup in the LDS-resident-weight layers (32->64 stride 2, 64->32 after upsampling): with one,
// a wave reads 1 pixel + NB weight fragments per NB MFMAs (294..353 B/clk/CU of LDS reads at matrix-core speed, over the 256 peak)
static int g_slab_pf = 6;    // weight slabs in flight per workgroup of the wide layers (3 | 6)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_slab_pf(int n) { g_slab_pf = n; return 0; }
#endif
static int g_halo_prefetch = 1;   // LDS-resident-weight layers: next tile's halo loads in flight during the current tile
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_halo_prefetch(int on) { g_halo_prefetch = on; return 0; }
#endif
static int g_wlds_pxg = 1;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_wlds_pxg(int n) { g_wlds_pxg = n; return 0; }
#endif

static int g_conv9_wgs = 512;  // 9x9 32->3: persistent workgroups
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_conv9_wgs(int n) { g_conv9_wgs = n < 1 ? 1 : n; return 0; }
#endif
static int g_conv9_band = 2;   // 9x9 32->3: 0 generic tile kernel, 1 band kernel, 2 kernel columns folded into the matrix rows
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_conv9_band(int on) { g_conv9_band = on; return 0; }
#endif

int spb_gconv_f32(const spb_gconv_args_t* a, hipStream_t stream);   // ghiasi_f32.hip
extern "C" int spb_gconv(int dtype, const spb_gconv_args_t* a, spb_stream_t stream) {
  if (!a || !a->X || !a->W || !a->Y) return SPB_E_ARG;
  if (dtype == SPB_F32) {      // reference-precision mode (csrc/ghiasi_f32.hip): any Cin / Cout, direct convolution on the vector units
    if (a->B <= 0 || a->Cin <= 0 || a->Cout <= 0 || (a->KH & 1) == 0 || a->ldc < a->Cout) return SPB_E_SHAPE;
    if ((a->stride != 1 && a->stride != 2) || (a->upsample != 1 && a->upsample != 2)) return SPB_E_SHAPE;
    if ((a->Hin * a->upsample) % a->stride || (a->Win * a->upsample) % a->stride) return SPB_E_SHAPE;
    if (a->KH / 2 >= a->Hin * a->upsample || a->KH / 2 >= a->Win * a->upsample) return SPB_E_SHAPE;
    const int e = spb_gconv_f32(a, (hipStream_t)stream);
    if (e) return e;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  if (a->B <= 0 || (a->Cin & 31) || a->Cout <= 0 || a->Cout > 128 || (a->KH != 3 && a->KH != 9)) return SPB_E_SHAPE;
  if ((a->stride != 1 && a->stride != 2) || (a->upsample != 1 && a->upsample != 2)) return SPB_E_SHAPE;
  const int Hu = a->Hin * a->upsample, Wu = a->Win * a->upsample;
  if (Hu % a->stride || Wu % a->stride) return SPB_E_SHAPE;
  const int Hout = Hu / a->stride, Wout = Wu / a->stride;
  if ((Hout & 7) || (Wout & 7) || a->ldc < a->Cout || (a->ldc & 3)) return SPB_E_SHAPE;
  if (a->KH / 2 >= Hu || a->KH / 2 >= Wu) return SPB_E_SHAPE;   // reflection padding needs pad < size
  if (g_conv9_band == 2 && a->KH == 9 && a->Cin == 32 && a->Cout <= 3 && a->stride == 1 && a->upsample == 1 && !(Wout % C9K_W) &&
      !(Hout % C9K_R) && a->ldc >= 4) {
    const size_t halo_b = (size_t)(C9K_R + 8) * (C9K_W + 8) * 40 * sizeof(bf16_t), p_b = (size_t)C9K_R * 32 * 64 * sizeof(float);
    const size_t ldsk = (128 + 32) * sizeof(float) + (halo_b > p_b ? halo_b : p_b) + (size_t)9 * 32 * 32 * sizeof(bf16_t);
    static bool oncek = false;
    if (!oncek) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv9_kxrows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); oncek = true; }
    const int nbands = (Hout / C9K_R) * (Wout / C9K_W) * a->B;
    const int wgs9 = nbands < g_conv9_wgs ? ((nbands + 7) & ~7) : (g_conv9_wgs < 8 ? 8 : (g_conv9_wgs & ~7));      // persistent: two per CU; a multiple of 8 (one share per XCD)
    hipLaunchKernelGGL(conv9_kxrows_kernel, dim3((unsigned)wgs9), dim3(256), ldsk, (hipStream_t)stream, *a, nbands);
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (g_conv9_band && a->KH == 9 && a->Cin == 32 && a->Cout <= 4 && a->stride == 1 && a->upsample == 1 && !(Wout % C9O_W) &&
      !(Hout % C9O_R) && a->ldc >= 4) {
    const size_t ldsb = (64 + 32) * sizeof(float) + ((size_t)(C9O_R + 8) * (C9O_W + 8) * 40 + (size_t)4 * (81 * 32 + 8)) * sizeof(bf16_t);
    static bool onceb = false;
    if (!onceb) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv9_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); onceb = true; }
    hipLaunchKernelGGL(conv9_band_kernel, dim3((unsigned)((Hout / C9O_R) * (Wout / C9O_W)), (unsigned)a->B), dim3(256), ldsb, (hipStream_t)stream, *a);
    SPB_CHECK_LAUNCH();
    return 0;
  }
  const int HT = 7 * a->stride + a->KH, KK = a->KH * a->KH;
  const int NB = a->Cout <= 16 ? 1 : (a->Cout <= 32 ? 2 : (a->Cout <= 64 ? 4 : 8));
  // fragment-order slots span all NB*16 rows (NB == 1: slot == row, Cout rows are enough)
  const size_t wbytes = (size_t)(NB == 1 ? a->Cout : NB * 16) * (KK * a->Cin + 8) * 2;
  const bool wlds = wbytes <= 64 * 1024;
  if (!wlds && a->Cout != NB * 16) return SPB_E_SHAPE;   // the streaming variant has no zero rows
  if (!wlds) {   // wide layers: shared weight slab, PXG tiles per workgroup
    const int PX = g_gconv_slab == 1 ? 4 : 2;
    const int tpi4 = (Hout >> 3) * (Wout >> 3);
    const int gpi4 = (tpi4 + PX - 1) / PX;
    const size_t lds4 = (size_t)a->Cin * 2 * sizeof(float) + (size_t)4 * NB * 16 * 2 * sizeof(float) +
                        (size_t)2 * NB * 16 * 40 * 2 + (size_t)PX * HT * HT * (a->Cin + 8) * 2;
    const int PFd = ((KK * (a->Cin >> 5)) % 6 == 0 && g_slab_pf == 6) ? 6 : 3;
    if (lds4 <= 160 * 1024 && (NB == 4 || NB == 8) && g_gconv_slab && (KK * (a->Cin >> 5)) % 3 == 0) {
      const dim3 grid4((unsigned)(a->B * gpi4));
      hipStream_t s4 = (hipStream_t)stream;
#define S_(NB_, PX_)                                                                                                   \
      { if (PFd == 6) S2_(NB_, PX_, 6) else S2_(NB_, PX_, 3) }
#define S2_(NB_, PX_, PF_)                                                                                             \
      {                                                                                                                \
        static bool once = false;                                                                                      \
        if (!once) {                                                                                                   \
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_slab_kernel<NB_, PX_, PF_>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
          once = true;                                                                                                 \
        }                                                                                                              \
        hipLaunchKernelGGL((gconv_slab_kernel<NB_, PX_, PF_>), grid4, dim3(256), lds4, s4, *a);                             \
      }
      if (NB == 4) { if (PX == 4) S_(4, 4) else S_(4, 2) }
      else { if (PX == 4) S_(8, 4) else S_(8, 2) }
#undef S_
#undef S2_
      SPB_CHECK_LAUNCH();
      return 0;
    }
  }
  const int pxg = wlds ? ((g_wlds_pxg == 2 && (NB == 2 || NB == 4)) ? 2 : 1) : 2;
  const size_t lds = (size_t)a->Cin * 2 * sizeof(float) + (size_t)4 * NB * 16 * 2 * sizeof(float) +
                     (size_t)pxg * HT * HT * (a->Cin + 8) * 2 + (wlds ? wbytes : 0);
  // tile groups per workgroup: the largest divisor of the groups of one image that still leaves >= 1024 workgroups
  const int tpi = (Hout >> 3) * (Wout >> 3);
  const int gpi = (tpi + pxg - 1) / pxg;
  int tpw = 1;
  for (int d = 1; d <= gpi; ++d)
    if (gpi % d == 0 && (long long)a->B * (gpi / d) >= 1024) tpw = d;
  const dim3 grid((unsigned)(a->B * (gpi / tpw)));
  hipStream_t s = (hipStream_t)stream;
  const bool pre = g_halo_prefetch && tpw > 1 && (long long)pxg * HT * HT * (a->Cin >> 3) <= 1280;
#define G2_(NB_, WL_, PX_, PRE_)                                                                                     \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_kernel<NB_, WL_, PX_, PRE_>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((gconv_kernel<NB_, WL_, PX_, PRE_>), grid, dim3(256), lds, s, *a, tpw);                       \
  }
#define G_(NB_, WL_, PX_) { if (pre && WL_) G2_(NB_, WL_, PX_, true) else G2_(NB_, WL_, PX_, false) }
  if (NB == 1) { if (wlds) G_(1, true, 1) else return SPB_E_SHAPE; }
  else if (NB == 2) { if (wlds) { if (pxg == 2) G_(2, true, 2) else G_(2, true, 1) } else G_(2, false, 2) }
  else if (NB == 4) { if (wlds) { if (pxg == 2) G_(4, true, 2) else G_(4, true, 1) } else G_(4, false, 2) }
  else G_(8, false, 2)
#undef G_
#undef G2_
  SPB_CHECK_LAUNCH();
  return 0;
}

static int g_up2_wreg = 2;       // phase layers: weights in registers (1: the 64 -> 32 layer; 2: the 128 -> 64 layer as well)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_up2_wreg(int on) { g_up2_wreg = on; return 0; }
#endif
static int g_up2_prefetch = 1;   // phase kernels: next tile group's halo loads in flight during the current group
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_up2_prefetch(int on) { g_up2_prefetch = on; return 0; }
#endif

// a->W: phase weights [4][Cout][4][Cin] (Ghiasi._pack builds them); a->upsample must be 2, a->stride 1, a->KH 3
extern "C" int spb_gconv_up2(int dtype, const spb_gconv_args_t* a, spb_stream_t stream) {
  if (!a || !a->X || !a->W || !a->Y) return SPB_E_ARG;
  if (dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  if (a->B <= 0 || (a->Cin & 31) || a->Cout <= 0 || a->Cout > 128 || a->KH != 3 || a->stride != 1 || a->upsample != 2) return SPB_E_SHAPE;
  const int Hout = 2 * a->Hin, Wout = 2 * a->Win;
  if ((Hout & 7) || (Wout & 7) || a->ldc < a->Cout || (a->ldc & 3)) return SPB_E_SHAPE;
  const int NB = a->Cout <= 16 ? 1 : (a->Cout <= 32 ? 2 : (a->Cout <= 64 ? 4 : 8));
  const size_t wbytes = (size_t)4 * NB * 16 * (4 * a->Cin + 8) * 2;
  const bool wreg4 = g_up2_wreg >= 2 && NB == 4 && a->Cin == 128 && a->Cout == 64;                  // ... 256 registers of them, one workgroup per CU
  const bool wreg = (g_up2_wreg && NB == 2 && a->Cin == 64) || wreg4;   // phase weights in registers (gconv_up2_kernel, WREG)
  const bool wlds = !wreg && wbytes <= 72 * 1024;
  if (!wlds && a->Cout != NB * 16) return SPB_E_SHAPE;
  if (NB != 2 && NB != 4) return SPB_E_UNSUPPORTED;
  const int pxg = (wlds || wreg) ? 2 : 4;
  const size_t lds = (size_t)a->Cin * 2 * sizeof(float) + (size_t)4 * NB * 16 * 2 * sizeof(float) +
                     (size_t)pxg * 36 * (a->Cin + 8) * 2 + (wlds ? wbytes : 0);
  const int tpi = (Hout >> 3) * (Wout >> 3);
  const int gpi = (tpi + pxg - 1) / pxg;
  int tpw = 1;
  for (int d = 1; d <= gpi; ++d)
    if (gpi % d == 0 && (long long)a->B * (gpi / d) >= (wreg4 ? 512 : 1024)) tpw = d;     // wreg4: one workgroup per CU, a few rounds
  const dim3 grid((unsigned)(a->B * (gpi / tpw)));
  hipStream_t s = (hipStream_t)stream;
#define U_(NB_, WL_, PX_, NV_) U5_(NB_, WL_, PX_, NV_, false)
#define U5_(NB_, WL_, PX_, NV_, WR_)                                                                                 \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_up2_kernel<NB_, WL_, PX_, NV_, WR_>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((gconv_up2_kernel<NB_, WL_, PX_, NV_, WR_>), grid, dim3(256), lds, s, *a, tpw);               \
  }
  const int hv = pxg * 36 * (a->Cin >> 3);          // halo vectors per tile group
  const bool pre = g_up2_prefetch && tpw > 1;
  if (wreg4) {
#define U6_(NV_)                                                                                                     \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_up2_kernel<4, false, 2, NV_, true, 16>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((gconv_up2_kernel<4, false, 2, NV_, true, 16>), grid, dim3(256), lds, s, *a, tpw);            \
  }
    if (pre && hv <= 1280) U6_(5) else U6_(0)
#undef U6_
  } else if (wreg) {
    if (pre) U5_(2, false, 2, 3, true) else U5_(2, false, 2, 0, true)
  } else if (NB == 2) {
    if (wlds) { if (pre && hv <= 768) U_(2, true, 2, 3) else U_(2, true, 2, 0) }
    else { if (pre && hv <= 2304) U_(2, false, 4, 9) else U_(2, false, 4, 0) }
  } else {
    if (wlds) { if (pre && hv <= 768) U_(4, true, 2, 3) else U_(4, true, 2, 0) }
    else { if (pre && hv <= 2304) U_(4, false, 4, 9) else U_(4, false, 4, 0) }
  }
#undef U_
#undef U5_
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_conv9_rgb(const float* x, const float* w, const float* bias, void* y, float* stats, int B, int H, int W,
                             spb_stream_t stream) {
  if (!x || !w || !y || B <= 0 || H < 5 || W < 16 || (W & 15)) return SPB_E_ARG;
  const size_t lds = (size_t)11 * 2 * 64 * 16 + ((size_t)4 * (C9_R + 8) * (W + 9) + 8) * sizeof(bf16_t) + 4 * 64 * sizeof(float) + 16;
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(&conv9_rgb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(conv9_rgb_kernel, dim3((unsigned)((H + C9_R - 1) / C9_R), (unsigned)B), dim3(256), lds, (hipStream_t)stream, x, w, bias,
                     (bf16_t*)y, stats, B, H, W);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_in_coef(const float* stats, const float* gamma, const float* beta, int ld, float* coef, int B, int C,
                           long long hw, float eps, spb_stream_t stream) {
  if (!stats || !coef || B <= 0 || C <= 0 || hw <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(in_coef_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta, ld, coef, B,
                     C, 1.f / (float)hw, eps);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_style_fc(const float* style, const float* W, const float* bias, float* out, int B, int N, spb_stream_t stream) {
  if (!style || !W || !bias || !out || B <= 0 || N <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(style_fc_kernel, dim3((B * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, style, W, bias, out, B, N);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_in_apply(const void* X, const float* coef, const void* res, void* Y, int B, long long hw, int C, int relu,
                            spb_stream_t stream) {
  if (!X || !coef || !Y || B <= 0 || hw <= 0 || (C & 7)) return SPB_E_ARG;
  const long long n8 = (long long)B * hw * (C >> 3);
  hipLaunchKernelGGL(in_apply_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, coef,
                     (const bf16_t*)res, (bf16_t*)Y, hw, C, relu, n8);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_in_apply_stats(const void* X, const float* stats, const float* gamma, const float* beta, int ld, float eps,
                                  const void* res, void* Y, int B, long long hw, int C, int relu, spb_stream_t stream) {
  if (!X || !stats || !Y || B <= 0 || hw <= 0 || (C & 7) || C > 2048) return SPB_E_ARG;
  const int CV = C >> 3;
  if (CV & (CV - 1)) return SPB_E_SHAPE;                      // channel chunks per pixel: a power of two
  const long long rows = (long long)(256 / CV) * IAS_PX;
  hipLaunchKernelGGL(in_apply_stats_kernel, dim3((unsigned)((hw + rows - 1) / rows), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)X, stats, gamma, beta, ld, 1.f / (float)hw, eps, (const bf16_t*)res, (bf16_t*)Y, hw, C, relu);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_final_sigmoid_stats(const void* Z, const float* stats, const float* gamma, const float* beta, int ld, float eps,
                                       float* out, int B, long long hw, int ldc, spb_stream_t stream) {
  if (!Z || !stats || !out || B <= 0 || hw <= 0 || ldc < 3) return SPB_E_ARG;
  hipLaunchKernelGGL(final_sigmoid_stats_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)Z, stats, gamma, beta, ld, 1.f / (float)hw, eps, out, hw, ldc);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_final_sigmoid(const void* Z, const float* coef, float* out, int B, long long hw, int ldc, spb_stream_t stream) {
  if (!Z || !coef || !out || B <= 0 || hw <= 0 || ldc < 3) return SPB_E_ARG;
  const long long n = (long long)B * hw;
  hipLaunchKernelGGL(final_sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Z, coef,
                     out, B, hw, ldc);
  SPB_CHECK_LAUNCH();
  return 0;
}
