// Depthwise 3x3 convolution, "tile" kernels (round 4) for the large maps (bf16, width >= 28): forward.
// Replaces nn.Conv2d(groups=C, k=3, pad=1, stride 1|2) + the BatchNorm2d + ReLU6 in FRONT of it (applied to the operand while it is
// staged) of torchvision's InvertedResidual blocks 1..7 (reference call site park2019.py:107-108); C-ABI: spb_dwconv_fwd
// (include/spb_hip.h) dispatches here for bf16 maps at least g_dw_tile_min wide (dwconv_rows.hip).
//
// Why.  The row-unit kernels (dwconv_rows.hip) are bound by instruction issue, not by memory: SQ counters of round 4 show their waves
// ACTIVE 40-50 % of the time at two waves per SIMD (the SIMDs issue nearly every cycle) for ~1.5 wave-instructions per output
// element -- 9 FMAs, 6 DPP moves for the left/right taps, the BatchNorm + ReLU6 of every operand element, 18 LDS weight reads per
// row step and ~280 scalar instructions of cursor / clamp / 64-bit address arithmetic per step.  Here an element costs ~0.3:
//   * a workgroup owns a 16 x 8 output tile of 32 channels.  Its input tile (with halo) is staged ONCE in LDS as channel-PLANAR
//     bf16 rows, already normalised + activated, zero outside the image;
//   * staging = one 16-byte global load per lane (16 pixels x 32 channels per wave instruction), then the MATRIX CORE as a
//     transposer: with the raw NHWC vector as the A operand and a 0/1 selector as B, v_mfma_f32_16x16x32_bf16 returns the
//     same values as exact f32 with FOUR CONSECUTIVE PIXELS OF ONE CHANNEL per lane (the C layout) -- conversion and transposition
//     without a single unpack or LDS scatter; scale/shift/ReLU6 run once per element on that, one 8-byte LDS store per 4 elements;
//   * the taps are v_dot2c_f32_bf16 on pixel PAIRS: out[x] = (a[x-1],a[x]).(w0,w1) + (a[x+1],.).(w2,0) -- 6 instructions per
//     output instead of 15, operands straight from ds_read_b128 (no unpack), the six packed weight pairs of a lane's channel in
//     registers for the whole kernel.  The depthwise weights are therefore rounded to bf16, like every other bf16-mode operand
//     (and like torch.autocast, which casts the depthwise weight too: the reference's --use_fp16 recipe, trainer.py:73-78);
//   * results leave through the matrix core again (selector transposer back to NHWC: 8 consecutive channels of one pixel per
//     lane, 16-byte stores); the per-channel batch sums stay in two registers per lane for the whole kernel.
#include "common.h"

namespace {

constexpr int TW = 8, TH = 16, CHK = 32;

template <int ST> struct TG {
  static constexpr int ROWP = ST == 1 ? 12 : 20;               // LDS pixel slots per tile row (slot 0 = input column x0 - 2): 10 | 18 used, multiple of 4
  static constexpr int ROWS = ST == 1 ? TH + 2 : 2 * TH + 1;   // input rows of a tile
  static constexpr int GROUPS = (ROWS * ROWP + 15) / 16;       // 16-slot staging groups: 14 | 42  (16 / 24-slot rows: 18 | 50 -- staging is half of the instructions)
  static constexpr int CHS = ST == 1 ? 464 : 1360;             // bytes per channel plane: >= GROUPS * 32, an ODD multiple of 16
  static constexpr int GPW = (GROUPS + 3) / 4;                 // groups per wave
};

#ifdef SPB_F16   // the IEEE-half twin (common.h): v_dot2_f32_f16
__device__ __forceinline__ float dot2(uint32_t a, uint32_t w, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(spb_h16x2, a), __builtin_bit_cast(spb_h16x2, w), acc, false);
}
#define DWT_ONE_LO 0x00003C00u
#define DWT_ONE_HI 0x3C000000u
#else
#define DWT_ONE_LO 0x00003F80u
#define DWT_ONE_HI 0x3F800000u
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(uint32_t a, uint32_t w, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a), __builtin_bit_cast(bf16x2_hw, w), acc, false);
}
#endif

// 0/1 selector (B operand): column j of the result = reduction index j + 16 h.  Lane (j = lane & 15, q = lane >> 4) holds k = 8q .. 8q+7.
__device__ __forceinline__ bf16x8_t selector(int lane, int h) {
  const int j = lane & 15, q = lane >> 4, t = j + 16 * h - 8 * q;     // position of the single 1.0 among this lane's 8 elements (if 0 <= t < 8)
  uint4 u;
  u.x = t == 0 ? DWT_ONE_LO : (t == 1 ? DWT_ONE_HI : 0u);
  u.y = t == 2 ? DWT_ONE_LO : (t == 3 ? DWT_ONE_HI : 0u);
  u.z = t == 4 ? DWT_ONE_LO : (t == 5 ? DWT_ONE_HI : 0u);
  u.w = t == 6 ? DWT_ONE_LO : (t == 7 ? DWT_ONE_HI : 0u);
  return __builtin_bit_cast(bf16x8_t, u);
}

struct TileGeo { int nty, ntx, ntiles, nchunks, nsp; };

// CLAMP: the operand's activation is ReLU / ReLU6 (one v_med3 per element); otherwise the general branch-free form of common.h.
// Everything inside the tile loop is straight-line code: masks are integer ANDs, every wave stages GPW groups (the last ones
// may repeat a group -- identical values to the same LDS address), loads sit on clamped addresses.  (The first version let the
// compiler turn `ok ? f(x) : 0` into one basic block per element and sank the last group's load into its conditional block.)
//
// XP (round 6, spb_dw_args_t::Xe): the input tensor is the output of a 1x1 expand convolution that is NOT in memory.  The transposer of
// phase 1 already IS a matrix product -- raw NHWC vector x selector -- so the expand convolution takes its place: the A operand becomes
// the (BatchNorm'd, re-rounded) 8 input channels of the pixel, the selector becomes this lane's 16 expand-weight rows, and the matrix core
// returns the expanded activation z = W x as exact f32 in the same "four consecutive pixels of one channel" layout.  Same instruction
// count as the transposer; the staged bytes per pixel drop from 2 C to 2 Ce (C = 6 Ce), shared by the C / 32 workgroups of a tile
// (one XCD: L2 hits), and the 96 / 144-channel tensor of the 112x112 / 56x56 maps is never written or read.
// SPB_DWT_XP_OCC3: the stride-2 expand-recompute instance without the cross-tile register prefetch (its rows are a few KB of L2-resident
// data; the prefetch array is 44 registers) at three workgroups per CU instead of two (168 registers, 8 bytes of scratch per lane).
// Measured in the step, two A/B pairs (scratch/build_variant.py occ3 -DSPB_DWT_XP_OCC3=1 against =0): 2.436 -> 2.425 ms.
#ifndef SPB_DWT_XP_OCC3
#define SPB_DWT_XP_OCC3 1
#endif
template <int ST, bool CLAMP, bool XP = false>
__global__ __launch_bounds__(256, ST == 1 ? 4 : ((XP && SPB_DWT_XP_OCC3) ? 3 : 2)) void dwt_fwd_kernel(const spb_dw_args_t a, const TileGeo tg) {
  typedef TG<ST> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = a.C, H = a.H, W = a.W;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  // all channel chunks of a spatial tile on ONE XCD (block b runs on XCD b % 8): the 64-byte pieces of a pixel's lines meet in one L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int chunk = idx % tg.nchunks, sp = (idx / tg.nchunks) * 8 + xcd;
  const int c0 = chunk * CHK;
  const bf16_t* X = reinterpret_cast<const bf16_t*>(XP ? a.Xe : a.X);
  bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);
  const float ahi = act_hi(a.pro.act), ans = act_ns(a.pro.act, a.pro.slope);
  const int CX = XP ? a.Ce : C;                              // channels of the staged tensor

  // ---- per-lane constants
  // B operand of the staging product.  Plain: the 0/1 selector.  XP: rows c0 + 16 h + r of the expand weights, input channels 8 q .. 8 q + 7
  // (zero past Ce and past C: whatever the clamped A lanes hold there is multiplied away)
  const bf16x8_t sel0 = selector(lane, 0), sel1 = selector(lane, 1);   // (XP: still the OUTPUT transposer of phase 3)
  uint4* wl = reinterpret_cast<uint4*>(smem + CHK * G::CHS + 256);   // XP: [2][64] staging B fragments (the same for all four waves; as 8 more
                                                                     // registers per lane the 128-register stride-1 instance spilled)
  // XP: scale | shift of the Ce <= 32 input channels, behind the tile planes (read back per staging group: as 16 registers per lane the
  // 128-register stride-1 instance spilled)
  float* xt = reinterpret_cast<float*>(smem + CHK * G::CHS);
  const bool xbn = XP && a.xe.gamma != nullptr;              // (uniform)
  const float xhi = act_hi(a.xe.act), xns = act_ns(a.xe.act, a.xe.slope);
  const bool xact = a.xe.act != SPB_ACT_NONE;
  if constexpr (XP) {
    const bf16_t* We = reinterpret_cast<const bf16_t*>(a.We);
    uint4 w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + 16 * h + r, k = 8 * q;
      w[h] = *reinterpret_cast<const uint4*>(We + (size_t)(c < C ? c : C - 1) * CX + (k < CX ? k : CX - 8));
      if (c >= C || k >= CX) w[h] = make_uint4(0, 0, 0, 0);
    }
    if (wave == 0) { wl[lane] = w[0]; wl[64 + lane] = w[1]; }
    // scale / shift of the input channels: built once per workgroup (one channel per thread: one memory round trip)
    if (threadIdx.x < 32) {
      float s_ = 1.f, h_ = 0.f;
      if (xbn) bn_fwd_coef(a.xe, (int)threadIdx.x < CX ? (int)threadIdx.x : CX - 1, s_, h_);
      xt[threadIdx.x] = s_; xt[32 + threadIdx.x] = h_;
    }
    __syncthreads();
  }
  const int xk0 = 8 * q < CX ? 8 * q : CX - 8;
  float sc[2], sh[2];      // staging: this lane's channels c0 + r and c0 + 16 + r
  if (a.pro.gamma != nullptr && !a.pro.moments) {   // (uniform) the sums of both channels requested together: one round trip, not two
    BNLoad bl[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int c = c0 + 16 * h + r; bn_issue(a.pro, c < C ? c : C - 1, bl[h]); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float mu, is;
      bn_finish(a.pro, bl[h], mu, is);
      sc[h] = bl[h].gm * is; sh[h] = bl[h].bt - mu * sc[h];
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + 16 * h + r;
      bn_fwd_coef(a.pro, c < C ? c : C - 1, sc[h], sh[h]);     // (clamped, masked below: a load inside a conditional is waited for inside it)
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (c0 + 16 * h + r >= C) { sc[h] = 0.f; sh[h] = 0.f; }
  // taps: units u = 0, 1 of this lane are channel planes 8 (r >> 2) + 4 u + (r & 3)  (the permutation the output transposer undoes)
  constexpr int NWP = ST == 1 ? 12 : 6;
  uint32_t wp[2][NWP];
  bool cok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = c0 + 8 * (r >> 2) + 4 * u + (r & 3);
    cok[u] = c < C;
    const float* wsrc = a.Wd + (size_t)(cok[u] ? c : C - 1) * 9;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = wsrc[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = cok[u] ? w[k] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float w0 = w[ky * 3], w1 = w[ky * 3 + 1], w2 = w[ky * 3 + 2];
      if (ST == 1) {
        wp[u][ky * 4 + 0] = pack_bf16x2(0.f, w0); wp[u][ky * 4 + 1] = pack_bf16x2(w1, w2);
        wp[u][ky * 4 + 2] = pack_bf16x2(w0, w1);  wp[u][ky * 4 + 3] = pack_bf16x2(w2, 0.f);
      } else {
        wp[u][ky * 2 + 0] = pack_bf16x2(0.f, w0); wp[u][ky * 2 + 1] = pack_bf16x2(w1, w2);
      }
    }
  }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const unsigned plane0 = (unsigned)((8 * (r >> 2) + (r & 3)) * G::CHS);   // unit 0; unit 1 is 4 planes further

  // the raw rows of the NEXT tile are requested as soon as this tile's are consumed (same registers): they travel during the
  // tap / store phases and the barriers
  Raw8<bf16_t> raw[G::GPW];
  auto request = [&](int t) {
    t = t < tg.ntiles ? t : tg.ntiles - 1;                   // (past the end: a valid tile again, never consumed)
    const int tx = t % tg.ntx, ty = (t / tg.ntx) % tg.nty, b = t / (tg.ntx * tg.nty);
    const int yin0 = ST * ty * TH - 1, xin0 = ST * tx * TW - 2;
    int cl = (XP ? 0 : c0) + 8 * q; cl = cl > CX - 8 ? CX - 8 : cl;
    const bf16_t* xb = X + (size_t)b * H * W * CX + cl;
#pragma unroll
    for (int i = 0; i < G::GPW; ++i) {
      int g = wave + 4 * i; g = g < G::GROUPS ? g : G::GROUPS - 1;
      const int slot = g * 16 + r;
      int y = yin0 + slot / G::ROWP, x = xin0 + slot % G::ROWP;
      y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y); x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
      raw[i] = ldraw<bf16_t>(xb + (size_t)(y * W + x) * CX);
    }
  };
  constexpr bool NOPF = XP && ST == 2 && SPB_DWT_XP_OCC3;
  if (!NOPF && sp < tg.ntiles) request(sp);
  for (int t = sp; t < tg.ntiles; t += tg.nsp) {
    if (NOPF) request(t);
    const int tx = t % tg.ntx, ty = (t / tg.ntx) % tg.nty, b = t / (tg.ntx * tg.nty);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int yin0 = ST * oy0 - 1, xin0 = ST * ox0 - 2;
    // ---- phase 1: stage the input tile: transposer + BatchNorm + activation + LDS stores
    // border tiles mask what lies outside the image (zero padding of the ACTIVATED operand); interior tiles skip the masks
    const bool border = yin0 < 0 || yin0 + G::ROWS > H || ST * ox0 - 1 < 0 || ST * (ox0 + TW - 1) + 1 > W - 1;
#pragma unroll
    for (int i = 0; i < G::GPW; ++i) {
      int g = wave + 4 * i; g = g < G::GROUPS ? g : G::GROUPS - 1;
      const int slot4 = g * 16 + 4 * q;                      // this lane's 4 result pixels: slots slot4 .. slot4+3, one tile row
      bf16x8_t av = __builtin_bit_cast(bf16x8_t, raw[i].u);
      if constexpr (XP) {
        if (xbn) {                                           // (uniform) the expand convolution's operand: round16(act(bn(x)))
          float xv[8], xsc[8], xsh[8];
          cvt8(raw[i], xv);
          *reinterpret_cast<float4*>(xsc) = *reinterpret_cast<const float4*>(xt + xk0);
          *reinterpret_cast<float4*>(xsc + 4) = *reinterpret_cast<const float4*>(xt + xk0 + 4);
          *reinterpret_cast<float4*>(xsh) = *reinterpret_cast<const float4*>(xt + 32 + xk0);
          *reinterpret_cast<float4*>(xsh + 4) = *reinterpret_cast<const float4*>(xt + 32 + xk0 + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[j] = xv[j] * xsc[j] + xsh[j];
          if (xact) {                                        // (uniform; MobileNetV2's block inputs are linear)
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = __builtin_amdgcn_fmed3f(xv[j], 0.f, xhi) + xns * fminf(xv[j], 0.f);
          }
          uint4 pk;
          pk.x = pack_bf16x2(xv[0], xv[1]); pk.y = pack_bf16x2(xv[2], xv[3]); pk.z = pack_bf16x2(xv[4], xv[5]); pk.w = pack_bf16x2(xv[6], xv[7]);
          av = __builtin_bit_cast(bf16x8_t, pk);
        }
      }
      int msk[4] = {-1, -1, -1, -1};
      if (border) {                                          // (uniform)
        const int y = yin0 + slot4 / G::ROWP, x = xin0 + slot4 % G::ROWP;
        const int ym = (unsigned)y < (unsigned)H ? -1 : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) msk[e] = ((unsigned)(x + e) < (unsigned)W ? -1 : 0) & ym;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t z = SPB_MFMA16(av, XP ? __builtin_bit_cast(bf16x8_t, wl[64 * h + lane]) : (h == 0 ? sel0 : sel1), ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float uu = z[e] * sc[h] + sh[h];
          float x = __builtin_amdgcn_fmed3f(uu, 0.f, ahi);
          if (!CLAMP) x += ans * fminf(uu, 0.f);
          v[e] = __int_as_float(__float_as_int(x) & msk[e]);
        }
        uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(smem + (16 * h + r) * G::CHS + slot4 * 2) = o;
      }
    }
    if (!NOPF) request(t + tg.nsp);
    lds_barrier();
    // ---- phase 2: taps.  Lane (r, q) of wave w: output row oyl = 4 w + q, 8 output columns, channel planes of units 0 and 1
    const int oyl = 4 * wave + q;
    float acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
      const char* pb = smem + plane0 + u * 4 * G::CHS + (ST * oyl) * G::ROWP * 2;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char* pr = pb + ky * G::ROWP * 2;
        if constexpr (ST == 1) {     // rows are 24 bytes: 8-byte aligned reads
          const uint2 d0 = *reinterpret_cast<const uint2*>(pr);
          const uint2 d1 = *reinterpret_cast<const uint2*>(pr + 8);
          const uint2 d2 = *reinterpret_cast<const uint2*>(pr + 16);
          const uint32_t D[6] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[u][2 * m] = dot2(D[m], wp[u][ky * 4 + 0], acc[u][2 * m]);
            acc[u][2 * m] = dot2(D[m + 1], wp[u][ky * 4 + 1], acc[u][2 * m]);
            acc[u][2 * m + 1] = dot2(D[m + 1], wp[u][ky * 4 + 2], acc[u][2 * m + 1]);
            acc[u][2 * m + 1] = dot2(D[m + 2], wp[u][ky * 4 + 3], acc[u][2 * m + 1]);
          }
        } else {
          const uint2 d0 = *reinterpret_cast<const uint2*>(pr);          // rows are 40 bytes: 8-byte aligned reads
          const uint2 d1 = *reinterpret_cast<const uint2*>(pr + 8);
          const uint2 d2 = *reinterpret_cast<const uint2*>(pr + 16);
          const uint2 d3 = *reinterpret_cast<const uint2*>(pr + 24);
          const uint32_t d4 = *reinterpret_cast<const uint32_t*>(pr + 32);
          const uint32_t D[9] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y, d4};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            acc[u][e] = dot2(D[e], wp[u][ky * 2 + 0], acc[u][e]);
            acc[u][e] = dot2(D[e + 1], wp[u][ky * 2 + 1], acc[u][e]);
          }
        }
      }
    }
    // ---- phase 3: batch sums, then back to NHWC through the matrix core
    const int nvx = OW - ox0 < TW ? OW - ox0 : TW;          // valid output columns of this tile
    const bool rowok = oy0 + oyl < OH;
    bf16x8_t af[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint4 p;
      p.x = pack_bf16x2(acc[u][0], acc[u][1]); p.y = pack_bf16x2(acc[u][2], acc[u][3]);
      p.z = pack_bf16x2(acc[u][4], acc[u][5]); p.w = pack_bf16x2(acc[u][6], acc[u][7]);
      af[u] = __builtin_bit_cast(bf16x8_t, p);
      const float on = (rowok && cok[u]) ? 1.f : 0.f;
      float zr[8];                                           // the sums see what is stored: the bf16-rounded values
      spb_unpack2(p.x, zr[0], zr[1]); spb_unpack2(p.y, zr[2], zr[3]); spb_unpack2(p.z, zr[4], zr[5]); spb_unpack2(p.w, zr[6], zr[7]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zz = e < nvx ? zr[e] * on : 0.f;
        s1[u] += zz; s2[u] += zz * zz;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4_t t0 = SPB_MFMA16(af[0], h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
      const f32x4_t t1 = SPB_MFMA16(af[1], h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
      // lane (r = column, q): pixel k = r + 16 h of the wave's 4 x 8 outputs, channels c0 + 8 q + (t0[0..3], t1[0..3])
      const int k = r + 16 * h;
      const int oy = oy0 + 4 * wave + (k >> 3), ox = ox0 + (k & 7);
      if (oy < OH && ox < OW && c0 + 8 * q < C) {
        uint4 o;
        o.x = pack_bf16x2(t0[0], t0[1]); o.y = pack_bf16x2(t0[2], t0[3]);
        o.z = pack_bf16x2(t1[0], t1[1]); o.w = pack_bf16x2(t1[2], t1[3]);
        *reinterpret_cast<uint4*>(Y + ((size_t)(b * OH + oy) * OW + ox) * C + c0 + 8 * q) = o;
      }
    }
    lds_barrier();   // every wave is done with the tile before the next one is staged
  }

  // ---- per-channel batch sums: over the 4 row lanes q (same channel, lanes r + 16 q), then over the waves in LDS, one atomic each
  if (a.epi_mode == 1) {
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2 units][2][16]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float x1 = xor32_sum(xor16_sum(s1[u])), x2 = xor32_sum(xor16_sum(s2[u]));
      if (q == 0) { red[((wave * 2 + u) * 2 + 0) * 16 + r] = x1; red[((wave * 2 + u) * 2 + 1) * 16 + r] = x2; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int p = threadIdx.x & 31, which = threadIdx.x >> 5;     // channel plane p, sum | sum of squares
      const int rr = 4 * (p >> 3) + (p & 3), u = (p >> 2) & 1;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[((w * 2 + u) * 2 + which) * 16 + rr];
      if (c0 + p < C) atomicAdd(a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C + (size_t)which * C + c0 + p, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Input gradient, stride 1 (the 56x56 / 28x28 / 112x112 stride-1 layers): the same tile machinery as the forward kernel.
//   dz = g p0 + z p1 + p2                  BatchNorm backward of the convolution OUTPUT, rebuilt while the tile is staged (two
//                                          16-byte loads per lane, four transposer MFMAs per 16 pixels x 32 channels)
//   dx[y][x] = sum_k w[k] dz[y+1-ky][x+1-kx]    = the forward taps with the kernel flipped
//   gin = round(dx) * act'(u_in)           mask of the input-side tensor (ReLU / ReLU6 / none: a 0/1 factor, so masking the ROUNDED
//                                          value equals rounding the masked one) + sum gin, sum gin * xhat_in          (EPI)
// The epilogue runs in NHWC after the output transposer (8 channels of one pixel per lane): the input-side z arrives as one 16-byte
// load per lane, requested before the taps.  Not covered (the caller keeps the row-unit kernel): a joining gradient (res), LeakyReLU.
template <bool EPI>
__global__ __launch_bounds__(256, 2) void dwt_dgrad1_kernel(const spb_dw_args_t a, const TileGeo tg) {
  spb_publish_entry(a.entry_flag, a.entry_val);
  typedef TG<1> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = a.C, H = a.H, W = a.W;                       // stride 1: the output map (g, z) and the input map have the same size
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int chunk = idx % tg.nchunks, sp = (idx / tg.nchunks) * 8 + xcd;
  const int c0 = chunk * CHK;
  const bf16_t* Gt = reinterpret_cast<const bf16_t*>(a.X);
  const bf16_t* Zt = reinterpret_cast<const bf16_t*>(a.X2);
  const bf16_t* Zo = reinterpret_cast<const bf16_t*>(a.Zout);
  bf16_t* Y = reinterpret_cast<bf16_t*>(a.Y);

  const bf16x8_t sel0 = selector(lane, 0), sel1 = selector(lane, 1);
  float p0[2], p1[2], p2[2];     // staging: BatchNorm-backward coefficients of this lane's channels c0 + r and c0 + 16 + r
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + 16 * h + r;
    bn_bwd_coef(a.pro, c < C ? c : C - 1, p0[h], p1[h], p2[h]);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (c0 + 16 * h + r >= C) { p0[h] = 0.f; p1[h] = 0.f; p2[h] = 0.f; }
  uint32_t wp[2][12];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = c0 + 8 * (r >> 2) + 4 * u + (r & 3);
    const bool ok = c < C;
    const float* wsrc = a.Wd + (size_t)(ok ? c : C - 1) * 9;
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = wsrc[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ok ? w[k] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {       // tile row offset ky reads kernel row 2 - ky, columns reversed
      const float w0 = w[(2 - ky) * 3 + 2], w1 = w[(2 - ky) * 3 + 1], w2 = w[(2 - ky) * 3 + 0];
      wp[u][ky * 4 + 0] = pack_bf16x2(0.f, w0); wp[u][ky * 4 + 1] = pack_bf16x2(w1, w2);
      wp[u][ky * 4 + 2] = pack_bf16x2(w0, w1);  wp[u][ky * 4 + 3] = pack_bf16x2(w2, 0.f);
    }
  }
  // epilogue: this lane's 8 channels c0 + 8 q .. + 7 of the input-side BatchNorm
  float esc[8], esh[8];
  const float ehi = act_hi(a.epi.act);
  const bool eact = a.epi.act != SPB_ACT_NONE;
  const int ce0 = c0 + 8 * q < C ? c0 + 8 * q : C - 8;
  if (EPI) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bn_fwd_coef(a.epi, ce0 + j, esc[j], esh[j]);
  }
  float sg[8], sgz[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sg[j] = 0.f; sgz[j] = 0.f; }
  const unsigned plane0 = (unsigned)((8 * (r >> 2) + (r & 3)) * G::CHS);

  Raw8<bf16_t> rg[G::GPW], rz[G::GPW];
  auto request = [&](int t) {
    t = t < tg.ntiles ? t : tg.ntiles - 1;
    const int tx = t % tg.ntx, ty = (t / tg.ntx) % tg.nty, b = t / (tg.ntx * tg.nty);
    const int yin0 = ty * TH - 1, xin0 = tx * TW - 2;
    int cl = c0 + 8 * q; cl = cl > C - 8 ? C - 8 : cl;
    const size_t ib = (size_t)b * H * W * C + cl;
#pragma unroll
    for (int i = 0; i < G::GPW; ++i) {
      int g = wave + 4 * i; g = g < G::GROUPS ? g : G::GROUPS - 1;
      const int slot = g * 16 + r;
      int y = yin0 + slot / G::ROWP, x = xin0 + slot % G::ROWP;
      y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y); x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
      const size_t o = ib + (size_t)(y * W + x) * C;
      rg[i] = ldraw<bf16_t>(Gt + o); rz[i] = ldraw<bf16_t>(Zt + o);
    }
  };
  if (sp < tg.ntiles) request(sp);
  for (int t = sp; t < tg.ntiles; t += tg.nsp) {
    const int tx = t % tg.ntx, ty = (t / tg.ntx) % tg.nty, b = t / (tg.ntx * tg.nty);
    const int y0 = ty * TH, x0 = tx * TW;
    const int yin0 = y0 - 1, xin0 = x0 - 2;
    const bool border = yin0 < 0 || yin0 + G::ROWS > H || x0 - 1 < 0 || x0 + TW > W - 1;
    // ---- phase 1: dz tile -> LDS (planar bf16, zero outside the map)
#pragma unroll
    for (int i = 0; i < G::GPW; ++i) {
      int g = wave + 4 * i; g = g < G::GROUPS ? g : G::GROUPS - 1;
      const int slot4 = g * 16 + 4 * q;
      const bf16x8_t ag = __builtin_bit_cast(bf16x8_t, rg[i].u), az = __builtin_bit_cast(bf16x8_t, rz[i].u);
      int msk[4] = {-1, -1, -1, -1};
      if (border) {
        const int y = yin0 + slot4 / G::ROWP, x = xin0 + slot4 % G::ROWP;
        const int ym = (unsigned)y < (unsigned)H ? -1 : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) msk[e] = ((unsigned)(x + e) < (unsigned)W ? -1 : 0) & ym;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t gg = SPB_MFMA16(ag, h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
        const f32x4_t zz = SPB_MFMA16(az, h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = gg[e] * p0[h] + (zz[e] * p1[h] + p2[h]);
          v[e] = __int_as_float(__float_as_int(d) & msk[e]);
        }
        uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(smem + (16 * h + r) * G::CHS + slot4 * 2) = o;
      }
    }
    request(t + tg.nsp);
    // the input-side z of this lane's two output pixels (epilogue operands): in flight during the taps
    Raw8<bf16_t> zin[2];
    if (EPI) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = r + 16 * h;
        int y = y0 + 4 * wave + (k >> 3), x = x0 + (k & 7);
        y = y > H - 1 ? H - 1 : y; x = x > W - 1 ? W - 1 : x;
        zin[h] = ldraw<bf16_t>(Zo + ((size_t)(b * H + y) * W + x) * C + ce0);
      }
    }
    lds_barrier();
    // ---- phase 2: taps (flipped kernel)
    const int yl = 4 * wave + q;
    float acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
      const char* pb = smem + plane0 + u * 4 * G::CHS + yl * G::ROWP * 2;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char* pr = pb + ky * G::ROWP * 2;
        const uint2 d0 = *reinterpret_cast<const uint2*>(pr);
        const uint2 d1 = *reinterpret_cast<const uint2*>(pr + 8);
        const uint2 d2 = *reinterpret_cast<const uint2*>(pr + 16);
        const uint32_t D[6] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[u][2 * m] = dot2(D[m], wp[u][ky * 4 + 0], acc[u][2 * m]);
          acc[u][2 * m] = dot2(D[m + 1], wp[u][ky * 4 + 1], acc[u][2 * m]);
          acc[u][2 * m + 1] = dot2(D[m + 1], wp[u][ky * 4 + 2], acc[u][2 * m + 1]);
          acc[u][2 * m + 1] = dot2(D[m + 2], wp[u][ky * 4 + 3], acc[u][2 * m + 1]);
        }
      }
    }
    // ---- phase 3: NHWC through the matrix core, mask, sums, store
    bf16x8_t af[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint4 p;
      p.x = pack_bf16x2(acc[u][0], acc[u][1]); p.y = pack_bf16x2(acc[u][2], acc[u][3]);
      p.z = pack_bf16x2(acc[u][4], acc[u][5]); p.w = pack_bf16x2(acc[u][6], acc[u][7]);
      af[u] = __builtin_bit_cast(bf16x8_t, p);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4_t t0 = SPB_MFMA16(af[0], h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
      const f32x4_t t1 = SPB_MFMA16(af[1], h == 0 ? sel0 : sel1, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
      const int k = r + 16 * h;
      const int y = y0 + 4 * wave + (k >> 3), x = x0 + (k & 7);
      const bool ok = y < H && x < W && c0 + 8 * q < C;
      float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};   // exact f32 images of the bf16-rounded results
      if (EPI) {
        float zf[8];
        cvt8(zin[h], zf);
        const float on = ok ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (eact) {
            const float uu = zf[j] * esc[j] + esh[j];
            v[j] = (uu > 0.f && uu < ehi) ? v[j] : 0.f;
          }
          const float vv = v[j] * on;
          sg[j] += vv; sgz[j] += vv * zf[j];
        }
      }
      if (ok) {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(Y + ((size_t)(b * H + y) * W + x) * C + c0 + 8 * q) = o;
      }
    }
    lds_barrier();
  }

  // ---- sum g, sum g * xhat of the input-side tensor: over the 16 pixel lanes r (same channels), over the waves in LDS
  if (EPI) {
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][4 q][16]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = row16_sum(sg[j]), x2 = row16_sum(sgz[j]);
      if (r == 0) { red[(wave * 4 + q) * 16 + j] = x1; red[(wave * 4 + q) * 16 + 8 + j] = x2; }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int c = c0 + threadIdx.x;                        // channel c0 + 8 qq + j
      const int qq = threadIdx.x >> 3, j = threadIdx.x & 7;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s1 += red[(w * 4 + qq) * 16 + j]; s2 += red[(w * 4 + qq) * 16 + 8 + j]; }
      if (c < C) {
        float mu = 0.f, is = 1.f;
        if (a.epi.gamma != nullptr) bn_moments(a.epi, c, mu, is);
        else { mu = 0.f; is = 1.f; }
        float* dst = a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C;
        atomicAdd(dst + c, s1);
        atomicAdd(dst + C + c, is * (s2 - mu * s1));       // sum g*xhat from sum g*z (dwconv_rows.hip)
      }
    }
  }
}

static int g_dw_tile_min = 28;     // bf16 maps at least this wide use the tile kernels (spb_debug_set_dw_tile; 0 = never)
static int g_dw_tile_wgs = 0;      // workgroups per launch (0: resident estimate)
static int g_dw_tile_dgrad = 0;    // stride-1 input gradient on the tile kernel: OFF by default -- measured slower than the row-unit kernel
                                   // (56x56x144: 71 vs 62 us, 28x28x192: 35 vs 31 us): its NHWC epilogue (mask + two sums per element) and
                                   // 184 registers (two workgroups per CU) eat what the cheaper taps save; kept as a tested instance

}  // namespace

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_tile(int min_width, int workgroups) {
  g_dw_tile_min = min_width <= 0 ? (1 << 30) : min_width;
  g_dw_tile_wgs = workgroups < 0 ? 0 : workgroups & 0xffff;
  g_dw_tile_dgrad = workgroups >= 0 && (workgroups >> 16) != 0;     // bit 16: also the stride-1 input gradient
  return 0;
}
#endif

// SPB_E_UNSUPPORTED: not covered (the caller keeps the row-unit / plane kernels)
int spb_dwt_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const bool xp = a->Xe != nullptr;
  if (xp) {   // expand recompute: this kernel or nothing (the plain kernels would read a tensor that does not exist)
    if (dtype != SPB_BF16 || a->W < 28 || a->H < 28 || (a->C & 7) || !a->We || a->Ce < 8 || a->Ce > 32 || (a->Ce & 7)) return SPB_E_UNSUPPORTED;
  } else if (dtype != SPB_BF16 || a->W < g_dw_tile_min || a->H < g_dw_tile_min || (a->C & 7)) return SPB_E_UNSUPPORTED;
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  TileGeo tg;
  tg.nty = (OH + TH - 1) / TH; tg.ntx = (OW + TW - 1) / TW; tg.ntiles = a->B * tg.nty * tg.ntx;
  tg.nchunks = (a->C + CHK - 1) / CHK;
  const size_t lds = (size_t)CHK * (st == 1 ? TG<1>::CHS : TG<2>::CHS) + (xp ? 256 + 2048 : 0);
  const int per_cu = st == 1 ? 4 : ((xp && SPB_DWT_XP_OCC3) ? 3 : 2);   // resident workgroups per CU (registers: 100 / 172 per lane)
  int target = g_dw_tile_wgs > 0 ? g_dw_tile_wgs : 256 * per_cu;   // persistent: one dispatch round
  int per_xcd = target / 8 / tg.nchunks;                     // spatial walkers per XCD
  const int maxsp = (tg.ntiles + 7) / 8;
  if (per_xcd > maxsp) per_xcd = maxsp;
  if (per_xcd < 1) per_xcd = 1;
  tg.nsp = per_xcd * 8;
  const dim3 grid((unsigned)(tg.nsp * tg.nchunks));
  const bool clamp = a->pro.act == SPB_ACT_RELU || a->pro.act == SPB_ACT_RELU6;
#define DWT_(ST_, CL_, XP_)                                                                                                    \
  {                                                                                                                            \
    static bool once = false;                                                                                                  \
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dwt_fwd_kernel<ST_, CL_, XP_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; } \
    hipLaunchKernelGGL((dwt_fwd_kernel<ST_, CL_, XP_>), grid, dim3(256), lds, s, *a, tg);                                      \
  }
  if (xp) {   // (the expanded tensor of MobileNetV2 carries ReLU6: the clamp form; anything else takes the general activation)
    if (st == 1) { if (clamp) DWT_(1, true, true) else DWT_(1, false, true) }
    else { if (clamp) DWT_(2, true, true) else DWT_(2, false, true) }
  } else if (st == 1) { if (clamp) DWT_(1, true, false) else DWT_(1, false, false) }
  else { if (clamp) DWT_(2, true, false) else DWT_(2, false, false) }
#undef DWT_
  return 0;
}

// input gradient (stride 1, no joining gradient, ReLU-type or no activation on the input side); SPB_E_UNSUPPORTED otherwise
int spb_dwt_dgrad(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  if (!g_dw_tile_dgrad || dtype != SPB_BF16 || a->stride != 1 || a->W < g_dw_tile_min || a->H < g_dw_tile_min || (a->C & 7)) return SPB_E_UNSUPPORTED;
  if (a->dW != nullptr || a->res != nullptr) return SPB_E_UNSUPPORTED;
  const bool epi = a->epi_mode == 2;
  if (epi && a->epi.act == SPB_ACT_LEAKY) return SPB_E_UNSUPPORTED;
  TileGeo tg;
  tg.nty = (a->H + TH - 1) / TH; tg.ntx = (a->W + TW - 1) / TW; tg.ntiles = a->B * tg.nty * tg.ntx;
  tg.nchunks = (a->C + CHK - 1) / CHK;
  const size_t lds = (size_t)CHK * TG<1>::CHS;
  int target = g_dw_tile_wgs > 0 ? g_dw_tile_wgs : 256 * 2;
  int per_xcd = target / 8 / tg.nchunks;
  const int maxsp = (tg.ntiles + 7) / 8;
  if (per_xcd > maxsp) per_xcd = maxsp;
  if (per_xcd < 1) per_xcd = 1;
  tg.nsp = per_xcd * 8;
  const dim3 grid((unsigned)(tg.nsp * tg.nchunks));
  if (epi) hipLaunchKernelGGL((dwt_dgrad1_kernel<true>), grid, dim3(256), lds, s, *a, tg);
  else hipLaunchKernelGGL((dwt_dgrad1_kernel<false>), grid, dim3(256), lds, s, *a, tg);
  return 0;
}
