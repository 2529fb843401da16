// MobileNetV2 stem (Conv2d(3,32,3,stride 2,pad 1), features[0][0]; reference call site park2019.py:107-108) as an
// implicit GEMM on the matrix cores, bf16 mode.  The scalar kernels in stem_head.hip spend 50 VALU instructions per
// output value (216 FMAs + 54 LDS weight reads per pixel and 8 channels) and had become the largest single launches of
// the step (stem_wgrad 276 us, stem_fwd 121 us at B=48); as a [pixels x 27] x [27 x 32] product the stem is 2 MFMAs per
// 16 pixels and the kernels stream at HBM speed.
//   forward        y^T[co, p]   = W[co, tap] * patch^T[tap, p]          (K = 27 taps padded to 32)
//                  -> each lane ends up with 4 consecutive channels of one pixel: an 8-byte NHWC store, no transpose
//   weight grad    dW[co, tap] += dz^T[co, p] * patch[p, tap]           (K = 32 pixels per step)
//                  dz^T comes from a wave-private LDS tile via the transpose load, the patches are gathered from x
// x is the fp32 NCHW image.  FORWARD (round 6): image and weights enter the matrix cores as a PAIR of 16-bit values each, hi = round16(v) and
// lo = round16(v - hi), and the product is W_hi x_hi + W_hi x_lo + W_lo x_hi (f32 accumulation; the dropped W_lo x_lo is 2^-16 of it): the stem convolution is
// exact to ~1e-5 instead of carrying the 2^-9 relative rounding of a bf16 image.  Why it matters: the network's INPUT is the one operand whose rounding the whole
// network amplifies -- on the "chaotic" conditioned state of tests/test_parity_conditioned_gpu.py the float64 CPU restatement of the network with bf16 rounding at all the HIP path's
// points has gradient cosine 0.49 to float64, and 0.90 with just the stem's operands left unrounded (scratch/bf16_state_analysis_r6.py,
// profiles/r6_bf16_state_analysis.txt).  The kernel is bound by its bytes; the two extra MFMAs per product cost nothing measurable.  The weight gradient still
// takes the image rounded (it only feeds the stem's own 864 weight gradients).
#include "common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ bf16x8_t pack8(const float v[8]) {
  uint4 u;
  u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
  u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
  return __builtin_bit_cast(bf16x8_t, u);
}

// hi / lo pair of 8 floats in the library's 16-bit format: hi = round16(v), lo = round16(v - hi)
__device__ __forceinline__ void pack8_split(const float v[8], bf16x8_t& hi, bf16x8_t& lo) {
  unsigned hp[4], lp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned q = pack_bf16x2(v[2 * k], v[2 * k + 1]);
    float a, b;
    spb_unpack2(q, a, b);
    hp[k] = q; lp[k] = pack_bf16x2(v[2 * k] - a, v[2 * k + 1] - b);
  }
  hi = __builtin_bit_cast(bf16x8_t, make_uint4(hp[0], hp[1], hp[2], hp[3]));
  lo = __builtin_bit_cast(bf16x8_t, make_uint4(lp[0], lp[1], lp[2], lp[3]));
}

__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* tile, int LD, int c0, int li, int lq) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  const bf16_t* p = tile + (lq * 8 + (li >> 2)) * LD + c0 + (li & 3) * 4;
  union { struct { s16x4_t lo, hi; } s; bf16x8_t v; } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * LD));
  return u.v;
}

// ------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            bf16_t* __restrict__ y, float* osums, int oR, int B, int H, int W) {
  __shared__ float red[4][2][32];
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4, wave = threadIdx.x >> 6;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long P = (long long)B * OH * OW;
  const long long groups = (P + 15) / 16;
  // A operand: W[co = cb*16+li][tap = lq*8+e], taps >= 27 are zero
  bf16x8_t Wa[2], Wl[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = lq * 8 + e;
      v[e] = t < 27 ? w[(cb * 16 + li) * 27 + t] : 0.f;
    }
    pack8_split(v, Wa[cb], Wl[cb]);
  }
  // the 8 taps this lane gathers for its pixel: (ci, ky, kx) of tap lq*8+e
  int tci[8], tky[8], tkx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int t = lq * 8 + e, tt = t < 27 ? t : 26;
    tci[e] = t < 27 ? tt / 9 : -1; tky[e] = (tt % 9) / 3; tkx[e] = tt % 3;
  }
  float s1[2][4], s2[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[cb][e] = 0.f; s2[cb][e] = 0.f; }

  auto gather = [&](long long gi, float v[8]) {
    long long p = gi * 16 + li;
    p = p < P ? p : P - 1;
    const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), b = (int)(p / ((long long)OW * OH));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ih = 2 * oh - 1 + tky[e], iw = 2 * ow - 1 + tkx[e];
      const bool ok = tci[e] >= 0 && ih >= 0 && ih < H && iw >= 0 && iw < W;
      const float xv = x[((size_t)(b * 3 + (tci[e] < 0 ? 0 : tci[e])) * H + clampi(ih, 0, H - 1)) * W + clampi(iw, 0, W - 1)];
      v[e] = ok ? xv : 0.f;
    }
  };

  const long long gstride = (long long)gridDim.x * 4;
  long long gi = (long long)blockIdx.x * 4 + wave;
  float nxt[8];
  if (gi < groups) gather(gi, nxt);
  for (; gi < groups; gi += gstride) {
    bf16x8_t bf, bl;
    pack8_split(nxt, bf, bl);
    if (gi + gstride < groups) gather(gi + gstride, nxt);   // next group's loads fly during the MFMAs / stores
    const long long p = gi * 16 + li;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      acc = SPB_MFMA16(Wl[cb], bf, acc);       // (small terms first)
      acc = SPB_MFMA16(Wa[cb], bl, acc);
      acc = SPB_MFMA16(Wa[cb], bf, acc);
      // lane (li, lq): pixel p, channels cb*16 + lq*4 .. +3
      uint2 o;
      o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
      if (p < P) {
        *reinterpret_cast<uint2*>(y + (size_t)p * 32 + cb * 16 + lq * 4) = o;
        float r0, r1, r2, r3;
        spb_unpack2(o.x, r0, r1); spb_unpack2(o.y, r2, r3);
        s1[cb][0] += r0; s1[cb][1] += r1; s1[cb][2] += r2; s1[cb][3] += r3;
        s2[cb][0] += r0 * r0; s2[cb][1] += r1 * r1; s2[cb][2] += r2 * r2; s2[cb][3] += r3 * r3;
      }
    }
  }
  if (osums) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = row16_sum(s1[cb][e]), b2 = row16_sum(s2[cb][e]);
        if (li == 0) { red[wave][0][cb * 16 + lq * 4 + e] = a; red[wave][1][cb * 16 + lq * 4 + e] = b2; }
      }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
      atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + which * 32 + c, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ forward, LDS tile
// The gather kernel above spends 327 vector instructions per 16 pixels on index arithmetic (64-bit pixel -> (image, row, column)
// divisions, clamps and masks of eight scattered 4-byte loads per lane): 46 us for 67 MB (profiles/r4_sq_stall_breakdown.txt).  Here a
// workgroup owns SR output rows of one image: the 2*SR+1 input rows of the three planes are fetched with 16-byte loads (one round trip),
// rounded to bf16 and laid out in LDS as [row][column + 1][4] -- three channels and a zero per column, a zero column on each side, zero
// rows outside the image -- so the taps of a pixel are contiguous: K is ordered (ky, column 0..3, channel 0..3) = 3 x 16 (+ 16 of zero
// weights), and the 8 values a lane feeds to one MFMA are ONE aligned ds_read_b128 (columns 2*ow, 2*ow+1 or 2*ow+2, 2*ow+3 of row
// 2*oh + ky).  Four MFMAs per 16 pixels (K = 64) instead of two, no masks, no packing.  The weight rows are permuted so that a lane ends
// up with 8 consecutive channels of its pixel: one 16-byte NHWC store, 1 KB contiguous per wave.
// Round 6: the image enters as hi + lo halves (header).  The hi tile's spare slot carries the low half of channel 0, a second tile of 4 bytes per column the low
// halves of channels 1 and 2: 12 bytes per column, 46.5 KB for a band of 8 output rows -- three workgroups per CU, so the 672 workgroups of a bs=48 launch are
// resident at once (with two full 8-byte tiles, 62 KB, the launch ran in two rounds: 41 us in the step instead of 20).
template <int SR>
__global__ __launch_bounds__(256) void stem_fwd_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            bf16_t* __restrict__ y, float* osums, int oR, int B, int H, int W,
                                                            int OH, int OW, int NG, int TW) {
  extern __shared__ __attribute__((aligned(16))) char tile[];    // [2*SR+1][TW] x 8 bytes (c0h, c1h, c2h, c0l), then [2*SR+1][TW] x 4 bytes (c1l, c2l)
  __shared__ float red[4][2][32];
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int nbands = (OH + SR - 1) / SR;
  const int b = blockIdx.x / nbands, oh0 = (blockIdx.x % nbands) * SR;
  const int RI = 2 * SR + 1, ih0 = 2 * oh0 - 1;
  char* tile_lo = tile + (size_t)RI * TW * 8;
  // ---- staging: task = (input row, 4-column group); the three planes' float4 of a task become four 8-byte LDS entries
  const int ncg = W >> 2, ntask = RI * ncg;
  for (int t0 = 0; t0 < ntask; t0 += 256 * 4) {
    float4 v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int task = t0 + u * 256 + t, tc = task < ntask ? task : ntask - 1;
      const int r = tc / ncg, c4 = tc % ncg;
      const int ih = clampi(ih0 + r, 0, H - 1);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) v[u][ci] = *reinterpret_cast<const float4*>(x + ((size_t)(b * 3 + ci) * H + ih) * W + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int task = t0 + u * 256 + t;
      if (task < ntask) {
        const int r = task / ncg, c4 = task % ncg;
        const bool rowok = ih0 + r >= 0 && ih0 + r < H;
        const float a0[4] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w};
        const float a1[4] = {v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
        const float a2[4] = {v[u][2].x, v[u][2].y, v[u][2].z, v[u][2].w};
        uint2* dst = reinterpret_cast<uint2*>(tile) + (size_t)r * TW + c4 * 4 + 1;
        unsigned* dlo = reinterpret_cast<unsigned*>(tile_lo) + (size_t)r * TW + c4 * 4 + 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // hi entry: the three channels rounded, and in the slot that used to be zero the LOW half of channel 0 (it rides along in the hi
          // products: its weight row repeats channel 0's); lo entry: the low halves of channels 1 and 2
          uint2 q;
          q.x = pack_bf16x2(a0[c], a1[c]);
          float h0, h1, h2, h3;
          spb_unpack2(q.x, h0, h1);
          q.y = pack_bf16x2(a2[c], a0[c] - h0);
          spb_unpack2(q.y, h2, h3);
          const unsigned ql = pack_bf16x2(a1[c] - h1, a2[c] - h2);
          dst[c] = rowok ? q : make_uint2(0u, 0u);
          dlo[c] = rowok ? ql : 0u;
        }
      }
    }
  }
  for (int i = t; i < RI * (TW - W); i += 256) {        // the zero columns: column 0 (iw = -1) and W+1 .. TW-1
    const int r = i / (TW - W), cc = i % (TW - W);
    reinterpret_cast<uint2*>(tile)[(size_t)r * TW + (cc == 0 ? 0 : W + cc)] = make_uint2(0u, 0u);
    reinterpret_cast<unsigned*>(tile_lo)[(size_t)r * TW + (cc == 0 ? 0 : W + cc)] = 0u;
  }
  // ---- A operand: Wa[cb][chunk], row li of block cb = output channel (li / 4) * 8 + cb * 4 + li % 4; k group gi = chunk * 4 + lq holds
  // (ky = gi / 2, columns 2 * (gi % 2) + {0, 1}, slots 0..3 = channels 0, 1, 2 and the low half of channel 0); ky == 3 and column 3 are zero weights.
  // Wa: hi halves of the weights (slot 3 repeats channel 0's); Wl: their lo halves against the hi image (slot 3 zero);
  // Wr[cb]: the hi weights against the low halves of channels 1, 2 -- ONE 32-deep step: k group lq = kernel row ky (3: zero), e = (kx = e / 2, channel 1 + e % 2)
  bf16x8_t Wa[2][2], Wl[2][2], Wr[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int co = (li >> 2) * 8 + cb * 4 + (li & 3);
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int gi = ch * 4 + lq, ky = gi >> 1;
      float v[8], vl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kx = 2 * (gi & 1) + (e >> 2), ci = e & 3;
        const bool ok = ky < 3 && kx < 3;
        v[e] = ok ? w[co * 27 + (ci == 3 ? 0 : ci) * 9 + ky * 3 + kx] : 0.f;
      }
      bf16x8_t lo_all;
      pack8_split(v, Wa[cb][ch], lo_all);
      // the lo weights see only the hi image: slot 3 (the image's low half of channel 0) gets a zero weight there
#pragma unroll
      for (int e = 0; e < 8; ++e) vl[e] = (e & 3) == 3 ? 0.f : v[e];
      bf16x8_t hi_unused;
      pack8_split(vl, hi_unused, Wl[cb][ch]);
    }
    float vr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kx = e >> 1, ci = 1 + (e & 1);
      vr[e] = (lq < 3 && kx < 3) ? w[co * 27 + ci * 9 + lq * 3 + kx] : 0.f;
    }
    Wr[cb] = pack8(vr);
  }
  unsigned loff[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int gi = ch * 4 + lq, ky = gi >> 1 < 3 ? gi >> 1 : 2;
    loff[ch] = (unsigned)((ky * TW + 2 * li + 2 * (gi & 1)) * 8);
  }
  const unsigned roff = (unsigned)(((lq < 3 ? lq : 2) * TW + 2 * li) * 4);      // lo tile: kernel row lq, columns 2 li .. 2 li + 3
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  __syncthreads();
  for (int g = wave; g < SR * NG; g += 4) {
    const int ohl = g / NG, owg = g % NG;
    const unsigned base = (unsigned)((2 * ohl * TW + 32 * owg) * 8);
    bf16x8_t pf[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) pf[ch] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(tile + base + loff[ch]));
    const char* lp = tile_lo + (base >> 1) + roff;
    const uint2 r0 = *reinterpret_cast<const uint2*>(lp), r1 = *reinterpret_cast<const uint2*>(lp + 8);
    const bf16x8_t pr = __builtin_bit_cast(bf16x8_t, make_uint4(r0.x, r0.y, r1.x, r1.y));
    f32x4_t acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      acc[cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      acc[cb] = SPB_MFMA16(Wr[cb], pr, acc[cb]);            // small terms first
      acc[cb] = SPB_MFMA16(Wl[cb][0], pf[0], acc[cb]);
      acc[cb] = SPB_MFMA16(Wl[cb][1], pf[1], acc[cb]);
      acc[cb] = SPB_MFMA16(Wa[cb][0], pf[0], acc[cb]);
      acc[cb] = SPB_MFMA16(Wa[cb][1], pf[1], acc[cb]);
    }
    // lane (li, lq): pixel ow = owg * 16 + li, channels lq * 8 + cb * 4 + i
    uint4 o;
    o.x = pack_bf16x2(acc[0][0], acc[0][1]); o.y = pack_bf16x2(acc[0][2], acc[0][3]);
    o.z = pack_bf16x2(acc[1][0], acc[1][1]); o.w = pack_bf16x2(acc[1][2], acc[1][3]);
    const int oh = oh0 + ohl, ow = owg * 16 + li;
    const bool ok = oh < OH && ow < OW;
    if (ok) *reinterpret_cast<uint4*>(y + (((size_t)b * OH + oh) * OW + ow) * 32 + lq * 8) = o;
    const unsigned q[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r0, r1;
      spb_unpack2(q[e], r0, r1);
      r0 = ok ? r0 : 0.f; r1 = ok ? r1 : 0.f;
      s1[2 * e] += r0; s1[2 * e + 1] += r1;
      s2[2 * e] += r0 * r0; s2[2 * e + 1] += r1 * r1;
    }
  }
  if (osums) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = row16_sum(s1[e]), b2 = row16_sum(s2[e]);
      if (li == 0) { red[wave][0][lq * 8 + e] = a; red[wave][1][lq * 8 + e] = b2; }
    }
    __syncthreads();
    if (t < 64) {
      const int which = t >> 5, c = t & 31;
      const float v = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
      atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + which * 32 + c, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// requires OW % 8 == 0 (8 consecutive pixels of a lane lie in one output row)
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ G,
                                                              const bf16_t* __restrict__ Z, const spb_bnref_t pro, float* dW,
                                                              int B, int H, int W) {
  constexpr int LD = 40;
  __shared__ __attribute__((aligned(16))) bf16_t dzt_all[4][32 * LD];
  __shared__ __attribute__((aligned(16))) float cf[3][32];
  __shared__ float red[4][32 * 32];
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4, wave = threadIdx.x >> 6;
  if (threadIdx.x < 32) {
    float p0, p1, p2;
    bn_bwd_coef(pro, threadIdx.x, p0, p1, p2);
    cf[0][threadIdx.x] = p0; cf[1][threadIdx.x] = p1; cf[2][threadIdx.x] = p2;
  }
  __syncthreads();
  bf16_t* dzt = dzt_all[wave];
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long P = (long long)B * OH * OW;
  const long long groups = (P + 31) / 32;
  // patch operand: lane column = tap tb*16+li, rows = pixels lq*8 .. lq*8+7
  int tci[2], tky[2], tkx[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int t = tb * 16 + li, tt = t < 27 ? t : 26;
    tci[tb] = t < 27 ? tt / 9 : -1; tky[tb] = (tt % 9) / 3; tkx[tb] = tt % 3;
  }
  const int px = lane >> 2, part = lane & 3;   // dz tile load: pixels px and px+16, channels part*8..+7
  float c0[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { c0[j] = cf[0][part * 8 + j]; c1[j] = cf[1][part * 8 + j]; c2[j] = cf[2][part * 8 + j]; }

  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  Raw8<bf16_t> gr[2], zr[2];
  float xr[2][8];
  auto load = [&](long long gi) {
    const long long p0 = gi * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long long p = p0 + px + 16 * i;
      p = p < P ? p : P - 1;
      gr[i] = ldraw<bf16_t>(G + (size_t)p * 32 + part * 8);
      zr[i] = ldraw<bf16_t>(Z + (size_t)p * 32 + part * 8);
    }
    long long pg = p0 + lq * 8;                 // first of this lane's 8 pixels (same output row: OW % 8 == 0)
    pg = pg < P ? pg : P - 8;
    const int ow0 = (int)(pg % OW), oh = (int)((pg / OW) % OH), b = (int)(pg / ((long long)OW * OH));
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const int ih = 2 * oh - 1 + tky[tb];
      const bool rowok = tci[tb] >= 0 && ih >= 0 && ih < H;
      const float* row = x + ((size_t)(b * 3 + (tci[tb] < 0 ? 0 : tci[tb])) * H + clampi(ih, 0, H - 1)) * W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int iw = 2 * (ow0 + e) - 1 + tkx[tb];
        const float v = row[clampi(iw, 0, W - 1)];
        xr[tb][e] = (rowok && iw >= 0 && iw < W) ? v : 0.f;
      }
    }
  };

  const long long gstride = (long long)gridDim.x * 4;
  long long gi = (long long)blockIdx.x * 4 + wave;
  if (gi < groups) load(gi);
  for (; gi < groups; gi += gstride) {
    const long long p0 = gi * 32;
    // dz tile (bf16, row-major [pixel][co]) for the transpose load
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float gf[8], zf[8], v[8];
      cvt8(gr[i], gf); cvt8(zr[i], zf);
      const bool ok = p0 + px + 16 * i < P;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ok ? gf[j] * c0[j] + zf[j] * c1[j] + c2[j] : 0.f;
      *reinterpret_cast<bf16x8_t*>(dzt + (px + 16 * i) * LD + part * 8) = pack8(v);
    }
    bf16x8_t pf[2];
    {
      const bool gok = p0 + lq * 8 < P;   // whole 8-pixel runs are valid or not (P % 8 == 0 when OW % 8 == 0)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gok ? xr[tb][e] : 0.f;
        pf[tb] = pack8(v);
      }
    }
    if (gi + gstride < groups) load(gi + gstride);
    asm volatile("" ::: "memory");   // the tile stores above must stay ahead of the transpose loads
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const bf16x8_t af = tr_frag(dzt, LD, cb * 16, li, lq);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) acc[cb][tb] = SPB_MFMA16(af, pf[tb], acc[cb][tb]);
    }
    asm volatile("" ::: "memory");
  }
  // C layout: column = tap tb*16+li, rows = co cb*16 + lq*4 + e
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][(cb * 16 + lq * 4 + e) * 32 + tb * 16 + li] = acc[cb][tb][e];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int co = i >> 5, t = i & 31;
    if (t < 27) atomicAdd(dW + co * 27 + t, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient, LDS tile
// Same idea as stem_fwd_tile_kernel for dW[co, tap] += dz^T[co, p] * patch[p, tap] (K = pixels): the patch operand of a lane is one tap at 8
// consecutive pixels of a row, i.e. input columns 2*ow + kx - 1 at stride 2.  The workgroup's input rows are therefore staged as NINE bf16
// planes [channel][kx][row][ow] = x[channel][row][2*ow + kx - 1] (zero outside the image), which makes that operand one aligned
// ds_read_b128 -- the gather kernel spends 8 masked, clamped 4-byte global loads and their index arithmetic on it (8e6 vector instructions
// per launch, 63 us for 106 MB, and it is the LAST kernel of the backward pass: nothing runs beside it).  dz comes from the wave-private
// transposed tile exactly as in the gather kernel.  Requires OW % 8 == 0, W % 4 == 0, H and W even.
template <int WSR>
__global__ __launch_bounds__(256) void stem_wgrad_tile_kernel(const float* __restrict__ x, const bf16_t* __restrict__ G,
                                                              const bf16_t* __restrict__ Z, const spb_bnref_t pro, float* dW,
                                                              int B, int H, int W, int OH, int OW, int PW) {
  constexpr int LD = 40, RI = 2 * WSR + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* planes = reinterpret_cast<bf16_t*>(smem);                       // [3][3][RI][PW]
  bf16_t* dzt_all = planes + (size_t)9 * RI * PW;                         // [4][32 * LD]
  float* cf = reinterpret_cast<float*>(dzt_all + 4 * 32 * LD);            // [3][32]
  float* red = cf + 96;                                                   // [4][32 * 32]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int nbands = (OH + WSR - 1) / WSR;
  const int b = blockIdx.x / nbands, oh0 = (blockIdx.x % nbands) * WSR;
  const int ih0 = 2 * oh0 - 1;
  const int rows = min(WSR, OH - oh0);                                     // output rows of this band
  const size_t plane = (size_t)RI * PW;
  // ---- staging (one round trip: all loads of a thread first)
  const int ncg = W >> 2, ntask = RI * ncg;
  for (int t0 = 0; t0 < ntask; t0 += 256 * 4) {
    float4 v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int task = t0 + u * 256 + t, tc = task < ntask ? task : ntask - 1;
      const int r = tc / ncg, c4 = tc % ncg;
      const int ih = clampi(ih0 + r, 0, H - 1);
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) v[u][ci] = *reinterpret_cast<const float4*>(x + ((size_t)(b * 3 + ci) * H + ih) * W + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int task = t0 + u * 256 + t;
      if (task < ntask) {
        const int r = task / ncg, c4 = task % ncg;
        const bool rowok = ih0 + r >= 0 && ih0 + r < H;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float4 q = v[u][ci];
          bf16_t* p0 = planes + (size_t)(ci * 3) * plane + (size_t)r * PW + c4 * 2;      // kx = 0: ow = c4*2 + 1, c4*2 + 2
          const unsigned even = rowok ? pack_bf16x2(q.x, q.z) : 0u, odd = rowok ? pack_bf16x2(q.y, q.w) : 0u;
          *reinterpret_cast<unsigned*>(p0 + plane) = even;                               // kx = 1: ow = c4*2, c4*2 + 1
          *reinterpret_cast<unsigned*>(p0 + 2 * plane) = odd;                            // kx = 2: ow = c4*2, c4*2 + 1
          p0[1] = (bf16_t)(odd & 0xffffu); p0[2] = (bf16_t)(odd >> 16);
        }
      }
    }
  }
  for (int i = t; i < 3 * RI; i += 256) planes[(size_t)((i / RI) * 3) * plane + (size_t)(i % RI) * PW] = 0;   // iw = -1
  if (t < 32) {
    float p0, p1, p2;
    bn_bwd_coef(pro, t, p0, p1, p2);
    cf[t] = p0; cf[32 + t] = p1; cf[64 + t] = p2;
  }
  __syncthreads();
  bf16_t* dzt = dzt_all + wave * 32 * LD;
  // patch operand: lane column = tap tb*16+li, rows = the 8 pixels of run lq
  unsigned toff[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int tp = tb * 16 + li, tt = tp < 27 ? tp : 26;
    const int ci = tt / 9, ky = (tt % 9) / 3, kx = tt % 3;
    toff[tb] = (unsigned)(((size_t)(ci * 3 + kx) * plane + (size_t)ky * PW) * 2);
  }
  const int px = lane >> 2, part = lane & 3;   // dz tile load: pixels px and px+16, channels part*8..+7
  float c0[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { c0[j] = cf[part * 8 + j]; c1[j] = cf[32 + part * 8 + j]; c2[j] = cf[64 + part * 8 + j]; }
  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) acc[a][bb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int npix = rows * OW, ngroups = (npix + 31) / 32;       // pixels of the band in row-major order; a run of 8 lies in one row
  const size_t pbase = ((size_t)b * OH + oh0) * OW;
  Raw8<bf16_t> gr[2], zr[2];
  auto load = [&](int g) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int p = g * 32 + px + 16 * i;
      p = p < npix ? p : npix - 1;
      gr[i] = ldraw<bf16_t>(G + (pbase + p) * 32 + part * 8);
      zr[i] = ldraw<bf16_t>(Z + (pbase + p) * 32 + part * 8);
    }
  };
  int g = wave;
  if (g < ngroups) load(g);
  for (; g < ngroups; g += 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float gf[8], zf[8], v[8];
      cvt8(gr[i], gf); cvt8(zr[i], zf);
      const bool ok = g * 32 + px + 16 * i < npix;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ok ? gf[j] * c0[j] + zf[j] * c1[j] + c2[j] : 0.f;
      *reinterpret_cast<bf16x8_t*>(dzt + (px + 16 * i) * LD + part * 8) = pack8(v);
    }
    if (g + 4 < ngroups) load(g + 4);
    int run = g * 4 + lq;                                        // this lane's run of 8 pixels
    run = run * 8 < npix ? run : 0;                              // beyond the band: dz is zero there, any finite patch will do
    const int ohl = run / (OW >> 3), ow0 = (run % (OW >> 3)) * 8;
    const unsigned poff = (unsigned)(((size_t)(2 * ohl) * PW + ow0) * 2);
    bf16x8_t pf[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) pf[tb] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + toff[tb] + poff));
    asm volatile("" ::: "memory");   // the tile stores above must stay ahead of the transpose loads
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const bf16x8_t af = tr_frag(dzt, LD, cb * 16, li, lq);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) acc[cb][tb] = SPB_MFMA16(af, pf[tb], acc[cb][tb]);
    }
    asm volatile("" ::: "memory");
  }
  // C layout: column = tap tb*16+li, rows = co cb*16 + lq*4 + e
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave * 1024 + (cb * 16 + lq * 4 + e) * 32 + tb * 16 + li] = acc[cb][tb][e];
  __syncthreads();
  for (int i = t; i < 32 * 32; i += 256) {
    const int co = i >> 5, tp = i & 31;
    if (tp < 27) atomicAdd(dW + co * 27 + tp, (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]));
  }
}

}  // namespace

static int g_stem_grid_fwd = 1024, g_stem_grid_wgrad = 512;   // measured: fwd 70 / 44 / 47 / 50 us at 512 / 1024 / 2048 / 4096
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_stem_grid(int fwd, int wgrad) {
  if (fwd > 0) g_stem_grid_fwd = fwd;
  if (wgrad > 0) g_stem_grid_wgrad = wgrad;
  return 0;
}
#endif

static int g_stem_tile = 1;     // 1: LDS-tile kernels, forward with 8 output rows per workgroup (46.5 KB of LDS); 2: 4 rows; 0: gather kernels
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_stem_tile(int on) { g_stem_tile = on; return 0; }
#endif

int spb_stem_fwd_mfma(const float* x, const float* w, void* y, float* osums, int oR, int B, int H, int W, hipStream_t s) {
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  {   // LDS-tile kernel: 16-byte input loads (W % 4 == 0), the band's two tiles within the default LDS limit
    const int NG = (OW + 15) / 16, TW = (32 * NG + 4 > W + 2 ? 32 * NG + 4 : W + 2);
    const int sr = g_stem_tile == 2 ? 4 : 8;      // step, A/B pairs: 8 rows 2.421 / 2.415 ms, 7 rows 2.423 / 2.421, 6 rows 2.422 / 2.419, 4 rows 2.427 / 2.425
    size_t lds = (size_t)(2 * sr + 1) * TW * 12;                // hi tile 8 bytes per column, lo tile 4
    int srx = sr;
    if (lds > 64 * 1024 - 1024 && sr == 8) { srx = 4; lds = (size_t)(2 * srx + 1) * TW * 12; }   // wide images (--input_shape): bands of 4 rows still fit
    if (g_stem_tile && (W & 3) == 0 && lds <= 64 * 1024 - 1024) {     // (+ the kernel's 1 KB of static LDS)
      const int nbands = (OH + srx - 1) / srx;
#define SPB_STEM_FWD(SRV) hipLaunchKernelGGL(stem_fwd_tile_kernel<SRV>, dim3((unsigned)(B * nbands)), dim3(256), lds, s, x, w, (bf16_t*)y, osums, oR, B, H, W, OH, OW, NG, TW)
      if (srx == 8) SPB_STEM_FWD(8); else SPB_STEM_FWD(4);
#undef SPB_STEM_FWD
      return 0;
    }
  }
  const long long groups = ((long long)B * OH * OW + 15) / 16;
  long long grid = (groups + 3) / 4;
  // the gather keeps only 8 scalar loads per lane in flight: 4 workgroups per CU instead of 2 hide more of the latency
  if (grid > g_stem_grid_fwd) grid = g_stem_grid_fwd;
  hipLaunchKernelGGL(stem_fwd_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, w, (bf16_t*)y, osums, oR, B, H, W);
  return 0;
}

template <int WSR>
static int launch_wgrad_tile(const float* x, const void* G, const void* Z, const spb_bnref_t* pro, float* dW, int B, int H, int W, hipStream_t s) {
  const int OH = H / 2, OW = W / 2, PW = OW + 8;
  const size_t lds = (size_t)9 * (2 * WSR + 1) * PW * 2 + 4 * 32 * 40 * 2 + 96 * 4 + 4 * 1024 * 4;
  if (lds > 160 * 1024) return SPB_E_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_tile_kernel<WSR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int nbands = (OH + WSR - 1) / WSR;
  hipLaunchKernelGGL(stem_wgrad_tile_kernel<WSR>, dim3((unsigned)(B * nbands)), dim3(256), lds, s, x, (const bf16_t*)G, (const bf16_t*)Z, *pro,
                     dW, B, H, W, OH, OW, PW);
  return 0;
}

static int g_stem_wtile_rows = 8;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_stem_wgrad_tile(int rows) { g_stem_wtile_rows = rows; return 0; }
#endif

int spb_stem_wgrad_mfma(const float* x, const void* G, const void* Z, const spb_bnref_t* pro, float* dW, int B, int H, int W,
                        hipStream_t s) {
  if (g_stem_tile && g_stem_wtile_rows > 0 && (W & 15) == 0 && (H & 1) == 0) {
    const int e = g_stem_wtile_rows >= 16 ? launch_wgrad_tile<16>(x, G, Z, pro, dW, B, H, W, s) : launch_wgrad_tile<8>(x, G, Z, pro, dW, B, H, W, s);
    if (e == 0) return 0;
  }
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long groups = ((long long)B * OH * OW + 31) / 32;
  long long grid = (groups + 3) / 4;
  if (grid > g_stem_grid_wgrad) grid = g_stem_grid_wgrad;
  hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, (const bf16_t*)G, (const bf16_t*)Z, *pro,
                     dW, B, H, W);
  return 0;
}
