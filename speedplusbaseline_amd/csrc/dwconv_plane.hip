// Depthwise 3x3 convolution on the SMALL feature maps (28x28, 14x14, 7x7): "plane" kernels, forward and backward
// (input gradient [+ weight gradient] + BatchNorm-backward sums).  Same contract as the row-unit kernels in
// dwconv_rows.hip (reference park2019.py:47-49 ConvDw, torchvision MobileNetV2 inverted residual, park2019.py:107-108).
//
// Why a third formulation.  On these maps a launch moves 5-30 MB.  The row-unit kernels -- a wave marching down the
// image rows behind a DMA ring -- take 12-30 us for it (round-3 trace: 14x14x384 backward 22.8 us = 1.3 TB/s; 7x7x960
// 23.7 us).  Phase timestamps (scratch/ubench_dwp.hip) showed what binds at this size: not memory latency but VALU
// ISSUE -- per element the nine taps are 9-18 FMAs, and everything around them (padding masks, index arithmetic, the
// cross-lane reduction of 72 weight-gradient partials per lane) was 4x that.  Here:
//   * a workgroup owns a TILE = (NB images) x (a segment of rows) x (all columns) x (32 channels = 64 bytes per pixel),
//     at most 256 output pixels; a thread owns up to 4 (pixel, 8-channel group) items, the group fixed per thread
//     (lane & 3), so the four lanes of a pixel read / write its 64 contiguous bytes;
//   * every global load of the thread (operands of its items, the BatchNorm sums and weights of its slab) is issued
//     before anything is waited for: one memory round trip per launch (measured: the 21.6 MB of a 14x14x384 backward
//     launch arrive in ~4 us = 5.4 TB/s);
//   * the only data that crosses threads -- the transformed conv input (forward) or dz = BN-backward(g, z) (backward)
//     -- goes through an f32 tile in LDS that is ZERO-PADDED by one pixel on every side (out-of-image rows included):
//     a tap is then a 2 x 16-byte LDS read at `item base + tap offset`, no mask, no compare; the two 16-byte halves of
//     an entry live in two planes so that consecutive lanes read consecutive 16 bytes (conflict-free ds_read_b128);
//   * pixel coordinates come from multiplications by host-computed reciprocals (no integer division in the kernel);
//   * per-channel sums leave by DPP rotations inside the 16-lane rows + v_permlane16/32_swap across them (vector ALU
//     only: ds_bpermute would put an LDS round trip into every step), LDS across the four waves, one f32 atomic per
//     value and workgroup;
//   * the weight gradient (72 partials per lane, each needing that cross-lane reduction) is an option (WG) of the
//     backward kernel; the KRN plan runs it on its side stream instead (row-unit instance), off the critical path;
//   * workgroups of one tile and different channel slabs sit on one XCD back to back (they share 128-byte lines).
#include "common.h"

// phase timestamps for scratch/ubench_dwp.hip (compiled out in the product build)
#ifndef SPB_PTS
#define SPB_PTS(i)
#endif

namespace {

constexpr int PCS = 32;          // channels per slab
constexpr int PNI = 4;           // output items per thread (256 pixels x 4 channel groups per workgroup)
constexpr int PMAX_OUT = 64 * PNI;
constexpr int PNI_DZ = 5;        // padded dz-tile entries per thread (backward): 320 pixels
constexpr int PMAX_DZ = 64 * PNI_DZ;

struct PGeo {
  int NB;      // images per workgroup
  int R;       // rows per segment (forward: output rows; backward: input rows)
  int nseg;    // segments per image
  int nimg;    // image groups
  int ntasks;  // nimg * nseg
  int nslab;   // channel slabs
  int TR, PW;  // padded tile: rows per image, columns (= map width + 2)
  unsigned m_tile, m_pw, m_out, m_w;   // reciprocals floor(2^32 / d) + 1 for d = TR*PW, PW, R*OWt, OWt (OWt = columns of the output grid)
};

__device__ __forceinline__ int mdiv(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }   // n / d, 0 <= n < 65536

__device__ __forceinline__ float ror4(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
}
__device__ __forceinline__ float ror8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
}
// sum over the 16 lanes of the wave with the same (lane & 3), result in every lane
__device__ __forceinline__ float class_sum(float v) {
  v += ror4(v); v += ror8(v);
  return xor32_sum(xor16_sum(v));
}
__device__ __forceinline__ float4 ldsv(const float* p) { return *reinterpret_cast<const float4*>(p); }

// block id -> (slab, task): blocks b, b+8, b+16, ... run on one XCD; consecutive ones take the slabs of one task
__device__ __forceinline__ bool plane_task(const PGeo& g, int& slab, int& task) {
  const int j = blockIdx.x >> 3, x = blockIdx.x & 7;
  slab = j % g.nslab; task = (j / g.nslab) * 8 + x;
  return task < g.ntasks;
}

// weight staging shared by both kernels: thread t fetches weights i1 = t and i2 = t + 256 of the slab's 9 x 32
struct WPair { float w1, w2; int k1, ch1, k2, ch2; };
__device__ __forceinline__ WPair load_wpair(const float* Wd, int c0, int C, int t) {
  WPair p;
  const int i1 = t, i2 = t + 256 < 9 * PCS ? t + 256 : 9 * PCS - 1;
  p.ch1 = mdiv(i1, 477218589u); p.k1 = i1 - p.ch1 * 9;        // / 9
  p.ch2 = mdiv(i2, 477218589u); p.k2 = i2 - p.ch2 * 9;
  p.w1 = Wd[(size_t)min(c0 + p.ch1, C - 1) * 9 + p.k1];
  p.w2 = Wd[(size_t)min(c0 + p.ch2, C - 1) * 9 + p.k2];
  return p;
}
__device__ __forceinline__ void store_wpair(float* wl, const WPair& p, int c0, int C, int t) {
  wl[p.k1 * PCS + p.ch1] = c0 + p.ch1 < C ? p.w1 : 0.f;
  if (t + 256 < 9 * PCS) wl[p.k2 * PCS + p.ch2] = c0 + p.ch2 < C ? p.w2 : 0.f;
}

// ------------------------------------------------------------------------------------------------------ forward
// y[oy][ox] = sum_k w[k] * a[oy*ST-1+ky][ox*ST-1+kx],  a = act(bn(x)), zero outside the image; + sum(y), sum(y^2)
// padded tile: row 0 = input row ro0*ST - 1, TR = (R-1)*ST + 3 rows per image, column 0 = input column -1
template <typename T, int ST, int NI_IN>
__global__ __launch_bounds__(256) void dwp_fwd_kernel(const spb_dw_args_t a, const PGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cf = reinterpret_cast<float*>(smem);   // [2][32]: scale, shift
  float* wl = cf + 64;                           // [9][32]
  float* red = wl + 288;                         // [4 waves][4 groups][16]
  float* tile = red + 256;                       // two planes of [entries][4] f32
  int slab, task;
  if (!plane_task(g, slab, task)) return;
  const int t = threadIdx.x, cg = t & 3, wave = t >> 6, lane = t & 63;
  const int C = a.C, H = a.H, W = a.W;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  const int ig = task / g.nseg, seg = task - ig * g.nseg;
  const int b0 = ig * g.NB, nb = min(g.NB, a.B - b0);
  const int ro0 = seg * g.R;
  const int iy0 = ro0 * ST - 1;                  // input row of padded tile row 0
  const int TR = g.TR, PW = g.PW;
  const int ntile = nb * TR * PW, nout = nb * g.R * OW;
  const int plane = g.NB * TR * PW * 16;         // floats per plane
  const int c0 = slab * PCS, c8 = c0 + cg * 8;
  const bool cgok = c8 < C;
  const int c8c = cgok ? c8 : C - 8;
  const T* X = reinterpret_cast<const T*>(a.X);
  T* Y = reinterpret_cast<T*>(a.Y);

  // ---- every global load first: tile entries e = t + 256 i  <->  (pixel = e >> 2, group = cg)
  Raw8<T> raw[NI_IN];
  bool inimg[NI_IN];
#pragma unroll
  for (int i = 0; i < NI_IN; ++i) {
    const int px = (t + 256 * i) >> 2;
    const int img = mdiv(px, g.m_tile), rem = px - img * (TR * PW), tr = mdiv(rem, g.m_pw), tc = rem - tr * PW;
    const int iy = iy0 + tr, ix = tc - 1;
    inimg[i] = px < ntile && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int bc = min(b0 + img, a.B - 1), yc = min(max(iy, 0), H - 1), xc = min(max(ix, 0), W - 1);
    raw[i] = ldraw<T>(X + (((size_t)bc * H + yc) * W + xc) * C + c8c);
  }
  {
    const int ch = t & (PCS - 1), c = c0 + ch, cc = c < C ? c : C - 1;
    const WPair wp = load_wpair(a.Wd, c0, C, t);
    float sc, sh;
    bn_fwd_coef(a.pro, cc, sc, sh);
    if (t < PCS) { cf[t] = sc; cf[PCS + t] = sh; }
    store_wpair(wl, wp, c0, C, t);
  }
  __syncthreads();
  // ---- transformed input (or zero padding) -> LDS
  {
    const float4 sca = ldsv(cf + cg * 8), scb = ldsv(cf + cg * 8 + 4), sha = ldsv(cf + PCS + cg * 8), shb = ldsv(cf + PCS + cg * 8 + 4);
    const float sc[8] = {sca.x, sca.y, sca.z, sca.w, scb.x, scb.y, scb.z, scb.w};
    const float sh[8] = {sha.x, sha.y, sha.z, sha.w, shb.x, shb.y, shb.z, shb.w};
    const int act = a.pro.act; const float slope = a.pro.slope;
#pragma unroll
    for (int i = 0; i < NI_IN; ++i) {
      const int e = t + 256 * i;
      if ((e >> 2) < ntile) {
        float v[8];
        cvt8(raw[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = inimg[i] ? act_fwd(v[j] * sc[j] + sh[j], act, slope) : 0.f;
        *reinterpret_cast<float4*>(tile + (size_t)e * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(tile + plane + (size_t)e * 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
  __syncthreads();
  // ---- taps
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  float acc[PNI][8];
  int base[PNI];
#pragma unroll
  for (int i = 0; i < PNI; ++i) {
    const int px = (t + 256 * i) >> 2;
    const int img = mdiv(px, g.m_out), rem = px - img * (g.R * OW), ry = mdiv(rem, g.m_w), ox = rem - ry * OW;
    base[i] = (((img * TR + ry * ST) * PW + ox * ST) * 4 + cg) * 4;      // float offset of tap (0, 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  }
  // (rolled: unrolled, the scheduler hoists all 72 tile reads -- 350 registers, one wave per SIMD)
#pragma unroll 1
  for (int k = 0; k < 9; ++k) {
      const int ky = k >= 6 ? 2 : (k >= 3 ? 1 : 0), kx = k - 3 * ky;
      const float4 wa = ldsv(wl + k * PCS + cg * 8), wb = ldsv(wl + k * PCS + cg * 8 + 4);
      const int off = (ky * PW + kx) * 16;
#pragma unroll
      for (int i = 0; i < PNI; ++i) {
        const float4 va = ldsv(tile + base[i] + off), vb = ldsv(tile + plane + base[i] + off);
        acc[i][0] += va.x * wa.x; acc[i][1] += va.y * wa.y; acc[i][2] += va.z * wa.z; acc[i][3] += va.w * wa.w;
        acc[i][4] += vb.x * wb.x; acc[i][5] += vb.y * wb.y; acc[i][6] += vb.z * wb.z; acc[i][7] += vb.w * wb.w;
      }
    }
#pragma unroll
  for (int i = 0; i < PNI; ++i) {
    const int px = (t + 256 * i) >> 2;
    const int img = mdiv(px, g.m_out), rem = px - img * (g.R * OW), ry = mdiv(rem, g.m_w), ox = rem - ry * OW, oy = ro0 + ry;
    rnd8<T>(acc[i]);
    if (px < nout && oy < OH && cgok) {
      st8<T>(Y + (((size_t)(b0 + img) * OH + oy) * OW + ox) * C + c8, acc[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += acc[i][j]; s2[j] += acc[i][j] * acc[i][j]; }
    }
  }
  // ---- batch sums of the output
  if (a.epi_mode == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = class_sum(s1[j]), v = class_sum(s2[j]);
      if (lane < 4) { red[(wave * 4 + lane) * 16 + j] = u; red[(wave * 4 + lane) * 16 + 8 + j] = v; }
    }
    __syncthreads();
    if (t < 64) {
      const int gi = t >> 4, idx = t & 15, c = c0 + gi * 8 + (idx & 7);
      const float s = red[(0 * 4 + gi) * 16 + idx] + red[(1 * 4 + gi) * 16 + idx] + red[(2 * 4 + gi) * 16 + idx] + red[(3 * 4 + gi) * 16 + idx];
      if (c < C) atomicAdd(a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C + (size_t)(idx >> 3) * C + c, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ backward
// dz[q] = g[q]*p0 + z[q]*p1 + p2 (BN backward of the conv output, 0 outside the image)
// dA[p] = sum_k dz[(p + 1 - k) / ST] * w[k]    (terms with a non-integer index absent)
// dW[k] += dz[(p + 1 - k) / ST] * a[p],  a = act(bn(z_in)) of the conv input                                  (WG)
// EPI: g_in = (dA + res) * act'(bn(z_in)), rounded; sum g_in, sum g_in * xhat_in
// padded dz tile: row 0 = dz row q_first = (ST == 1 ? r0 - 1 : (r0 - 1) >> 1), TR rows per image, column 0 = dz column -1
template <typename T, int ST, bool WG, bool EPI>
__global__ __launch_bounds__(256) void dwp_bwd_kernel(const spb_dw_args_t a, const PGeo g) {
  spb_publish_entry(a.entry_flag, a.entry_val);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool IN = WG || EPI;
  constexpr int NRED = 16 + (WG ? 72 : 0);
  float* cf = reinterpret_cast<float*>(smem);   // [7][32]: p0 p1 p2 | sc sh mu is
  float* wl = cf + 7 * PCS;                     // [9][32]
  float* red = wl + 9 * PCS;                    // [4 waves][4 groups][NRED]
  float* tile = red + 16 * NRED;                // two planes of [entries][4] f32
  int slab, task;
  if (!plane_task(g, slab, task)) return;
  SPB_PTS(0);
  const int t = threadIdx.x, cg = t & 3, wave = t >> 6, lane = t & 63;
  const int C = a.C, H = a.H, W = a.W;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  const int ig = task / g.nseg, seg = task - ig * g.nseg;
  const int b0 = ig * g.NB, nb = min(g.NB, a.B - b0);
  const int r0 = seg * g.R;                                            // first input row produced here
  const int qf = ST == 1 ? r0 - 1 : (r0 - 1) >> 1;                     // dz row of padded tile row 0 (may be -1)
  const int TR = g.TR, PW = g.PW;
  const int ntile = nb * TR * PW, nout = nb * g.R * W;
  const int plane = g.NB * TR * PW * 16;
  const int c0 = slab * PCS, c8 = c0 + cg * 8;
  const bool cgok = c8 < C;
  const int c8c = cgok ? c8 : C - 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Zo = reinterpret_cast<const T*>(a.Zout);
  const T* Rg = reinterpret_cast<const T*>(a.res);
  T* Y = reinterpret_cast<T*>(a.Y);

  // ---- every global load first
  Raw8<T> gr[PNI_DZ], zr[PNI_DZ], zi[IN ? PNI : 1];
  bool inimg[PNI_DZ];
#pragma unroll
  for (int i = 0; i < PNI_DZ; ++i) {
    const int px = (t + 256 * i) >> 2;
    const int img = mdiv(px, g.m_tile), rem = px - img * (TR * PW), tr = mdiv(rem, g.m_pw), tc = rem - tr * PW;
    const int qy = qf + tr, qx = tc - 1;
    inimg[i] = px < ntile && qy >= 0 && qy < OH && qx >= 0 && qx < OW;
    const int bc = min(b0 + img, a.B - 1), yc = min(max(qy, 0), OH - 1), xc = min(max(qx, 0), OW - 1);
    const size_t o = (((size_t)bc * OH + yc) * OW + xc) * C + c8c;
    gr[i] = ldraw<T>(G + o); zr[i] = ldraw<T>(Z + o);
  }
  int base[PNI], oy[ST == 2 ? PNI : 1], ox[ST == 2 ? PNI : 1];
  size_t ooff[PNI];
  bool okp[PNI];
#pragma unroll
  for (int i = 0; i < PNI; ++i) {
    const int px = (t + 256 * i) >> 2;
    const int img = mdiv(px, g.m_out), rem = px - img * (g.R * W), ry = mdiv(rem, g.m_w), x = rem - ry * W, y = r0 + ry;
    okp[i] = px < nout && y < H && cgok;
    const int bc = min(b0 + img, a.B - 1), yc = min(y, H - 1);
    ooff[i] = (((size_t)bc * H + yc) * W + x) * C + c8c;
    if (IN) zi[IN ? i : 0] = ldraw<T>(Zo + ooff[i]);
    // ST 1: float offset of tap (0, 0) = entry of dz(y + 1, x + 1).  ST 2: entry index of dz(0, 0) of the image; the
    // tap adds ((y + 1 - ky) >> 1) * PW + ((x + 1 - kx) >> 1)
    if (ST == 1) base[i] = (((img * TR + (y + 1 - qf)) * PW + x + 2) * 4 + cg) * 4;
    else { base[i] = (img * TR - qf) * PW + 1; oy[ST == 2 ? i : 0] = y; ox[ST == 2 ? i : 0] = x; }
  }
  {
    const int ch = t & (PCS - 1), c = c0 + ch, cc = c < C ? c : C - 1;
    const WPair wp = load_wpair(a.Wd, c0, C, t);
    float p0, p1, p2, sc, sh, mu, is;
    bn_bwd_epi_coef(a.pro, a.epi, IN && a.epi.gamma != nullptr, cc, p0, p1, p2, sc, sh, mu, is);
    if (t < PCS) {
      cf[t] = p0; cf[PCS + t] = p1; cf[2 * PCS + t] = p2;
      cf[3 * PCS + t] = sc; cf[4 * PCS + t] = sh; cf[5 * PCS + t] = mu; cf[6 * PCS + t] = is;
    }
    store_wpair(wl, wp, c0, C, t);
  }
  SPB_PTS(1);
  __syncthreads();
  SPB_PTS(2);
  // ---- dz (or zero padding) -> LDS
  {
    const float4 a0 = ldsv(cf + cg * 8), b0_ = ldsv(cf + cg * 8 + 4), a1 = ldsv(cf + PCS + cg * 8), b1 = ldsv(cf + PCS + cg * 8 + 4);
    const float4 a2 = ldsv(cf + 2 * PCS + cg * 8), b2 = ldsv(cf + 2 * PCS + cg * 8 + 4);
    const float p0[8] = {a0.x, a0.y, a0.z, a0.w, b0_.x, b0_.y, b0_.z, b0_.w};
    const float p1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
    const float p2[8] = {a2.x, a2.y, a2.z, a2.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int i = 0; i < PNI_DZ; ++i) {
      const int e = t + 256 * i;
      if ((e >> 2) < ntile) {
        float gf[8], zf[8];
        cvt8(gr[i], gf); cvt8(zr[i], zf);
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[j] = inimg[i] ? gf[j] * p0[j] + zf[j] * p1[j] + p2[j] : 0.f;
        *reinterpret_cast<float4*>(tile + (size_t)e * 4) = make_float4(gf[0], gf[1], gf[2], gf[3]);
        *reinterpret_cast<float4*>(tile + plane + (size_t)e * 4) = make_float4(gf[4], gf[5], gf[6], gf[7]);
      }
    }
  }
  __syncthreads();
  SPB_PTS(3);
  // ---- taps
  const int eact = a.epi.act; const float eslope = a.epi.slope;
  float acc[PNI][8], ap[WG ? PNI : 1][8];
  float sc[8], sh[8];
  {
    const float4 sa = ldsv(cf + 3 * PCS + cg * 8), sb = ldsv(cf + 3 * PCS + cg * 8 + 4), ha = ldsv(cf + 4 * PCS + cg * 8), hb = ldsv(cf + 4 * PCS + cg * 8 + 4);
    sc[0] = sa.x; sc[1] = sa.y; sc[2] = sa.z; sc[3] = sa.w; sc[4] = sb.x; sc[5] = sb.y; sc[6] = sb.z; sc[7] = sb.w;
    sh[0] = ha.x; sh[1] = ha.y; sh[2] = ha.z; sh[3] = ha.w; sh[4] = hb.x; sh[5] = hb.y; sh[6] = hb.z; sh[7] = hb.w;
  }
#pragma unroll
  for (int i = 0; i < PNI; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    if (WG) {
      float zf[8];
      cvt8(zi[IN ? i : 0], zf);
#pragma unroll
      for (int j = 0; j < 8; ++j) ap[WG ? i : 0][j] = okp[i] ? act_fwd(zf[j] * sc[j] + sh[j], eact, eslope) : 0.f;
    }
  }
  // one tap: acc[i] += dz * w[k] for the thread's items (+ the tap's weight-gradient partial)
  auto tap = [&](int k, int ky, int kx) {
    const float4 wa = ldsv(wl + k * PCS + cg * 8), wb = ldsv(wl + k * PCS + cg * 8 + 4);
    float aw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[j] = 0.f;
    const int off = -(ky * PW + kx) * 16;
#pragma unroll
    for (int i = 0; i < PNI; ++i) {
      float4 va, vb;
      if (ST == 1) {
        va = ldsv(tile + base[i] + off); vb = ldsv(tile + plane + base[i] + off);
      } else {
        const int ty = oy[ST == 2 ? i : 0] + 1 - ky, tx = ox[ST == 2 ? i : 0] + 1 - kx;
        const int e = ((base[i] + (ty >> 1) * PW + (tx >> 1)) * 4 + cg) * 4;
        va = ldsv(tile + e); vb = ldsv(tile + plane + e);
        if ((ty | tx) & 1) { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va; }
      }
      acc[i][0] += va.x * wa.x; acc[i][1] += va.y * wa.y; acc[i][2] += va.z * wa.z; acc[i][3] += va.w * wa.w;
      acc[i][4] += vb.x * wb.x; acc[i][5] += vb.y * wb.y; acc[i][6] += vb.z * wb.z; acc[i][7] += vb.w * wb.w;
      if (WG) {
        const float* p = ap[WG ? i : 0];
        aw[0] += va.x * p[0]; aw[1] += va.y * p[1]; aw[2] += va.z * p[2]; aw[3] += va.w * p[3];
        aw[4] += vb.x * p[4]; aw[5] += vb.y * p[5]; aw[6] += vb.z * p[6]; aw[7] += vb.w * p[7];
      }
    }
    if (WG) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s = class_sum(aw[j]);
        if (lane < 4) red[(wave * 4 + lane) * NRED + 16 + k * 8 + j] = s;
      }
    }
  };
  // the tap loop stays rolled: unrolled, the scheduler hoists all 72 tile reads of the nine taps (350-460 registers, one wave per SIMD)
#pragma unroll 1
  for (int k = 0; k < 9; ++k) { const int ky = k >= 6 ? 2 : (k >= 3 ? 1 : 0); tap(k, ky, k - 3 * ky); }
  SPB_PTS(4);
  // ---- finish the input pixels
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < PNI; ++i) {
    if (EPI) {
      float zf[8];
      cvt8(zi[IN ? i : 0], zf);
      if (Rg) {   // residual gradient (KRN: only the DANN feature tap reaches a plane-sized layer): loaded here, not held over the taps
        float rf[8];
        cvt8(ldraw<T>(Rg + ooff[i]), rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += rf[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = zf[j] * sc[j] + sh[j];
        acc[i][j] = rnd<T>(acc[i][j] * act_grad(uu, eact, eslope));
        if (okp[i]) { s1[j] += acc[i][j]; s2[j] += acc[i][j] * zf[j]; }
      }
    }
    if (okp[i]) st8<T>(Y + ooff[i], acc[i]);
  }
  if (EPI) {
    const float4 ma = ldsv(cf + 5 * PCS + cg * 8), mb = ldsv(cf + 5 * PCS + cg * 8 + 4), ia = ldsv(cf + 6 * PCS + cg * 8), ib = ldsv(cf + 6 * PCS + cg * 8 + 4);
    const float mu[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
    const float is[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = class_sum(s1[j]), v = class_sum(s2[j]);
      if (lane < 4) { red[(wave * 4 + lane) * NRED + j] = u; red[(wave * 4 + lane) * NRED + 8 + j] = is[j] * (v - mu[j] * u); }
    }
  }
  SPB_PTS(5);
  if (EPI || WG) {
    __syncthreads();
    for (int i = t; i < 4 * NRED; i += 256) {
      const int gi = i / NRED, idx = i - gi * NRED;
      const float s = red[(0 * 4 + gi) * NRED + idx] + red[(1 * 4 + gi) * NRED + idx] + red[(2 * 4 + gi) * NRED + idx] +
                      red[(3 * 4 + gi) * NRED + idx];
      if (idx < 16) {
        const int c = c0 + gi * 8 + (idx & 7);
        if (EPI && c < C) atomicAdd(a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C + (size_t)(idx >> 3) * C + c, s);
      } else if (WG) {
        const int k = (idx - 16) >> 3, c = c0 + gi * 8 + ((idx - 16) & 7);
        if (c < C) atomicAdd(a.dW + (size_t)c * 9 + k, s);
      }
    }
  }
  SPB_PTS(6);
}

// ---- tile planning ---------------------------------------------------------------------------------------------------
int g_plane_min_wgs = 384;   // fewer workgroups than this: smaller image groups
int g_plane_max_w = 14;      // widest map the plane kernels take (measured at bs=48: 28x28 is no faster than the row-unit kernels)

unsigned recip(int d) { return (unsigned)(0x100000000ull / (unsigned)d) + 1u; }

// image groups: as many images per workgroup as the item capacities allow, fewer when the launch would not fill the chip
void finish_geo(PGeo& g, int B, int C, int out_px, int tile_px, int tile_cap, int out_cols) {
  g.nslab = (C + PCS - 1) / PCS;
  int nb = 1;
  if (g.nseg == 1) {
    nb = PMAX_OUT / out_px;
    if (tile_cap / tile_px < nb) nb = tile_cap / tile_px;
    if (nb > 4) nb = 4;
    if (nb > B) nb = B;
    if (nb < 1) nb = 1;
    while (nb > 1 && (long long)g.nslab * ((B + nb - 1) / nb) < g_plane_min_wgs) --nb;
  }
  g.NB = nb;
  g.nimg = (B + g.NB - 1) / g.NB;
  g.ntasks = g.nimg * g.nseg;
  g.m_tile = recip(g.TR * g.PW); g.m_pw = recip(g.PW); g.m_out = recip(g.R * out_cols); g.m_w = recip(out_cols);
}

// backward: tile = input rows [r0, r0 + R); padded dz rows derived.  false: shape not covered (the row-unit kernel takes it)
bool plan_bwd(int B, int H, int W, int C, int st, PGeo& g) {
  if (W > g_plane_max_w || W > 28 || H > 64) return false;
  const int OW = (W - 1) / st + 1;
  g.PW = OW + 2;
  for (int nseg = 1;; ++nseg) {
    const int R = (H + nseg - 1) / nseg;
    // dz rows of a segment: ST 1: r0-1 .. r0+R;  ST 2: (r0-1)>>1 .. (r0+R)>>1, at most R/2 + 2 rows
    const int TR = st == 1 ? R + 2 : R / 2 + 2;
    if (R * W <= PMAX_OUT && TR * g.PW <= PMAX_DZ) { g.R = R; g.TR = TR; break; }
    if (R == 1) return false;
  }
  g.nseg = (H + g.R - 1) / g.R;
  finish_geo(g, B, C, g.R * W, g.TR * g.PW, PMAX_DZ, W);
  return true;
}
// forward: tile = output rows [ro0, ro0 + R); padded input rows derived (at most tile_cap entries)
bool plan_fwd(int B, int H, int W, int C, int st, int tile_cap, PGeo& g) {
  if (W > g_plane_max_w || W > 28 || H > 64) return false;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  g.PW = W + 2;
  for (int nseg = 1;; ++nseg) {
    const int R = (OH + nseg - 1) / nseg;
    const int TR = (R - 1) * st + 3;
    if (R * OW <= PMAX_OUT && TR * g.PW <= tile_cap) { g.R = R; g.TR = TR; break; }
    if (R == 1) return false;
  }
  g.nseg = (OH + g.R - 1) / g.R;
  finish_geo(g, B, C, g.R * OW, g.TR * g.PW, tile_cap, OW);
  return true;
}

template <typename K>
void plane_allow_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
unsigned plane_grid(const PGeo& g) { return (unsigned)(g.nslab * ((g.ntasks + 7) / 8) * 8); }

}  // namespace

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_plane_min_wgs(int n) { g_plane_min_wgs = n; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_plane_max_w(int w) { g_plane_max_w = w; return 0; }
#endif

// returns 0 when launched, SPB_E_UNSUPPORTED when the shape is left to the row-unit kernels
int spb_dwp_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  PGeo g;
  const int tile_cap = st == 1 ? 320 : 512;
  if (!plan_fwd(a->B, a->H, a->W, a->C, st, tile_cap, g)) return SPB_E_UNSUPPORTED;
  const size_t lds = (size_t)(64 + 288 + 256) * sizeof(float) + (size_t)g.NB * g.TR * g.PW * PCS * sizeof(float);
#define F_(T_, ST_, NI_)                                                                                   \
  {                                                                                                        \
    static bool once = false;                                                                              \
    if (!once) { plane_allow_lds(dwp_fwd_kernel<T_, ST_, NI_>); once = true; }                             \
    hipLaunchKernelGGL((dwp_fwd_kernel<T_, ST_, NI_>), dim3(plane_grid(g)), dim3(256), lds, s, *a, g);     \
  }
  if (dtype == SPB_BF16) { if (st == 1) F_(bf16_t, 1, 5) else F_(bf16_t, 2, 8) }
  else { if (st == 1) F_(float, 1, 5) else F_(float, 2, 8) }
#undef F_
  return 0;
}

int spb_dwp_bwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  PGeo g;
  if (!plan_bwd(a->B, a->H, a->W, a->C, st, g)) return SPB_E_UNSUPPORTED;
  const bool wg = a->dW != nullptr, epi = a->epi_mode == 2;
  const size_t lds = (size_t)(7 * PCS + 9 * PCS + 16 * (16 + (wg ? 72 : 0))) * sizeof(float) + (size_t)g.NB * g.TR * g.PW * PCS * sizeof(float);
#define L_(T_, ST_, WG_, EPI_)                                                                               \
  {                                                                                                          \
    static bool once = false;                                                                                \
    if (!once) { plane_allow_lds(dwp_bwd_kernel<T_, ST_, WG_, EPI_>); once = true; }                         \
    hipLaunchKernelGGL((dwp_bwd_kernel<T_, ST_, WG_, EPI_>), dim3(plane_grid(g)), dim3(256), lds, s, *a, g); \
  }
#define P_(T_, ST_)                                                              \
  {                                                                              \
    if (wg) { if (epi) L_(T_, ST_, true, true) else L_(T_, ST_, true, false) }   \
    else { if (epi) L_(T_, ST_, false, true) else L_(T_, ST_, false, false) }    \
  }
  if (dtype == SPB_BF16) { if (st == 1) P_(bf16_t, 1) else P_(bf16_t, 2) }
  else { if (st == 1) P_(float, 1) else P_(float, 2) }
#undef P_
#undef L_
  return 0;
}
