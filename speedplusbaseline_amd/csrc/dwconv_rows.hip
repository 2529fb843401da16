// Depthwise 3x3 convolution, "row unit" kernels: forward and fused backward (input gradient + weight gradient).
// Same contract as dwconv.hip (reference park2019.py:47-49, torchvision MobileNetV2 inverted residual, park2019.py:107-108).
//
// Why a second formulation.  The tiled kernels in dwconv.hip stage an 8x8 window in LDS and read 9 taps x (32 B data +
// 32 B weights) per pixel and 8 channels from it.  Ablation on MI355X (scratch/ubench_dwbwd.hip, profiles/) showed that
// to be LDS-bandwidth bound: half of the kernel time is the tap loop at ~2x the 128 B/clk/CU LDS floor, and another
// 40 % is the staging + 3 barriers per tile.  Here nothing is staged:
//   * a UNIT is one 16-lane DPP row: 16 adjacent columns x one 8-channel group, marching down R image rows;
//   * each lane keeps its own column of the last three transformed rows in registers (sliding window); the left and
//     right taps come from the neighbouring lanes through DPP row shifts (no LDS, no barrier, no bank conflicts);
//   * 14 of the 16 lanes produce outputs (the outer two are the halo columns), which tiles every KRN feature-map
//     width exactly (112, 56, 28, 14 = k x 14); stride 2 maps lanes to the coarse (output / dz) columns, two fine
//     columns per lane, so that no tap is computed and masked away;
//   * weights and BN coefficients of the unit's 8 channels sit in 512 B of LDS and are read as broadcasts;
//   * global loads are 16 B per lane, 64 contiguous bytes per pixel across the 4 rows of a wave; the next image row is
//     fetched while the current one is computed (register prefetch, one step deep);
//   * per-channel BN sums and the 9 x 8 weight-gradient partials stay in registers for the whole unit, are reduced
//     over the 16 lanes with DPP butterflies and leave as one atomic per value.
// HBM traffic is the algorithmic minimum plus 2/R re-read halo rows (they hit L2).
#include "common.h"

namespace {

constexpr int CFN = 128;  // floats of per-unit coefficients in LDS: w[9][8] | p0,p1,p2 (bwd) or sc,sh (fwd) | sc,sh,mu,is

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void ld_lds8(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// value of the lane to the left / right inside the 16-lane row (0 past the row ends)
__device__ __forceinline__ float from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111 /*row_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101 /*row_shl:1*/, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ float row_sum(float v) {
  v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
  return v;
}

struct Unit { int c0, sx, sy, b; bool live; };
__device__ __forceinline__ Unit decode_unit(long long nunits, int ncg, int nstrip, int nseg) {
  long long u = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  Unit q;
  q.live = u < nunits;
  if (!q.live) u = nunits - 1;   // dead rows shadow the last unit (valid addresses), they never write
  q.c0 = (int)(u % ncg) * 8; u /= ncg;    // channel group fastest: the 4 rows of a wave read 64 contiguous bytes per pixel
  q.sx = (int)(u % nstrip); u /= nstrip;
  q.sy = (int)(u % nseg);
  q.b = (int)(u / nseg);
  return q;
}

// per-channel partial sums of one unit -> global accumulators (one atomic per value)
template <int NV>
__device__ __forceinline__ void push_rows(float* scratch /*unit's LDS*/, const float (&v)[NV][8], int l16, bool live, float* dst0,
                                          int stride_c, int stride_v) {
  // scratch[vi*8 + j] = sum over the row of v[vi][j];  then lanes share the atomics: dst0[j*stride_c + vi*stride_v]
#pragma unroll
  for (int vi = 0; vi < NV; ++vi)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = row_sum(v[vi][j]);
      if (l16 == ((vi * 8 + j) & 15)) scratch[vi * 8 + j] = s;
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (live) {
    for (int i = l16; i < NV * 8; i += 16) atomicAdd(dst0 + (size_t)(i & 7) * stride_c + (size_t)(i >> 3) * stride_v, scratch[i]);
  }
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------ forward
// ST = 1: lane column = output column = input column.     y[r][x] = sum_k w[k] * a[r-1+ky][x-1+kx]
// ST = 2: lane column = output column q, the lane owns input columns 2q, 2q+1; column 2q-1 is the left lane's 2q+1.
template <typename T, int ST>
__global__ __launch_bounds__(256, 2) void dwr_fwd_kernel(const spb_dw_args_t a, int R, int nseg, int nstrip, long long nunits) {
  __shared__ __attribute__((aligned(16))) float cfs[16][CFN];
  constexpr int NP = ST == 1 ? 14 : 15;
  const int l16 = threadIdx.x & 15;
  float* cf = cfs[threadIdx.x >> 4];
  const int C = a.C, H = a.H, W = a.W;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  const Unit u = decode_unit(nunits, C >> 3, nstrip, nseg);
  if (l16 < 8) {
    const int c = u.c0 + l16;
    float sc, sh;
    bn_fwd_coef(a.pro, c, sc, sh);
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k * 8 + l16] = a.Wd[(size_t)c * 9 + k];
    cf[72 + l16] = sc; cf[80 + l16] = sh;
  }
  __syncthreads();
  const int q = u.sx * NP + l16 - 1;                    // output column of this lane (lane 0 is the left halo)
  const bool prod = u.live && l16 >= 1 && l16 <= NP && q < OW;
  const int r0 = u.sy * R, r1 = min(r0 + R, OH);
  const T* X = reinterpret_cast<const T*>(a.X);
  T* Y = reinterpret_cast<T*>(a.Y);
  const int act = a.pro.act; const float slope = a.pro.slope;
  float s[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }

  if constexpr (ST == 1) {
    const bool colok = q >= 0 && q < W;
    const T* px = X + ((size_t)u.b * H * W + clampi(q, 0, W - 1)) * C + u.c0;
    const size_t rs = (size_t)W * C;
    auto ld = [&](int y) { return ldraw<T>(px + (size_t)clampi(y, 0, H - 1) * rs); };
    auto tf = [&](const Raw8<T>& r, int y, float o[8]) {
      float v[8], sc[8], sh[8];
      cvt8(r, v); ld_lds8(cf + 72, sc); ld_lds8(cf + 80, sh);
      const bool ok = colok && y >= 0 && y < H;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ok ? act_fwd(v[j] * sc[j] + sh[j], act, slope) : 0.f;
    };
    float xm[8], x0[8], xp[8];
    { const Raw8<T> ra = ld(r0 - 1), rb = ld(r0); tf(ra, r0 - 1, xm); tf(rb, r0, x0); }
    Raw8<T> nx = ld(r0 + 1);
    for (int r = r0; r < r1; ++r) {
      tf(nx, r + 1, xp);
      nx = ld(r + 2);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#define DWR_ROW(ky, XR)                                                                    \
      {                                                                                    \
        float w0[8], w1[8], w2[8];                                                         \
        ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);        \
        ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                      \
          acc[j] += from_left(XR[j]) * w0[j] + XR[j] * w1[j] + from_right(XR[j]) * w2[j];  \
      }
      DWR_ROW(0, xm) DWR_ROW(1, x0) DWR_ROW(2, xp)
#undef DWR_ROW
      rnd8<T>(acc);
      if (prod) {
        st8<T>(Y + ((size_t)(u.b * OH + r) * OW + q) * C + u.c0, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[0][j] += acc[j]; s[1][j] += acc[j] * acc[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { xm[j] = x0[j]; x0[j] = xp[j]; }
    }
  } else {
    const int ca = 2 * q, cb = 2 * q + 1;                // owned input columns
    const bool oka = ca >= 0 && ca < W, okb = cb >= 0 && cb < W;
    const T* pa = X + ((size_t)u.b * H * W + clampi(ca, 0, W - 1)) * C + u.c0;
    const T* pb = X + ((size_t)u.b * H * W + clampi(cb, 0, W - 1)) * C + u.c0;
    const size_t rs = (size_t)W * C;
    auto tf = [&](const Raw8<T>& r, bool cok, int y, float o[8]) {
      float v[8], sc[8], sh[8];
      cvt8(r, v); ld_lds8(cf + 72, sc); ld_lds8(cf + 80, sh);
      const bool ok = cok && y >= 0 && y < H;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ok ? act_fwd(v[j] * sc[j] + sh[j], act, slope) : 0.f;
    };
    float ma[8], mb[8];                                   // input row 2r-1
    {
      const int y = 2 * r0 - 1;
      const Raw8<T> ra = ldraw<T>(pa + (size_t)clampi(y, 0, H - 1) * rs), rb = ldraw<T>(pb + (size_t)clampi(y, 0, H - 1) * rs);
      tf(ra, oka, y, ma); tf(rb, okb, y, mb);
    }
    Raw8<T> n0a, n0b, n1a, n1b;
#define DWR_LD(r_)                                                                             \
    {                                                                                          \
      const size_t o0 = (size_t)clampi(2 * (r_), 0, H - 1) * rs, o1 = (size_t)clampi(2 * (r_) + 1, 0, H - 1) * rs; \
      n0a = ldraw<T>(pa + o0); n0b = ldraw<T>(pb + o0); n1a = ldraw<T>(pa + o1); n1b = ldraw<T>(pb + o1);        \
    }
    DWR_LD(r0)
    for (int r = r0; r < r1; ++r) {
      float ea[8], eb[8], oa[8], ob[8];                   // input rows 2r (e) and 2r+1 (o)
      tf(n0a, oka, 2 * r, ea); tf(n0b, okb, 2 * r, eb); tf(n1a, oka, 2 * r + 1, oa); tf(n1b, okb, 2 * r + 1, ob);
      DWR_LD(r + 1)
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#define DWR_ROW(ky, XA, XB)                                                                \
      {                                                                                    \
        float w0[8], w1[8], w2[8];                                                         \
        ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);        \
        ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                      \
          acc[j] += from_left(XB[j]) * w0[j] + XA[j] * w1[j] + XB[j] * w2[j];              \
      }
      DWR_ROW(0, ma, mb) DWR_ROW(1, ea, eb) DWR_ROW(2, oa, ob)
#undef DWR_ROW
      rnd8<T>(acc);
      if (prod) {
        st8<T>(Y + ((size_t)(u.b * OH + r) * OW + q) * C + u.c0, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[0][j] += acc[j]; s[1][j] += acc[j] * acc[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { ma[j] = oa[j]; mb[j] = ob[j]; }
    }
#undef DWR_LD
  }
  if (a.epi_mode == 1) {
    const long long uid = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    __syncthreads();  // all units of the block are done with their coefficients: cf becomes the reduction scratch
    push_rows<2>(cf, s, l16, u.live, a.osums + (size_t)(uid % a.oR) * 2 * C + u.c0, 1, C);
  }
}

// ------------------------------------------------------------------------------------------------------ backward
// dz[q] = g[q]*p0 + z[q]*p1 + p2 (BN backward of the conv output, rebuilt on the fly; 0 outside the image)
// dA[p] = sum_k dz[(p + 1 - k) / ST] * w[k]   (terms with non-integer index absent)
// dW[k] += dz[q] * a[p] for the same (p, k, q) triples, a = act(bn(z_in)) of the conv input       (WG)
// EPI: dA *= act'(u_in), + residual gradient, rounded, and its BN-backward sums  sum g, sum g*xhat.
// ST = 1: lane column = input column = dz column.  ST = 2: lane column = dz column q; the lane produces the input
// pixels (2r | 2r+1, 2q | 2q+1) of every dz row r -- exactly the 9 (tap, pixel) products, none masked away.
template <typename T, int ST, bool WG, bool EPI>
__global__ __launch_bounds__(256, 2) void dwr_bwd_kernel(const spb_dw_args_t a, int R, int nseg, int nstrip, long long nunits) {
  __shared__ __attribute__((aligned(16))) float cfs[16][CFN];
  constexpr int NP = ST == 1 ? 14 : 15;
  constexpr int LH = ST == 1 ? 1 : 0;                    // stride 2 only needs the right neighbour
  constexpr bool IN = WG || EPI;                         // the conv-input tensor is read
  const int l16 = threadIdx.x & 15;
  float* cf = cfs[threadIdx.x >> 4];
  const int C = a.C, H = a.H, W = a.W;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  const Unit u = decode_unit(nunits, C >> 3, nstrip, nseg);
  if (l16 < 8) {
    const int c = u.c0 + l16;
    float p0, p1, p2, sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
    bn_bwd_coef(a.pro, c, p0, p1, p2);
    if (IN && a.epi.gamma != nullptr) {
      bn_moments(a.epi, c, mu, is);
      sc = a.epi.gamma[c] * is;
      sh = a.epi.beta[c] - mu * sc;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k * 8 + l16] = a.Wd[(size_t)c * 9 + k];
    cf[72 + l16] = p0; cf[80 + l16] = p1; cf[88 + l16] = p2;
    cf[96 + l16] = sc; cf[104 + l16] = sh; cf[112 + l16] = mu; cf[120 + l16] = is;
  }
  __syncthreads();
  const int q = u.sx * NP + l16 - LH;                    // dz column of this lane
  const bool lane_prod = u.live && l16 >= LH && l16 < LH + NP;
  const bool qok = q >= 0 && q < OW;
  const int r0 = u.sy * R, r1 = min(r0 + R, OH);         // dz rows [r0, r1) -- for ST = 1 also the input rows
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Zo = reinterpret_cast<const T*>(a.Zout);
  const T* Rg = reinterpret_cast<const T*>(a.res);
  T* Y = reinterpret_cast<T*>(a.Y);
  const int eact = a.epi.act; const float eslope = a.epi.slope;
  const size_t grs = (size_t)OW * C;
  const size_t goff = ((size_t)u.b * OH * OW + clampi(q, 0, OW - 1)) * C + u.c0;
  const T* pg = G + goff;
  const T* pz = Z + goff;
  float s[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }
  float aw[WG ? 9 : 1][8];
#pragma unroll
  for (int k = 0; k < (WG ? 9 : 1); ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[k][j] = 0.f;

  auto dzf = [&](const Raw8<T>& g, const Raw8<T>& z, int y, float o[8]) {
    float gf[8], zf[8], p0[8], p1[8], p2[8];
    cvt8(g, gf); cvt8(z, zf); ld_lds8(cf + 72, p0); ld_lds8(cf + 80, p1); ld_lds8(cf + 88, p2);
    const bool ok = qok && y >= 0 && y < OH;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ok ? gf[j] * p0[j] + zf[j] * p1[j] + p2[j] : 0.f;
  };
  // a[p] of the conv input from its raw z (WG), 0 for lanes/pixels that do not exist
  auto apf = [&](const float zf[8], bool ok, float o[8]) {
    float sc[8], sh[8];
    ld_lds8(cf + 96, sc); ld_lds8(cf + 104, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ok ? act_fwd(zf[j] * sc[j] + sh[j], eact, eslope) : 0.f;
  };
  // finish one input pixel: residual, activation mask, rounding, BN-backward sums, store.  The second sum is kept as
  // sum g*z and turned into sum g*xhat = invstd * (sum g*z - mean * sum g) once per unit (two coefficient vectors less
  // in the loop; the subtraction costs log2(|mean|/std) bits of the f32 partial, a few at most).
  auto fin = [&](float acc[8], const float zf[8], const T* rp, bool ok, size_t off) {
    if (EPI) {
      float sc[8], sh[8];
      ld_lds8(cf + 96, sc); ld_lds8(cf + 104, sh);
      if (rp) {
        float rf[8];
        cvt8(ldraw<T>(rp + off), rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += rf[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = zf[j] * sc[j] + sh[j];
        acc[j] = rnd<T>(acc[j] * act_grad(uu, eact, eslope));
        if (ok) { s[0][j] += acc[j]; s[1][j] += acc[j] * zf[j]; }
      }
    }
    if (ok) st8<T>(Y + off, acc);
  };

  if constexpr (ST == 1) {
    const size_t ioff = ((size_t)u.b * H * W + clampi(q, 0, W - 1)) * C + u.c0;   // H == OH, W == OW
    const size_t irs = (size_t)W * C;
    float dm[8], d0[8], dp[8];
    {
      const size_t oa = (size_t)clampi(r0 - 1, 0, OH - 1) * grs, ob = (size_t)clampi(r0, 0, OH - 1) * grs;
      const Raw8<T> ga = ldraw<T>(pg + oa), za = ldraw<T>(pz + oa), gb = ldraw<T>(pg + ob), zb = ldraw<T>(pz + ob);
      dzf(ga, za, r0 - 1, dm); dzf(gb, zb, r0, d0);
    }
    Raw8<T> gn, zn, zon;
#define DWR_LD(y_)                                                                         \
    {                                                                                      \
      const size_t o_ = (size_t)clampi((y_) + 1, 0, OH - 1) * grs;                         \
      gn = ldraw<T>(pg + o_); zn = ldraw<T>(pz + o_);                                      \
      if (IN) {                                                                            \
        const size_t i_ = ioff + (size_t)clampi((y_), 0, H - 1) * irs;                     \
        zon = ldraw<T>(Zo + i_);                                                           \
      }                                                                                    \
    }
    DWR_LD(r0)
    for (int r = r0; r < r1; ++r) {
      // fused variant: keep the coefficient / weight reads inside the loop (hoisted they would cost 128 VGPRs on top
      // of the 72 weight-gradient accumulators)
      if (WG) asm volatile("" ::: "memory");
      dzf(gn, zn, r + 1, dp);
      float zf[8];
      if (IN) cvt8(zon, zf);
      DWR_LD(r + 1)
      const bool ok = lane_prod && qok;
      float acc[8], ap[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      if (WG) apf(zf, ok, ap);
      // taps: input pixel (r, q) <- dz(r + 1 - ky, q + 1 - kx)
#define DWR_ROW(ky, DR)                                                                    \
      {                                                                                    \
        if (WG) asm volatile("" ::: "memory"); /* one weight row in flight at a time */    \
        float w0[8], w1[8], w2[8];                                                         \
        ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);        \
        ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                    \
          const float vr = from_right(DR[j]), vl = from_left(DR[j]);                       \
          acc[j] += vr * w0[j] + DR[j] * w1[j] + vl * w2[j];                               \
          if (WG) {                                                                        \
            aw[WG ? (ky) * 3 + 0 : 0][j] += vr * ap[j];                                    \
            aw[WG ? (ky) * 3 + 1 : 0][j] += DR[j] * ap[j];                                 \
            aw[WG ? (ky) * 3 + 2 : 0][j] += vl * ap[j];                                    \
          }                                                                                \
        }                                                                                  \
      }
      DWR_ROW(0, dp) DWR_ROW(1, d0) DWR_ROW(2, dm)
#undef DWR_ROW
      fin(acc, zf, Rg, ok, ioff + (size_t)r * irs);
#pragma unroll
      for (int j = 0; j < 8; ++j) { dm[j] = d0[j]; d0[j] = dp[j]; }
    }
#undef DWR_LD
  } else {
    const int ca = 2 * q, cb = 2 * q + 1;                  // input columns produced by this lane
    const bool oka = lane_prod && ca >= 0 && ca < W, okb = lane_prod && cb >= 0 && cb < W;
    const size_t ia = ((size_t)u.b * H * W + clampi(ca, 0, W - 1)) * C + u.c0;
    const size_t ib = ((size_t)u.b * H * W + clampi(cb, 0, W - 1)) * C + u.c0;
    const size_t irs = (size_t)W * C;
    float d0[8], d1[8];                                    // dz rows r and r+1, own column
    {
      const size_t o_ = (size_t)clampi(r0, 0, OH - 1) * grs;
      const Raw8<T> g0 = ldraw<T>(pg + o_), z0 = ldraw<T>(pz + o_);
      dzf(g0, z0, r0, d0);
    }
    Raw8<T> gn, zn, zq[4];                                  // zq: conv-input z of (2r,ca) (2r,cb) (2r+1,ca) (2r+1,cb)
#define DWR_LD(r_)                                                                         \
    {                                                                                      \
      const size_t o_ = (size_t)clampi((r_) + 1, 0, OH - 1) * grs;                         \
      gn = ldraw<T>(pg + o_); zn = ldraw<T>(pz + o_);                                      \
      if (IN) {                                                                            \
        const size_t y0_ = (size_t)clampi(2 * (r_), 0, H - 1) * irs, y1_ = (size_t)clampi(2 * (r_) + 1, 0, H - 1) * irs; \
        zq[0] = ldraw<T>(Zo + ia + y0_); zq[1] = ldraw<T>(Zo + ib + y0_);                  \
        zq[2] = ldraw<T>(Zo + ia + y1_); zq[3] = ldraw<T>(Zo + ib + y1_);                  \
      }                                                                                    \
    }
    DWR_LD(r0)
    for (int r = r0; r < r1; ++r) {
      if (WG || EPI) asm volatile("" ::: "memory");
      dzf(gn, zn, r + 1, d1);
      Raw8<T> zc[4];
      if (IN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) zc[i] = zq[i];
      }
      DWR_LD(r + 1)
      const bool row1 = 2 * r + 1 < H;
      const size_t y0o = (size_t)(2 * r) * irs, y1o = (size_t)clampi(2 * r + 1, 0, H - 1) * irs;
      // one input pixel at a time (keeps acc/ap at 16 registers); residual gradient loaded where it is used
#define DWR_PIX(PI, OKP, OFF, BODY)                                                        \
      {                                                                                    \
        if (WG || EPI) asm volatile("" ::: "memory");                                      \
        const bool ok = (OKP);                                                             \
        float zf[8], ap[8], acc[8];                                                        \
        if (IN) cvt8(zc[PI], zf);                                                          \
        if (WG) apf(zf, ok, ap);                                                           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = 0.f;                        \
        BODY                                                                               \
        fin(acc, zf, Rg, ok, (OFF));                                                       \
      }
#define DWR_TAP(k, V)                                                                      \
      {                                                                                    \
        float w[8];                                                                        \
        ld_lds8(cf + (k) * 8, w);                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                    \
          const float v = (V);                                                             \
          acc[j] += v * w[j];                                                              \
          if (WG) aw[WG ? (k) : 0][j] += v * ap[j];                                        \
        }                                                                                  \
      }
      DWR_PIX(0, oka, ia + y0o, DWR_TAP(4, d0[j]))
      DWR_PIX(1, okb, ib + y0o, DWR_TAP(3, from_right(d0[j])) DWR_TAP(5, d0[j]))
      DWR_PIX(2, oka && row1, ia + y1o, DWR_TAP(1, d1[j]) DWR_TAP(7, d0[j]))
      DWR_PIX(3, okb && row1, ib + y1o,
              DWR_TAP(0, from_right(d1[j])) DWR_TAP(2, d1[j]) DWR_TAP(6, from_right(d0[j])) DWR_TAP(8, d0[j]))
#undef DWR_PIX
#undef DWR_TAP
#pragma unroll
      for (int j = 0; j < 8; ++j) d0[j] = d1[j];
    }
#undef DWR_LD
  }
  const long long uid = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (EPI) {
    float mu[8], is[8];
    ld_lds8(cf + 112, mu); ld_lds8(cf + 120, is);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[1][j] = is[j] * (s[1][j] - mu[j] * s[0][j]);
  }
  __syncthreads();  // coefficients no longer needed by any unit of the block: cf becomes the reduction scratch
  if (EPI) push_rows<2>(cf, s, l16, u.live, a.osums + (size_t)(uid % a.oR) * 2 * C + u.c0, 1, C);
  if constexpr (WG) push_rows<9>(cf, aw, l16, u.live, a.dW + (size_t)u.c0 * 9, 9, 1);
}

// rows per unit: as long as possible (2/R of the rows are re-read as halo) while the launch still has several rounds
// of units for every CU (256 CUs x 8 waves x 4 units resident)
int g_rows_override = 0;
struct RowGrid { int R, nseg, nstrip; long long nunits; };
RowGrid row_grid(int B, int C, int lane_rows, int lane_cols, int NP) {
  RowGrid g;
  g.nstrip = (lane_cols + NP - 1) / NP;
  const long long per_seg = (long long)B * (C >> 3) * g.nstrip;
  int R = lane_rows;
  while (R > 7 && per_seg * ((lane_rows + R - 1) / R) < 4 * 8192) R = (R + 1) / 2;
  if (g_rows_override > 0) R = g_rows_override < lane_rows ? g_rows_override : lane_rows;
  g.R = R; g.nseg = (lane_rows + R - 1) / R;
  g.nunits = per_seg * g.nseg;
  return g;
}

}  // namespace

extern "C" int spb_debug_set_dw_rows(int rows) { g_rows_override = rows; return 0; }

int spb_dwr_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const RowGrid g = row_grid(a->B, a->C, OH, OW, st == 1 ? 14 : 15);
  const dim3 grid((unsigned)((g.nunits + 15) / 16));
#define L_(T_, ST_) hipLaunchKernelGGL((dwr_fwd_kernel<T_, ST_>), grid, dim3(256), 0, s, *a, g.R, g.nseg, g.nstrip, g.nunits)
  if (dtype == SPB_BF16) { if (st == 1) L_(bf16_t, 1); else L_(bf16_t, 2); }
  else { if (st == 1) L_(float, 1); else L_(float, 2); }
#undef L_
  return 0;
}

int spb_dwr_bwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const RowGrid g = row_grid(a->B, a->C, OH, OW, st == 1 ? 14 : 15);
  const dim3 grid((unsigned)((g.nunits + 15) / 16));
  const bool wg = a->dW != nullptr, epi = a->epi_mode == 2;
#define L_(T_, ST_, WG_, EPI_) \
  hipLaunchKernelGGL((dwr_bwd_kernel<T_, ST_, WG_, EPI_>), grid, dim3(256), 0, s, *a, g.R, g.nseg, g.nstrip, g.nunits)
#define P_(T_, ST_)                                                              \
  {                                                                              \
    if (wg) { if (epi) L_(T_, ST_, true, true); else L_(T_, ST_, true, false); } \
    else { if (epi) L_(T_, ST_, false, true); else L_(T_, ST_, false, false); }  \
  }
  if (dtype == SPB_BF16) { if (st == 1) P_(bf16_t, 1) else P_(bf16_t, 2) }
  else { if (st == 1) P_(float, 1) else P_(float, 2) }
#undef P_
#undef L_
  return 0;
}
