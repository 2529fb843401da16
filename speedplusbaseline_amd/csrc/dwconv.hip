// Depthwise 3x3 convolution (pad 1, stride 1|2), NHWC, forward / input-gradient / weight-gradient.
// Replaces nn.Conv2d(C,C,3,groups=C) + BN + ReLU(6) at reference park2019.py:47-49 and in the torchvision
// MobileNetV2 inverted-residual blocks (park2019.py:107-108).
//
// 9 MACs per element: no matrix-core work, these kernels are HBM/L2 streaming.  A workgroup owns a 64-channel slab
// (8 lanes x 16-byte vectors = one 128-byte line per pixel) and is PERSISTENT over spatial tiles:
//   * the input window of a tile (with halo) is loaded once, branch-free, all loads in flight together, and transformed
//     on the fly -- the previous layer's BN+activation in forward, the BN-backward reconstruction of dz in the
//     gradient kernels -- into an f32 LDS tile; the 9 taps of every output then come from LDS;
//   * the raw loads of tile t+1 are issued before tile t is computed (register prefetch), so HBM latency overlaps the
//     LDS phase (measured before this: 6 us of exposed load latency per 2 us of compute);
//   * per-channel BN sums / weight gradients stay in registers across the tile loop, are reduced with wave shuffles
//     and leave the workgroup as one atomic per channel.
// The normalised tensors never exist in HBM and each element is read from HBM/L2 once per kernel.
#include "common.h"

// phase timestamps for scratch/ubench_dw.hip (compiled out in the product build)
#ifndef SPB_TS
#define SPB_TS(i)
#endif

namespace {

constexpr int PADC = 72;    // floats per pixel in the LDS tile: 64 channels + 8 pad (32-byte skew across pixels)
constexpr int NIT = 5;      // vec8 tile loads per thread (9x17 pixels x 8 channel groups / 256 threads, rounded up)

struct DwThread { int cg, cgl, pl, npl, cgl_n; bool valid; };

__device__ __forceinline__ DwThread dw_thread(int C) {
  DwThread d;
  const int CG = C >> 3;
  d.cgl_n = CG < 8 ? 4 : 8;
  const int t = threadIdx.x;
  d.cgl = t % d.cgl_n;
  d.pl = t / d.cgl_n;
  d.npl = 256 / d.cgl_n;
  d.cg = blockIdx.y * d.cgl_n + d.cgl;
  d.valid = d.cg < CG;
  return d;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Which (row, column, channel group) of a th x tw window each of this thread's NIT load slots covers.  The same for
// every tile, so the divisions happen once per kernel.
struct TileMap {
  short dy[NIT], dx[NIT];
  int lds[NIT];      // float offset of the slot in the LDS tile, -1 if the slot is past the window
  int coff;          // channel offset of this thread's slots in the global tensor (clamped to a valid group)
  bool cok;          // channel group exists
};
__device__ __forceinline__ TileMap make_map(int th, int tw, const DwThread& d, int C) {
  TileMap m;
  const int items = th * tw * d.cgl_n;
  const int cl = threadIdx.x % d.cgl_n;  // 256 % cgl_n == 0: every slot of a thread has the same channel group
  const int cg = blockIdx.y * d.cgl_n + cl;
  const int CG = C >> 3;
  m.cok = cg < CG;
  m.coff = (cg < CG ? cg : CG - 1) * 8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = threadIdx.x + 256 * it;
    const int ic = i < items ? i : items - 1;
    const int pix = ic / d.cgl_n;
    m.dy[it] = (short)(pix / tw); m.dx[it] = (short)(pix % tw);
    m.lds[it] = i < items ? pix * PADC + cl * 8 : -1;
  }
  return m;
}

// issue the raw loads of one window (origin y0,x0 of image b; clamped addresses, no branches)
template <typename T, int MODE>
__device__ __forceinline__ void tile_issue(Raw8<T> r1[NIT], Raw8<T> r2[NIT], const TileMap& m, const T* X, const T* X2, int b,
                                           int y0, int x0, int H, int W, int C) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int y = clampi(y0 + m.dy[it], 0, H - 1), x = clampi(x0 + m.dx[it], 0, W - 1);
    const size_t o = ((size_t)(b * H + y) * W + x) * C + m.coff;
    r1[it] = ldraw<T>(X + o);
    if (MODE == 1) { if (X2) r2[it] = ldraw<T>(X2 + o); }
  }
}

// transform the raw window and park it in LDS as f32.  MODE 0: act(x*c0 + c1);  MODE 1: g*c0 + z*c1 + c2.  Pixels
// outside the image and channel groups beyond C become zeros (zero padding applies to the TRANSFORMED tensor).
template <typename T, int MODE>
__device__ __forceinline__ void tile_store(float* tile, const Raw8<T> r1[NIT], const Raw8<T> r2[NIT], const TileMap& m, bool has2,
                                           int y0, int x0, int H, int W, const float* cf, int cl8, int act, float slope) {
  const float* c0 = cf + cl8;
  const float* c1 = cf + 64 + cl8;
  const float* c2 = cf + 128 + cl8;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    __builtin_amdgcn_sched_barrier(0);  // one slot at a time: keeps the live set at ~24 VGPRs instead of 5x that
    if (m.lds[it] >= 0) {
      const int yy = y0 + m.dy[it], xx = x0 + m.dx[it];
      const bool ok = m.cok && yy >= 0 && yy < H && xx >= 0 && xx < W;
      float v[8], a[8], z[8];
      cvt8(r1[it], a);
      if (MODE == 1) {
        if (has2) cvt8(r2[it], z);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) z[j] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u;
        if (MODE == 0) u = act_fwd(a[j] * c0[j] + c1[j], act, slope);
        else u = a[j] * c0[j] + z[j] * c1[j] + c2[j];
        v[j] = ok ? u : 0.f;
      }
      float* dst = tile + m.lds[it];
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

__device__ __forceinline__ void ld_lds8(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ void fill_weights(float* wl, const float* Wd, int C, int slab) {
  for (int i = threadIdx.x; i < 9 * 64; i += 256) {
    const int k = i >> 6, cl = i & 63;
    const int c = blockIdx.y * slab + cl;
    wl[i] = (cl < slab && c < C) ? Wd[(size_t)c * 9 + k] : 0.f;
  }
}

// sum v over the lanes of a wave that share a channel group (lane % cgl_n), result valid in lanes < cgl_n
__device__ __forceinline__ float cg_sum(float v, int cgl_n) {
  v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64);
  if (cgl_n == 4) v += __shfl_xor(v, 4, 64);
  return v;
}

// reduce per-thread channel sums over the workgroup (shuffles, then 4 wave partials in LDS) and push one atomic per
// channel to the global accumulator
__device__ __forceinline__ void dw_push_sums(float* red /*[4][2][64]*/, float s1[8], float s2[8], const DwThread& d, int C,
                                             float* osums, int oR) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = cg_sum(d.valid ? s1[j] : 0.f, d.cgl_n);
    s2[j] = cg_sum(d.valid ? s2[j] : 0.f, d.cgl_n);
  }
  __syncthreads();
  if (lane < d.cgl_n) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[w * 128 + lane * 8 + j] = s1[j]; red[w * 128 + 64 + lane * 8 + j] = s2[j]; }
  }
  __syncthreads();
  if (t < 128) {
    const int which = t >> 6, cl = t & 63;
    const int c = blockIdx.y * d.cgl_n * 8 + cl;
    if (cl < d.cgl_n * 8 && c < C) {
      const float s = red[t] + red[128 + t] + red[256 + t] + red[384 + t];
      const int rep = (blockIdx.x + blockIdx.y) % oR;
      atomicAdd(osums + (size_t)rep * 2 * C + (size_t)which * C + c, s);
    }
  }
}

struct Tiles { int TH, TW, ty, tx, per_img; long long total; };
__host__ __device__ inline Tiles make_tiles(int B, int OH, int OW, int stride_like) {
  Tiles t;
  t.TH = stride_like == 2 ? 4 : 8; t.TW = 8;
  t.ty = (OH + t.TH - 1) / t.TH; t.tx = (OW + t.TW - 1) / t.TW;
  t.per_img = t.ty * t.tx; t.total = (long long)B * t.per_img;
  return t;
}
struct TilePos { int b, y0, x0; };
__device__ __forceinline__ TilePos tile_pos(const Tiles& tl, long long ti) {
  TilePos p;
  p.b = (int)(ti / tl.per_img);
  const int tr = (int)(ti % tl.per_img);
  p.y0 = (tr / tl.tx) * tl.TH; p.x0 = (tr % tl.tx) * tl.TW;
  return p;
}

// ------------------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256, 3) void dw_fwd_kernel(const spb_dw_args_t a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;             // [9][64]
  float* cf = wl + 9 * 64;      // [3][64]
  float* red = cf + 3 * 64;     // [4][2][64]
  float* tile = red + 512;      // [ITH*ITW][PADC]
  SPB_TS(0);
  const DwThread d = dw_thread(a.C);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const int slab = d.cgl_n * 8;
  fill_weights(wl, a.Wd, C, slab);
  if (threadIdx.x < 64) {
    const int c = blockIdx.y * slab + threadIdx.x;
    float sc = 0.f, sh = 0.f;
    if (threadIdx.x < slab && c < C) bn_fwd_coef(a.pro, c, sc, sh);
    cf[threadIdx.x] = sc; cf[64 + threadIdx.x] = sh; cf[128 + threadIdx.x] = 0.f;
  }
  const Tiles tl = make_tiles(a.B, OH, OW, st);
  const int ITH = (tl.TH - 1) * st + 3, ITW = (tl.TW - 1) * st + 3;
  const TileMap m = make_map(ITH, ITW, d, C);
  const int c0 = d.cg * 8, l0 = d.cgl * 8, cl8 = (threadIdx.x % d.cgl_n) * 8;
  const T* X = reinterpret_cast<const T*>(a.X);
  T* Y = reinterpret_cast<T*>(a.Y);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  Raw8<T> r1[NIT], r2[NIT];
  long long ti = blockIdx.x;
  TilePos p = tile_pos(tl, ti < tl.total ? ti : 0);
  if (ti < tl.total) tile_issue<T, 0>(r1, r2, m, X, nullptr, p.b, p.y0 * st - 1, p.x0 * st - 1, H, W, C);
  __syncthreads();
  SPB_TS(1);
  for (; ti < tl.total; ti += gridDim.x) {
    tile_store<T, 0>(tile, r1, r2, m, false, p.y0 * st - 1, p.x0 * st - 1, H, W, cf, cl8, a.pro.act, a.pro.slope);
    const TilePos cur = p;
    __syncthreads();
    SPB_TS(2);
    if (ti + gridDim.x < tl.total) {  // next tile's loads fly while this one is computed
      p = tile_pos(tl, ti + gridDim.x);
      tile_issue<T, 0>(r1, r2, m, X, nullptr, p.b, p.y0 * st - 1, p.x0 * st - 1, H, W, C);
    }
    for (int o = d.pl; o < tl.TH * tl.TW; o += d.npl) {
      const int oy = o / tl.TW, ox = o % tl.TW;
      const int oh = cur.y0 + oy, ow = cur.x0 + ox;
      if (d.valid && oh < OH && ow < OW) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky)  // not unrolled: 3 taps (48 VGPRs of LDS reads) in flight instead of 9 (144)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float x[8], w[8];
            ld_lds8(tile + ((oy * st + ky) * ITW + ox * st + kx) * PADC + l0, x);
            ld_lds8(wl + (ky * 3 + kx) * 64 + l0, w);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += x[j] * w[j];
          }
        rnd8<T>(acc);
        st8<T>(Y + ((size_t)(cur.b * OH + oh) * OW + ow) * C + c0, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
      }
    }
    __syncthreads();
    SPB_TS(3);
  }
  if (a.epi_mode == 1) dw_push_sums(red, s1, s2, d, C, a.osums, a.oR);
  SPB_TS(4);
}

// ------------------------------------------------------------------------------------------------ input gradient
// dA[b,ih,iw,c] = sum_{ky,kx} dz[b,oh,ow,c] * w[c,ky,kx]  with  oh*stride - 1 + ky = ih.  Tiles are 8x8 INPUT pixels;
// the dz window they need is 10x10 (stride 1) or 5x5 (stride 2).
template <typename T>
__global__ __launch_bounds__(256, 2) void dw_dgrad_kernel(const spb_dw_args_t a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;             // [9][64]
  float* cf = wl + 9 * 64;      // [3][64] p0,p1,p2
  float* ce = cf + 3 * 64;      // [4][64] epi scale, shift, mean, invstd
  float* red = ce + 4 * 64;     // [4][2][64]
  float* tile = red + 512;
  const DwThread d = dw_thread(a.C);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const int slab = d.cgl_n * 8;
  fill_weights(wl, a.Wd, C, slab);
  if (threadIdx.x < 64) {
    const int c = blockIdx.y * slab + threadIdx.x;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (threadIdx.x < slab && c < C) {
      bn_bwd_coef(a.pro, c, p0, p1, p2);
      if (a.epi_mode == 2 && a.epi.gamma != nullptr) {
        bn_moments(a.epi, c, mu, is);
        sc = a.epi.gamma[c] * is;
        sh = a.epi.beta[c] - mu * sc;
      }
    }
    cf[threadIdx.x] = p0; cf[64 + threadIdx.x] = p1; cf[128 + threadIdx.x] = p2;
    ce[threadIdx.x] = sc; ce[64 + threadIdx.x] = sh; ce[128 + threadIdx.x] = mu; ce[192 + threadIdx.x] = is;
  }
  const Tiles tl = make_tiles(a.B, H, W, 1);  // 8x8 tiles over the INPUT image
  const int DTH = st == 2 ? tl.TH / 2 + 1 : tl.TH + 2, DTW = st == 2 ? tl.TW / 2 + 1 : tl.TW + 2;
  const TileMap m = make_map(DTH, DTW, d, C);
  const int c0 = d.cg * 8, l0 = d.cgl * 8, cl8 = (threadIdx.x % d.cgl_n) * 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Rg = reinterpret_cast<const T*>(a.res);
  const T* Zo = reinterpret_cast<const T*>(a.Zout);
  T* Y = reinterpret_cast<T*>(a.Y);
  const bool epi = a.epi_mode == 2;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  Raw8<T> r1[NIT], r2[NIT], zo[2], rr[2];
  // dz window origin (output coordinates) of the tile at input origin (y0,x0)
#define DW_DY0(y0_) (st == 2 ? (y0_) / 2 : (y0_) - 1)
#define DW_ISSUE(pp)                                                                                       \
  {                                                                                                        \
    tile_issue<T, 1>(r1, r2, m, G, Z, (pp).b, DW_DY0((pp).y0), DW_DY0((pp).x0), OH, OW, C);                 \
    if (epi) {                                                                                             \
      _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                      \
        const int o = d.pl + r * d.npl;                                                                    \
        const int ih = clampi((pp).y0 + o / tl.TW, 0, H - 1), iw = clampi((pp).x0 + o % tl.TW, 0, W - 1);   \
        const size_t off = ((size_t)((pp).b * H + ih) * W + iw) * C + (d.valid ? c0 : 0);                  \
        zo[r] = ldraw<T>(Zo + off);                                                                        \
        if (Rg) rr[r] = ldraw<T>(Rg + off);                                                                \
      }                                                                                                    \
    }                                                                                                      \
  }
  long long ti = blockIdx.x;
  TilePos p = tile_pos(tl, ti < tl.total ? ti : 0);
  if (ti < tl.total) DW_ISSUE(p);
  __syncthreads();
  for (; ti < tl.total; ti += gridDim.x) {
    const TilePos cur = p;
    const int dy0 = DW_DY0(cur.y0), dx0 = DW_DY0(cur.x0);
    tile_store<T, 1>(tile, r1, r2, m, Z != nullptr, dy0, dx0, OH, OW, cf, cl8, 0, 0.f);
    float zf[2][8], rf[2][8];
    if (epi) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        cvt8(zo[r], zf[r]);
        if (Rg) cvt8(rr[r], rf[r]);
      }
    }
    __syncthreads();
    if (ti + gridDim.x < tl.total) {
      p = tile_pos(tl, ti + gridDim.x);
      DW_ISSUE(p);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int o = d.pl + r * d.npl;
      const int iy = o / tl.TW, ix = o % tl.TW;
      const int ih = cur.y0 + iy, iw = cur.x0 + ix;
      if (o < tl.TH * tl.TW && d.valid && ih < H && iw < W) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
          const int ty = ih + 1 - ky;            // = oh*stride
          const bool oky = !(st == 2 && (ty & 1));
          const int ly = clampi((st == 2 ? ty / 2 : ty) - dy0, 0, DTH - 1);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int tx = iw + 1 - kx;
            const bool ok = oky && !(st == 2 && (tx & 1));
            const int lx = clampi((st == 2 ? tx / 2 : tx) - dx0, 0, DTW - 1);
            float x[8], w[8];
            ld_lds8(tile + (ly * DTW + lx) * PADC + l0, x);
            ld_lds8(wl + (ky * 3 + kx) * 64 + l0, w);
            const float msk = ok ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += msk * x[j] * w[j];
          }
        }
        if (epi) {
          if (Rg) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += rf[r][j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float z = zf[r][j];
            const float u = z * ce[l0 + j] + ce[64 + l0 + j];
            acc[j] = rnd<T>(acc[j] * act_grad(u, a.epi.act, a.epi.slope));
            s1[j] += acc[j];
            s2[j] += acc[j] * ((z - ce[128 + l0 + j]) * ce[192 + l0 + j]);
          }
        }
        st8<T>(Y + ((size_t)(cur.b * H + ih) * W + iw) * C + c0, acc);
      }
    }
    __syncthreads();
  }
#undef DW_ISSUE
#undef DW_DY0
  if (epi) dw_push_sums(red, s1, s2, d, C, a.osums, a.oR);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[c,ky,kx] += sum_{b,oh,ow} dz[b,oh,ow,c] * act(bn(x))[b, oh*s-1+ky, ow*s-1+kx, c]
template <typename T>
__global__ __launch_bounds__(256, 2) void dw_wgrad_kernel(const spb_dw_args_t a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* cf = smem;             // [3][64] input scale, shift, 0
  float* cz = cf + 3 * 64;      // [3][64] p0,p1,p2
  float* red = cz + 3 * 64;     // [4][64*9]
  float* tile = red + 4 * 64 * 9;
  const DwThread d = dw_thread(a.C);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const int slab = d.cgl_n * 8;
  const int t = threadIdx.x;
  if (t < 64) {
    const int c = blockIdx.y * slab + t;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, sc = 0.f, sh = 0.f;
    if (t < slab && c < C) { bn_bwd_coef(a.pro, c, p0, p1, p2); bn_fwd_coef(a.pro_in, c, sc, sh); }
    cz[t] = p0; cz[64 + t] = p1; cz[128 + t] = p2; cf[t] = sc; cf[64 + t] = sh; cf[128 + t] = 0.f;
  }
  const Tiles tl = make_tiles(a.B, OH, OW, st);
  const int ITH = (tl.TH - 1) * st + 3, ITW = (tl.TW - 1) * st + 3;
  const TileMap m = make_map(ITH, ITW, d, C);
  const int c0 = d.cg * 8, l0 = d.cgl * 8, cl8 = (threadIdx.x % d.cgl_n) * 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Xin = reinterpret_cast<const T*>(a.Xin);
  float aw[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[k][j] = 0.f;
  Raw8<T> r1[NIT], r2[NIT], gr[2], zr[2];
#define DW_ISSUE(pp)                                                                                         \
  {                                                                                                          \
    tile_issue<T, 0>(r1, r2, m, Xin, nullptr, (pp).b, (pp).y0 * st - 1, (pp).x0 * st - 1, H, W, C);            \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                          \
      const int o = d.pl + r * d.npl;                                                                        \
      const int oh = clampi((pp).y0 + o / tl.TW, 0, OH - 1), ow = clampi((pp).x0 + o % tl.TW, 0, OW - 1);     \
      const size_t off = ((size_t)((pp).b * OH + oh) * OW + ow) * C + (d.valid ? c0 : 0);                    \
      gr[r] = ldraw<T>(G + off);                                                                             \
      if (Z) zr[r] = ldraw<T>(Z + off);                                                                      \
    }                                                                                                        \
  }
  long long ti = blockIdx.x;
  TilePos p = tile_pos(tl, ti < tl.total ? ti : 0);
  if (ti < tl.total) DW_ISSUE(p);
  __syncthreads();
  for (; ti < tl.total; ti += gridDim.x) {
    const TilePos cur = p;
    tile_store<T, 0>(tile, r1, r2, m, false, cur.y0 * st - 1, cur.x0 * st - 1, H, W, cf, cl8, a.pro_in.act, a.pro_in.slope);
    float dz[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float g[8], z[8];
      cvt8(gr[r], g);
      if (Z) cvt8(zr[r], z);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) dz[r][j] = g[j] * cz[l0 + j] + z[j] * cz[64 + l0 + j] + cz[128 + l0 + j];
    }
    __syncthreads();
    if (ti + gridDim.x < tl.total) {
      p = tile_pos(tl, ti + gridDim.x);
      DW_ISSUE(p);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int o = d.pl + r * d.npl;
      const int oy = o / tl.TW, ox = o % tl.TW;
      if (o < tl.TH * tl.TW && d.valid && cur.y0 + oy < OH && cur.x0 + ox < OW) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float x[8];
            ld_lds8(tile + ((oy * st + ky) * ITW + ox * st + kx) * PADC + l0, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) aw[ky * 3 + kx][j] += dz[r][j] * x[j];
          }
      }
    }
    __syncthreads();
  }
#undef DW_ISSUE
  // workgroup reduction: shuffles over the lanes that share a channel group, 4 wave partials in LDS, one atomic per weight
  const int lane = t & 63, w = t >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = cg_sum(d.valid ? aw[k][j] : 0.f, d.cgl_n);
      if (lane < d.cgl_n) red[w * 576 + (lane * 8 + j) * 9 + k] = v;
    }
  __syncthreads();
  for (int i = t; i < slab * 9; i += 256) {
    const int c = blockIdx.y * slab + i / 9;
    if (c < C) atomicAdd(a.dW + (size_t)c * 9 + (i % 9), red[i] + red[576 + i] + red[1152 + i] + red[1728 + i]);
  }
}

int dw_check(const spb_dw_args_t* a) {
  if (!a || !a->X || !a->Wd) return SPB_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 7)) return SPB_E_SHAPE;
  if (a->stride != 1 && a->stride != 2) return SPB_E_SHAPE;
  return 0;
}

// persistent grid: about `per_cu` workgroups per CU in total (256 CUs), never more than one per tile
dim3 dw_grid(const spb_dw_args_t& a, const Tiles& tl, int per_cu) {
  const int CG = a.C >> 3;
  const int cgl_n = CG < 8 ? 4 : 8;
  const int gy = (CG + cgl_n - 1) / cgl_n;
  long long gx = (256LL * per_cu + gy - 1) / gy;
  if (gx > tl.total) gx = tl.total;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

template <typename K>
void dw_set_lds(K kernel) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

}  // namespace

extern "C" int spb_dwconv_fwd(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y || (a->epi_mode == 1 && (!a->osums || a->oR < 1))) return SPB_E_ARG;
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const Tiles tl = make_tiles(a->B, OH, OW, st);
  const int ITH = (tl.TH - 1) * st + 3, ITW = (tl.TW - 1) * st + 3;
  const size_t lds = (size_t)(9 * 64 + 3 * 64 + 512 + ITH * ITW * PADC) * sizeof(float);
  const dim3 grid = dw_grid(*a, tl, st == 2 ? 3 : 4);
  static bool once = false;
  if (!once) { dw_set_lds(dw_fwd_kernel<bf16_t>); dw_set_lds(dw_fwd_kernel<float>); once = true; }
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dwconv_dgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y) return SPB_E_ARG;
  if (a->epi_mode == 2 && (!a->osums || a->oR < 1 || !a->Zout)) return SPB_E_ARG;
  const int st = a->stride;
  const Tiles tl = make_tiles(a->B, a->H, a->W, 1);
  const int DTH = st == 2 ? tl.TH / 2 + 1 : tl.TH + 2, DTW = st == 2 ? tl.TW / 2 + 1 : tl.TW + 2;
  const size_t lds = (size_t)(9 * 64 + 3 * 64 + 4 * 64 + 512 + DTH * DTW * PADC) * sizeof(float);
  const dim3 grid = dw_grid(*a, tl, 2);
  static bool once = false;
  if (!once) { dw_set_lds(dw_dgrad_kernel<bf16_t>); dw_set_lds(dw_dgrad_kernel<float>); once = true; }
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_dgrad_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_dgrad_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dwconv_wgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->dW || !a->Xin) return SPB_E_ARG;
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const Tiles tl = make_tiles(a->B, OH, OW, st);
  const int ITH = (tl.TH - 1) * st + 3, ITW = (tl.TW - 1) * st + 3;
  const size_t lds = (size_t)(3 * 64 + 3 * 64 + 4 * 64 * 9 + ITH * ITW * PADC) * sizeof(float);
  const dim3 grid = dw_grid(*a, tl, 2);
  static bool once = false;
  if (!once) { dw_set_lds(dw_wgrad_kernel<bf16_t>); dw_set_lds(dw_wgrad_kernel<float>); once = true; }
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_wgrad_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_wgrad_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}
