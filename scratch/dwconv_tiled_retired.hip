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
// ablation switches for scratch/ubench_dwbwd.hip (0 in the product build): 1 no taps, 2 no transform, 4 no DMA after
// the first tile, 8 no output store, 16 no input-side loads
#ifndef SPB_ABL
#define SPB_ABL 0
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
template <typename T, int MODE, int N = NIT>
__device__ __forceinline__ void tile_issue(Raw8<T>* r1, Raw8<T>* r2, const TileMap& m, const T* X, const T* X2, int b,
                                           int y0, int x0, int H, int W, int C) {
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int y = clampi(y0 + m.dy[it], 0, H - 1), x = clampi(x0 + m.dx[it], 0, W - 1);
    const size_t o = ((size_t)(b * H + y) * W + x) * C + m.coff;
    r1[it] = ldraw<T>(X + o);
    if (MODE == 1) { if (X2) r2[it] = ldraw<T>(X2 + o); }
  }
}

// transform the raw window and park it in LDS as f32.  MODE 0: act(x*c0 + c1);  MODE 1: g*c0 + z*c1 + c2.  Pixels
// outside the image and channel groups beyond C become zeros (zero padding applies to the TRANSFORMED tensor).
template <typename T, int MODE, int N = NIT>
__device__ __forceinline__ void tile_store(float* tile, const Raw8<T>* r1, const Raw8<T>* r2, const TileMap& m, bool has2,
                                           int y0, int x0, int H, int W, const float* cf, int cl8, int act, float slope) {
  const float* c0 = cf + cl8;
  const float* c1 = cf + 64 + cl8;
  const float* c2 = cf + 128 + cl8;
#pragma unroll
  for (int it = 0; it < N; ++it) {
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

template <typename R>
__device__ __forceinline__ R sel8(int r, const R& a, const R& b) { return r == 0 ? a : b; }

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
// WG = true fuses the weight gradient: the (output pixel, tap) pairs of dW[c,ky,kx] = sum dz[q]*a[q*s-1+k] are in
// one-to-one correspondence with (input pixel p, tap) pairs, and the thread that owns input pixel p already holds a[p]
// (it loads the input-side z for the activation mask) and gathers exactly the dz taps the sum needs -- 8 more FMAs per
// tap instead of a second kernel that re-reads g, z and the input (measured: 0.94 ms of the 6.8 ms step).
//
// Staging.  The raw g / z windows of tile t+1 are fetched by LDS-DMA (global_load_lds_dwordx4, no VGPR in flight) into
// the second of two LDS buffers while tile t is computed; a buffer is then transformed IN PLACE (raw bf16 g,z -> f32 dz,
// the same number of bytes) and the taps read the f32 tile.  A register prefetch of the window plus the 72 weight-
// gradient accumulators does not fit 256 VGPRs (167..312 spilled dwords in every variant tried, and a spilled load
// destination turns the prefetch into a blocking load).
struct DgLayout { int DTH, DTW, npix, pixB, RB, NI, BUFB; };
__host__ __device__ inline DgLayout dg_layout(int st, int slab, int esize) {
  DgLayout L;
  L.DTH = st == 2 ? 5 : 10; L.DTW = L.DTH;          // dz window of an 8x8 input tile
  L.npix = L.DTH * L.DTW;
  L.pixB = slab * esize;                            // bytes of one pixel's channel slab
  L.RB = (L.npix * L.pixB + 1023) & ~1023;          // one raw window = whole 64-lane x 16-byte DMA instructions
  L.NI = L.RB >> 10;
  const int tileB = L.npix * PADC * 4;
  L.BUFB = ((2 * L.RB > tileB ? 2 * L.RB : tileB) + 15) & ~15;
  return L;
}

template <typename T, bool WG, bool EPI>
__global__ __launch_bounds__(256, 2) void dw_dgrad_kernel(const spb_dw_args_t a) {
  constexpr int ND = 4;                                   // 10x10 window x 8 channel groups / 256 threads
  constexpr int MAXI = sizeof(T) == 2 ? 4 : 7;            // DMA instructions per wave per raw window
  constexpr int EPG = 16 / sizeof(T);                     // elements per 16-byte DMA granule
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;             // [9][64]
  float* cf = wl + 9 * 64;      // [3][64] p0,p1,p2
  float* ce = cf + 3 * 64;      // [4][64] input-side scale, shift, mean, invstd
  float* red = ce + 4 * 64;     // [4][2][64]  (WG: [4][64*9])
  char* bufs = reinterpret_cast<char*>(red + (WG ? 4 * 64 * 9 : 512));
  const DwThread d = dw_thread(a.C);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const int slab = d.cgl_n * 8;
  constexpr bool epi = EPI;
  const DgLayout L = dg_layout(st, slab, (int)sizeof(T));
  fill_weights(wl, a.Wd, C, slab);
  if (threadIdx.x < 64) {
    const int c = blockIdx.y * slab + threadIdx.x;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (threadIdx.x < slab && c < C) {
      bn_bwd_coef(a.pro, c, p0, p1, p2);
      if ((epi || WG) && a.epi.gamma != nullptr) {
        bn_moments(a.epi, c, mu, is);
        sc = a.epi.gamma[c] * is;
        sh = a.epi.beta[c] - mu * sc;
      }
    }
    cf[threadIdx.x] = p0; cf[64 + threadIdx.x] = p1; cf[128 + threadIdx.x] = p2;
    ce[threadIdx.x] = sc; ce[64 + threadIdx.x] = sh; ce[128 + threadIdx.x] = mu; ce[192 + threadIdx.x] = is;
  }
  const Tiles tl = make_tiles(a.B, H, W, 1);  // 8x8 tiles over the INPUT image
  const int DTH = L.DTH, DTW = L.DTW;
  const TileMap m = make_map(DTH, DTW, d, C);   // transform slots: (pixel, channel group) -> f32 tile offset
  int rawoff[ND];                               // byte offset of the slot's 8 channels in a raw window
#pragma unroll
  for (int k = 0; k < ND; ++k)
    rawoff[k] = m.lds[k] >= 0 ? (m.lds[k] / PADC) * L.pixB + (threadIdx.x % d.cgl_n) * 8 * (int)sizeof(T) : 0;
  // DMA map: instruction i = wave + 4k covers granules i*64 + lane; granule -> (window pixel, 16-byte part of its slab)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int gpp = L.pixB >> 4;
  short qy[MAXI], qx[MAXI];
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    int pix = ((wv + 4 * k) * 64 + lane) / gpp;
    pix = pix < L.npix ? pix : L.npix - 1;
    qy[k] = (short)(pix / DTW); qx[k] = (short)(pix % DTW);
  }
  int qch = blockIdx.y * slab + (lane % gpp) * EPG;
  qch = qch + EPG <= C ? qch : C - EPG;
  const unsigned buf_lds = lds_addr(bufs);
  const int c0 = d.cg * 8, l0 = d.cgl * 8, cl8 = (threadIdx.x % d.cgl_n) * 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Rg = reinterpret_cast<const T*>(a.res);
  const T* Zo = reinterpret_cast<const T*>(a.Zout);
  T* Y = reinterpret_cast<T*>(a.Y);
  constexpr bool need_in = EPI || WG;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  float aw[WG ? 9 : 1][8];
#pragma unroll
  for (int k = 0; k < (WG ? 9 : 1); ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[k][j] = 0.f;
  Raw8<T> zo[2], rr[2];
  // dz window origin (output coordinates) of the tile at input origin (y0,x0)
#define DW_DY0(y0_) (st == 2 ? (y0_) / 2 : (y0_) - 1)
#define DW_ISSUE(pp, bi)                                                                                   \
  {                                                                                                        \
    const int oy0 = DW_DY0((pp).y0), ox0 = DW_DY0((pp).x0);                                                \
    const unsigned lb = buf_lds + (unsigned)((bi) * L.BUFB);                                               \
    _Pragma("unroll") for (int k = 0; k < MAXI; ++k) {                                                     \
      const int i = wv + 4 * k;                                                                            \
      if (i < L.NI) {                                                                                      \
        const int y = clampi(oy0 + qy[k], 0, OH - 1), x = clampi(ox0 + qx[k], 0, OW - 1);                   \
        const size_t o = ((size_t)((pp).b * OH + y) * OW + x) * C + qch;                                   \
        dma16(G + o, lb + (unsigned)(i << 10));                                                            \
        dma16(Z + o, lb + (unsigned)(L.RB + (i << 10)));                                                    \
      }                                                                                                    \
    }                                                                                                      \
  }
  // the owned pixels' input-side z (activation mask, a[p]) and residual gradient: ordinary loads, issued AFTER the taps
  // of the previous tile so that they are not live across them; the counted wait at the loop top lets them fly on
#define DW_ISSUE_IN(pp)                                                                                    \
  if (need_in) {                                                                                           \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                        \
      const int o = d.pl + r * d.npl;                                                                      \
      const int ih = clampi((pp).y0 + o / tl.TW, 0, H - 1), iw = clampi((pp).x0 + o % tl.TW, 0, W - 1);     \
      const size_t off = ((size_t)((pp).b * H + ih) * W + iw) * C + (d.valid ? c0 : 0);                    \
      zo[r] = ldraw<T>(Zo + off);                                                                          \
      if (epi) { if (Rg) rr[r] = ldraw<T>(Rg + off); }                                                     \
    }                                                                                                      \
  }
  // one row of taps (fixed ky): gather dz from the LDS window; accumulate the input gradient and, fused, dz * a[p]
#define DW_TAPS(ky)                                                                                        \
  {                                                                                                        \
    const int ty = ih + 1 - (ky);                                                                          \
    const bool oky = !(st == 2 && (ty & 1));                                                               \
    const int ly = clampi((st == 2 ? ty / 2 : ty) - dy0, 0, DTH - 1);                                      \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                     \
      const int tx = iw + 1 - kx;                                                                          \
      const bool ok = oky && !(st == 2 && (tx & 1));                                                       \
      const int lx = clampi((st == 2 ? tx / 2 : tx) - dx0, 0, DTW - 1);                                    \
      float x[8], w[8];                                                                                    \
      ld_lds8(tile + (ly * DTW + lx) * PADC + l0, x);                                                      \
      ld_lds8(wl + ((ky) * 3 + kx) * 64 + l0, w);                                                          \
      const float msk = ok ? 1.f : 0.f;                                                                    \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                      \
        const float xm = msk * x[j];                                                                       \
        acc[j] += xm * w[j];                                                                               \
        if (WG) aw[WG ? (ky) * 3 + kx : 0][j] += xm * ap[j];                                               \
      }                                                                                                    \
    }                                                                                                      \
  }
#define DW_TAP_LD(k, X_, W_, M_)                                                                           \
  {                                                                                                        \
    const int ty = ih + 1 - (k) / 3, tx = iw + 1 - (k) % 3;                                                \
    const bool ok = !(st == 2 && ((ty | tx) & 1));                                                         \
    const int ly = clampi((st == 2 ? ty / 2 : ty) - dy0, 0, DTH - 1);                                      \
    const int lx = clampi((st == 2 ? tx / 2 : tx) - dx0, 0, DTW - 1);                                      \
    ld_lds8(tile + (ly * DTW + lx) * PADC + l0, X_);                                                       \
    ld_lds8(wl + (k) * 64 + l0, W_);                                                                       \
    M_ = ok ? 1.f : 0.f;                                                                                   \
    asm volatile("" ::: "memory");                                                                         \
  }
#define DW_TAP_FMA(k, X_, W_, M_)                                                                          \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
      const float xm = M_ * X_[j];                                                                         \
      acc[j] += xm * W_[j];                                                                                \
      aw[WG ? (k) : 0][j] += xm * ap[j];                                                                   \
    }                                                                                                      \
  }
  long long ti = blockIdx.x;
  TilePos p = tile_pos(tl, ti < tl.total ? ti : 0);
  if (ti < tl.total) { DW_ISSUE(p, 0); DW_ISSUE_IN(p); }
  int cb = 0;
  constexpr int LPT = sizeof(T) == 2 ? 1 : 2;   // global_load_dwordx4 per Raw8
#pragma clang loop unroll(disable)
  for (; ti < tl.total; ti += gridDim.x, cb ^= 1) {
    const TilePos cur = p;
    const int dy0 = DW_DY0(cur.y0), dx0 = DW_DY0(cur.x0);
    char* buf = bufs + cb * L.BUFB;
    float* tile = reinterpret_cast<float*>(buf);
    // this wave's share of the window has landed (vm ops return in order; only the zo/rr loads are younger) ...
    if (!need_in) wait_vmcnt<0>();
    else if (epi && Rg) wait_vmcnt<4 * LPT>();
    else wait_vmcnt<2 * LPT>();
    __syncthreads();     // ... and so has everybody else's (first pass: also orders the coefficient tables)
    Raw8<T> r1[ND], r2[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      r1[k] = *reinterpret_cast<const Raw8<T>*>(buf + rawoff[k]);
      r2[k] = *reinterpret_cast<const Raw8<T>*>(buf + L.RB + rawoff[k]);
    }
    __syncthreads();     // every raw read done: the f32 tile may overwrite the window
    if (!(SPB_ABL & 2)) tile_store<T, 1, ND>(tile, r1, r2, m, true, dy0, dx0, OH, OW, cf, cl8, 0, 0.f);
    __syncthreads();
    if (ti + gridDim.x < tl.total) {  // the other buffer was last read before the barriers above
      p = tile_pos(tl, ti + gridDim.x);
      if (!(SPB_ABL & 4)) DW_ISSUE(p, cb ^ 1);
    }
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int o = d.pl + r * d.npl;
      const int iy = o / tl.TW, ix = o % tl.TW;
      const int ih = cur.y0 + iy, iw = cur.x0 + ix;
      if (o < tl.TH * tl.TW && d.valid && ih < H && iw < W) {
        float acc[8], ap[8], zf[8];
        if (need_in) cvt8(sel8(r, zo[0], zo[1]), zf);
        asm volatile("" ::: "memory");  // keep the coefficient reads here (hoisted out of the r loop they cost 32 VGPRs)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[j] = 0.f;
          ap[j] = WG ? act_fwd(zf[j] * ce[l0 + j] + ce[64 + l0 + j], a.epi.act, a.epi.slope) : 0.f;
        }
        if (SPB_ABL & 1) {
        } else if constexpr (WG) {
          // all 9 taps unrolled (aw needs static indices) but software-pipelined two deep by hand: left alone the
          // scheduler hoists all 18 LDS reads to the top (144 VGPRs) and spills
          float xa[8], wa[8], xb[8], wb[8], ma, mb;
          DW_TAP_LD(0, xa, wa, ma)
          DW_TAP_LD(1, xb, wb, mb) DW_TAP_FMA(0, xa, wa, ma)
          DW_TAP_LD(2, xa, wa, ma) DW_TAP_FMA(1, xb, wb, mb)
          DW_TAP_LD(3, xb, wb, mb) DW_TAP_FMA(2, xa, wa, ma)
          DW_TAP_LD(4, xa, wa, ma) DW_TAP_FMA(3, xb, wb, mb)
          DW_TAP_LD(5, xb, wb, mb) DW_TAP_FMA(4, xa, wa, ma)
          DW_TAP_LD(6, xa, wa, ma) DW_TAP_FMA(5, xb, wb, mb)
          DW_TAP_LD(7, xb, wb, mb) DW_TAP_FMA(6, xa, wa, ma)
          DW_TAP_LD(8, xa, wa, ma) DW_TAP_FMA(7, xb, wb, mb)
          DW_TAP_FMA(8, xa, wa, ma)
        } else {
#pragma unroll 1
          for (int ky = 0; ky < 3; ++ky) DW_TAPS(ky)
        }
        if (epi) {
          asm volatile("" ::: "memory");
          if (Rg) {
            float rf[8];
            cvt8(sel8(r, rr[0], rr[1]), rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += rf[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float z = zf[j];
            const float u = z * ce[l0 + j] + ce[64 + l0 + j];
            acc[j] = rnd<T>(acc[j] * act_grad(u, a.epi.act, a.epi.slope));
            s1[j] += acc[j];
            s2[j] += acc[j] * ((z - ce[128 + l0 + j]) * ce[192 + l0 + j]);
          }
        }
        if (!(SPB_ABL & 8) || acc[0] == 123.f) st8<T>(Y + ((size_t)(cur.b * H + ih) * W + iw) * C + c0, acc);
      }
    }
    if (!(SPB_ABL & 16) && ti + gridDim.x < tl.total) DW_ISSUE_IN(p);
  }
#undef DW_ISSUE
#undef DW_ISSUE_IN
#undef DW_DY0
#undef DW_TAPS
#undef DW_TAP_LD
#undef DW_TAP_FMA
  __syncthreads();
  if (epi) dw_push_sums(red, s1, s2, d, C, a.osums, a.oR);
  if constexpr (WG) {
    const int t = threadIdx.x, w = t >> 6;
    __syncthreads();
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

// row-unit kernels (dwconv_rows.hip); g_dw_mode 1 (default) routes forward and input gradient through them, 0 keeps
// the LDS-tiled kernels of this file (A/B measurements, spb_debug_set_dw_mode)
int spb_dwr_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s);
int spb_dwr_bwd(int dtype, const spb_dw_args_t* a, hipStream_t s);
int spb_dwr_wgrad(int dtype, const spb_dw_args_t* a, hipStream_t s);
static int g_dw_mode = 1;
extern "C" int spb_debug_set_dw_mode(int mode) { g_dw_mode = mode; return 0; }

extern "C" int spb_dwconv_fwd(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y || (a->epi_mode == 1 && (!a->osums || a->oR < 1))) return SPB_E_ARG;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  if (g_dw_mode == 1) {
    spb_dwr_fwd(dtype, a, (hipStream_t)stream);
    SPB_CHECK_LAUNCH();
    return 0;
  }
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

// dW != NULL fuses the weight gradient into the same pass: then Zout must be the input tensor of the convolution and
// `epi` its BN/activation (what spb_dwconv_wgrad takes as Xin / pro_in), also when epi_mode == 0.
extern "C" int spb_dwconv_dgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y) return SPB_E_ARG;
  if (a->epi_mode == 2 && (!a->osums || a->oR < 1 || !a->Zout)) return SPB_E_ARG;
  const bool wg = a->dW != nullptr;
  if (wg && !a->Zout) return SPB_E_ARG;
  const Tiles tl = make_tiles(a->B, a->H, a->W, 1);
  const int slab = (a->C >> 3) < 8 ? 32 : 64;
  const DgLayout L = dg_layout(a->stride, slab, dtype == SPB_BF16 ? 2 : 4);
  const size_t lds = (size_t)(9 * 64 + 3 * 64 + 4 * 64 + (wg ? 4 * 64 * 9 : 512)) * sizeof(float) + 2 * (size_t)L.BUFB;
  const dim3 grid = dw_grid(*a, tl, 2);
  spb_dw_args_t k = *a;
  if (!k.X2) k.X2 = k.X;  // no BN behind the convolution: p1 == 0, the kernel still reads a (finite) second operand
  const bool epi = k.epi_mode == 2;
  hipStream_t s = (hipStream_t)stream;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  if (g_dw_mode == 1) {
    spb_dwr_bwd(dtype, &k, s);
    SPB_CHECK_LAUNCH();
    return 0;
  }
#define DG_LAUNCH(T_, WG_, EPI_)                                                       \
  {                                                                                    \
    static bool once = false;                                                          \
    if (!once) { dw_set_lds(dw_dgrad_kernel<T_, WG_, EPI_>); once = true; }            \
    hipLaunchKernelGGL((dw_dgrad_kernel<T_, WG_, EPI_>), grid, dim3(256), lds, s, k);  \
  }
#define DG_PICK(T_)                                                                    \
  {                                                                                    \
    if (wg) { if (epi) DG_LAUNCH(T_, true, true) else DG_LAUNCH(T_, true, false) }     \
    else { if (epi) DG_LAUNCH(T_, false, true) else DG_LAUNCH(T_, false, false) }      \
  }
  if (dtype == SPB_BF16) DG_PICK(bf16_t)
  else if (dtype == SPB_F32) DG_PICK(float)
  else return SPB_E_ARG;
#undef DG_PICK
#undef DG_LAUNCH
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dwconv_wgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->dW || !a->Xin) return SPB_E_ARG;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  if (g_dw_mode == 1) {    // row-unit kernel, weight-gradient-only instance: the input tensor travels as Zout / epi there
    spb_dw_args_t k = *a;
    k.Zout = a->Xin; k.epi = a->pro_in; k.Y = nullptr; k.epi_mode = 0; k.res = nullptr;
    if (!k.X2) k.X2 = k.X;
    spb_dwr_wgrad(dtype, &k, (hipStream_t)stream);
    SPB_CHECK_LAUNCH();
    return 0;
  }
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
