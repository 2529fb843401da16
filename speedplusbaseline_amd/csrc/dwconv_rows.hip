// Depthwise 3x3 convolution, "row unit" kernels: forward and fused backward (input gradient + weight gradient).
// Replaces nn.Conv2d(groups=C, k=3) + BatchNorm2d + ReLU6/ReLU of reference park2019.py:47-49 and of the torchvision MobileNetV2
// inverted residual (park2019.py:107-108); C-ABI: spb_dwconv_fwd / _dgrad / _wgrad (include/spb_hip.h), at the end of this file.
//
// Why this formulation.  The first, LDS-tiled kernels (retired: scratch/dwconv_tiled_retired.hip) staged an 8x8 window in LDS and read 9 taps x (32 B data +
// 32 B weights) per pixel and 8 channels from it.  Ablation on MI355X (scratch/ubench_dwbwd.hip) showed that to be
// LDS-bandwidth bound: half of the kernel time is the tap loop at ~2x the 128 B/clk/CU LDS floor, and another 40 % is
// the staging + 3 barriers per tile.  Here the taps never touch LDS:
//   * a UNIT is one 16-lane DPP row: 16 adjacent columns x one 8-channel group, marching down image rows;
//   * each lane keeps its own column of the last three transformed rows in registers (sliding window); the left and
//     right taps come from the neighbouring lanes through DPP row shifts (no barrier, no bank conflicts);
//   * 14 of the 16 lanes produce outputs (the outer two are the halo columns), which tiles every KRN feature-map
//     width exactly (112, 56, 28, 14 = k x 14); stride 2 maps lanes to the coarse (output / dz) columns, two fine
//     columns per lane, so that no tap is computed and masked away;
//   * the 4 rows of a wave are 4 adjacent channel groups of the same columns: 64 contiguous bytes per pixel;
//   * a wave is PERSISTENT over (image, column strip, row segment) tasks of its channel quad; the rows it will need are
//     one continuous stream that an LDS-DMA ring (global_load_lds_dwordx4, RD elements deep, wave-private, no VGPRs,
//     counted s_waitcnt) keeps ahead of the compute -- also across task boundaries;
//   * weights and BN coefficients of the block's 4 channel groups sit in 2 KB of LDS and are read as broadcasts;
//   * per-channel BN sums and the 9 x 8 weight-gradient partials stay in registers for the whole kernel, are reduced
//     over the 16 lanes with DPP butterflies, over the 4 waves in LDS, and leave as one atomic per value and block.
// HBM traffic is the algorithmic minimum plus 2/R re-read halo rows (they hit L2).
#include "common.h"

// ablation switches for scratch/ubench_dwbwd.hip (0 in the product build): 32 empty stream, 64 no final reduction,
// 128 no coefficient prologue
#ifndef SPB_ABL
#define SPB_ABL 0
#endif

namespace {

constexpr int CFN = 128;  // floats of per-channel-group coefficients: w[9][8] | p0,p1,p2 (bwd) or sc,sh (fwd) | sc,sh,mu,is
// ring depth: stream elements in flight per wave, sized so that a block's rings stay under ~48 KB (3 blocks per CU by
// LDS; with 96 KB rings the stride-2 backward ran one block per CU)
#ifndef SPB_DW_RING_BUDGET
#define SPB_DW_RING_BUDGET 49152
#endif
#ifndef SPB_DW_RING_CAP
#define SPB_DW_RING_CAP 4
#endif
template <typename T, int NS> struct RingDepth {
  static constexpr int bytes = NS * (sizeof(T) == 2 ? 1 : 2) * 4096;   // one element of all 4 waves
  static constexpr int v = SPB_DW_RING_BUDGET / bytes >= SPB_DW_RING_CAP ? SPB_DW_RING_CAP : (SPB_DW_RING_BUDGET / bytes >= 2 ? SPB_DW_RING_BUDGET / bytes : 2);
};

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
// DPP butterflies (VALU, no LDS crossbar): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.  The
// __shfl_xor form compiled to ds_bpermute_b32: 352 of them in the backward kernel's final reduction (88 values), ~3 us
// of every launch of a kernel whose whole job is 9 row steps on the 14x14 maps.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return v;
}

// launch geometry (host computed)
struct Geo {
  int R;        // output rows (ST=2 backward: dz rows) per task
  int E;        // stream elements per task (R + halo)
  int nseg, nstrip, ntasks;
  int nb;       // blocks per channel quad
  int xcd;      // 1: blocks of one task range (bq) and ALL channel quads sit on one XCD (nb % 8 == 0)
};

// wave-uniform position in the stream of (task, element) pairs
struct Cursor { int k, i, sx, sy, b; };
__device__ __forceinline__ void decode_task(Cursor& c, const Geo& g) {
  int t = c.k;
  c.sx = t % g.nstrip; t /= g.nstrip;
  c.sy = t % g.nseg;
  c.b = t / g.nseg;
}
__device__ __forceinline__ void advance(Cursor& c, const Geo& g, int nwq) {
  if (++c.i == g.E) { c.i = 0; c.k += nwq; decode_task(c, g); }
}

// DMA of one 8-channel vector per lane into ring piece `piece` (1 KB per 16-byte part: f32 needs two)
template <typename T>
__device__ __forceinline__ void dma_vec(const T* src, unsigned slot_lds, int piece) {
  constexpr int PARTS = sizeof(T) == 2 ? 1 : 2;
#pragma unroll
  for (int h = 0; h < PARTS; ++h)
    dma16(reinterpret_cast<const char*>(src) + 16 * h, slot_lds + (unsigned)((piece * PARTS + h) << 10));
}
template <typename T>
__device__ __forceinline__ Raw8<T> ring_vec(const char* slot, int piece, int lane);
template <>
__device__ __forceinline__ Raw8<bf16_t> ring_vec<bf16_t>(const char* slot, int piece, int lane) {
  Raw8<bf16_t> r; r.u = *reinterpret_cast<const uint4*>(slot + (piece << 10) + lane * 16); return r;
}
template <>
__device__ __forceinline__ Raw8<float> ring_vec<float>(const char* slot, int piece, int lane) {
  Raw8<float> r;
  r.a = *reinterpret_cast<const float4*>(slot + ((piece * 2) << 10) + lane * 16);
  r.b = *reinterpret_cast<const float4*>(slot + ((piece * 2 + 1) << 10) + lane * 16);
  return r;
}

// block-level reduction of per-lane channel partials: rows (DPP) -> waves (LDS) -> one atomic per value.
// scratch: [4 waves][4 rows][NV*8] floats.  dst(row, j, vi) = dst0[row] + j*stride_c + vi*stride_v
template <int NV>
__device__ __forceinline__ void push_block(float* scratch, const float (&v)[NV][8], int wave, int lane, const bool rowok[4],
                                           float* const dst0[4], int stride_c, int stride_v) {
  const int l16 = lane & 15, row = lane >> 4;
  float* mine = scratch + (wave * 4 + row) * NV * 8;
#pragma unroll
  for (int vi = 0; vi < NV; ++vi)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = row_sum(v[vi][j]);
      if (l16 == ((vi * 8 + j) & 15)) mine[vi * 8 + j] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * NV * 8; i += 256) {
    const int r = i / (NV * 8), e = i % (NV * 8);
    if (rowok[r]) {
      const float s = scratch[(0 * 4 + r) * NV * 8 + e] + scratch[(1 * 4 + r) * NV * 8 + e] + scratch[(2 * 4 + r) * NV * 8 + e] +
                      scratch[(3 * 4 + r) * NV * 8 + e];
      atomicAdd(dst0[r] + (size_t)(e & 7) * stride_c + (size_t)(e >> 3) * stride_v, s);
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------ forward
// ST = 1: lane column = output column = input column.     y[r][x] = sum_k w[k] * a[r-1+ky][x-1+kx]
//         stream element i of a task = input row r0-1+i; output row r0+i-2 is produced from i >= 2.
// ST = 2: lane column = output column q, the lane owns input columns 2q, 2q+1; column 2q-1 is the left lane's 2q+1.
//         stream element i = input rows 2r, 2r+1 of step r = r0+i-1; output row r is produced from i >= 1.
template <typename T, int ST>
__global__ __launch_bounds__(256, 2) void dwr_fwd_kernel(const spb_dw_args_t a, const Geo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = ST == 1 ? 14 : 15;
  constexpr int NS = ST == 1 ? 1 : 4;                     // vectors per stream element
  constexpr int RD = RingDepth<T, NS>::v;
  constexpr int SLOT = NS * (sizeof(T) == 2 ? 1 : 2) * 1024;
  float* cfs = reinterpret_cast<float*>(smem);            // [4][CFN]
  char* rings = smem + 4 * CFN * sizeof(float);           // [4 waves][RD][SLOT]
  const int lane = threadIdx.x & 63, l16 = lane & 15, row = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = a.C, H = a.H, W = a.W, ncg = C >> 3;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  // A wave reads / writes 64 contiguous bytes per pixel (4 channel groups) = HALF a 128-byte L2 line; the other half belongs
  // to the neighbouring channel quad.  With quad-major block ids the two halves were fetched by workgroups on different XCDs
  // (different L2s): rocprofv3 FETCH_SIZE / WRITE_SIZE showed 2.1x the algorithmic bytes.  Here the dispatcher's round-robin
  // (block b -> XCD b % 8) is undone so that the workgroups of one task range and all channel quads share an XCD and start
  // together: the second half of every line is an L2 hit.
  int quad, bq;
  if (g.xcd) {
    const int nquads = gridDim.x / g.nb, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    quad = j % nquads; bq = (j / nquads) * 8 + x;
  } else { quad = blockIdx.x / g.nb; bq = blockIdx.x % g.nb; }
  const int cgi = quad * 4 + row;
  const bool cgok = cgi < ncg;
  const int c0 = (cgok ? cgi : ncg - 1) * 8;
  float* cf = cfs + row * CFN;
  if (!(SPB_ABL & 128) && wave == 0 && l16 < 8) {
    const int c = c0 + l16;
    float sc, sh;
    bn_fwd_coef(a.pro, c, sc, sh);
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k * 8 + l16] = a.Wd[(size_t)c * 9 + k];
    cf[72 + l16] = sc; cf[80 + l16] = sh;
  }
  __syncthreads();
  const T* X = reinterpret_cast<const T*>(a.X);
  T* Y = reinterpret_cast<T*>(a.Y);
  const int act = a.pro.act; const float slope = a.pro.slope;
  char* ring = rings + wave * RD * SLOT;
  const unsigned ring_lds = lds_addr(ring);
  const int wq = bq * 4 + wave, nwq = g.nb * 4;
  const int ntw = wq < g.ntasks ? (g.ntasks - wq + nwq - 1) / nwq : 0;
  const int N = (SPB_ABL & 32) ? 0 : ntw * g.E;
  float s[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }

  auto issue = [&](const Cursor& c, int slot) {
    const unsigned sl = ring_lds + (unsigned)(slot * SLOT);
    const int q = c.sx * NP + l16 - 1;
    if (ST == 1) {
      const int y = clampi(c.sy * g.R - 1 + c.i, 0, H - 1);
      dma_vec<T>(X + ((size_t)(c.b * H + y) * W + clampi(q, 0, W - 1)) * C + c0, sl, 0);
    } else {
      const int r = c.sy * g.R + c.i - 1;
      const int ya = clampi(2 * r, 0, H - 1), yb = clampi(2 * r + 1, 0, H - 1);
      const int xa = clampi(2 * q, 0, W - 1), xb = clampi(2 * q + 1, 0, W - 1);
      dma_vec<T>(X + ((size_t)(c.b * H + ya) * W + xa) * C + c0, sl, 0);
      dma_vec<T>(X + ((size_t)(c.b * H + ya) * W + xb) * C + c0, sl, 1);
      dma_vec<T>(X + ((size_t)(c.b * H + yb) * W + xa) * C + c0, sl, 2);
      dma_vec<T>(X + ((size_t)(c.b * H + yb) * W + xb) * C + c0, sl, 3);
    }
  };
  auto tf = [&](const Raw8<T>& r, bool ok, float o[8]) {
    float v[8], sc[8], sh[8];
    cvt8(r, v); ld_lds8(cf + 72, sc); ld_lds8(cf + 80, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ok ? act_fwd(v[j] * sc[j] + sh[j], act, slope) : 0.f;
  };

  Cursor pc, cc;
  pc.k = wq; pc.i = 0; decode_task(pc, g);
  cc = pc;
  int pe = 0;
  for (; pe < RD - 1 && pe < N; ++pe) { issue(pc, pe % RD); advance(pc, g, nwq); }
  float wa[8], wb[8], xa_[8], xb_[8];  // sliding window: ST1 rows (y-2: wa, y-1: wb); ST2 row 2r-1 cols a,b (xa_, xb_)
#pragma unroll
  for (int j = 0; j < 8; ++j) { wa[j] = 0.f; wb[j] = 0.f; xa_[j] = 0.f; xb_[j] = 0.f; }
  for (int e = 0; e < N; ++e) {
    if (pe < N) { issue(pc, pe % RD); advance(pc, g, nwq); ++pe; }
    if (N - 1 - e >= RD - 1) wait_vmcnt<NS * (sizeof(T) == 2 ? 1 : 2) * (RD - 1)>();
    else wait_vmcnt<0>();
    const char* slot = ring + (e % RD) * SLOT;
    const int q = cc.sx * NP + l16 - 1;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int orow;
    bool have;
    if constexpr (ST == 1) {
      const int y = cc.sy * g.R - 1 + cc.i;
      float xp[8];
      tf(ring_vec<T>(slot, 0, lane), q >= 0 && q < W && y >= 0 && y < H, xp);
      orow = y - 1; have = cc.i >= 2;
      if (have) {
#define DWR_ROW(ky, XR)                                                                    \
        {                                                                                  \
          float w0[8], w1[8], w2[8];                                                       \
          ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);      \
          ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                            \
          _Pragma("unroll") for (int j = 0; j < 8; ++j)                                    \
            acc[j] += from_left(XR[j]) * w0[j] + XR[j] * w1[j] + from_right(XR[j]) * w2[j]; \
        }
        DWR_ROW(0, wa) DWR_ROW(1, wb) DWR_ROW(2, xp)
#undef DWR_ROW
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { wa[j] = wb[j]; wb[j] = xp[j]; }
    } else {
      const int r = cc.sy * g.R + cc.i - 1;
      const bool oka = 2 * q >= 0 && 2 * q < W, okb = 2 * q + 1 >= 0 && 2 * q + 1 < W;
      const bool ye = 2 * r >= 0 && 2 * r < H, yo = 2 * r + 1 >= 0 && 2 * r + 1 < H;
      float ea[8], eb[8], oa[8], ob[8];
      tf(ring_vec<T>(slot, 0, lane), oka && ye, ea); tf(ring_vec<T>(slot, 1, lane), okb && ye, eb);
      tf(ring_vec<T>(slot, 2, lane), oka && yo, oa); tf(ring_vec<T>(slot, 3, lane), okb && yo, ob);
      orow = r; have = cc.i >= 1;
      if (have) {
#define DWR_ROW(ky, XA, XB)                                                                \
        {                                                                                  \
          float w0[8], w1[8], w2[8];                                                       \
          ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);      \
          ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                            \
          _Pragma("unroll") for (int j = 0; j < 8; ++j)                                    \
            acc[j] += from_left(XB[j]) * w0[j] + XA[j] * w1[j] + XB[j] * w2[j];            \
        }
        DWR_ROW(0, xa_, xb_) DWR_ROW(1, ea, eb) DWR_ROW(2, oa, ob)
#undef DWR_ROW
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { xa_[j] = oa[j]; xb_[j] = ob[j]; }
    }
    if (have) {
      rnd8<T>(acc);
      if (cgok && l16 >= 1 && l16 <= NP && q < OW && orow < OH) {
        st8<T>(Y + ((size_t)(cc.b * OH + orow) * OW + q) * C + c0, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[0][j] += acc[j]; s[1][j] += acc[j] * acc[j]; }
      }
    }
    advance(cc, g, nwq);
  }
  if (!(SPB_ABL & 64) && a.epi_mode == 1) {
    __syncthreads();  // rings are idle: reuse as reduction scratch
    bool rowok[4]; float* dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rowok[r] = quad * 4 + r < ncg;
      dst[r] = a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C + (size_t)min(quad * 4 + r, ncg - 1) * 8;
    }
    push_block<2>(reinterpret_cast<float*>(rings), s, wave, lane, rowok, dst, 1, C);
  }
}

#ifndef SPB_DW_HOIST
#define SPB_DW_HOIST 1
#endif
// ------------------------------------------------------------------------------------------------------ backward
// dz[q] = g[q]*p0 + z[q]*p1 + p2 (BN backward of the conv output, rebuilt on the fly; 0 outside the image)
// dA[p] = sum_k dz[(p + 1 - k) / ST] * w[k]   (terms with non-integer index absent)
// dW[k] += dz[q] * a[p] for the same (p, k, q) triples, a = act(bn(z_in)) of the conv input       (WG)
// EPI: dA *= act'(u_in), + residual gradient, rounded, and its BN-backward sums  sum g, sum g*xhat.
// ST = 1: lane column = input column = dz column.  Stream element i = dz row r0-1+i (+ the conv-input row r0+i-2);
//         input row r0+i-2 is produced from i >= 2.
// ST = 2: lane column = dz column q; element i = dz row r0+i (+ conv-input rows 2r, 2r+1, columns 2q, 2q+1 of step
//         r = r0+i-1); from i >= 1 the lane produces the input pixels (2r | 2r+1, 2q | 2q+1) -- exactly the 9
//         (tap, pixel) products of step r, none masked away.
// DG = false: weight gradient only (no input-gradient taps, nothing stored): the instance the KRN plan runs on its side stream
// for the 14x14 / 7x7 maps, where the fused kernel is bound by VALU issue (18 FMAs per element instead of 9) on the launch
// stream's critical path while most of the chip idles.
// XP (round 6, spb_dw_args_t::Xe; 16-bit storage): the conv-input tensor z_in is the output of a 1x1 expand convolution that is not in
// memory.  A wave's four DPP rows are the four channel octets of the SAME 16 pixels -- exactly the C layout of a 16x16x32 MFMA whose B
// operand is "pixel l16, input channels 8 row .. 8 row + 7" and whose A rows are the quad's 32 expand-weight rows, permuted as in the fused
// pointwise backward (pw_bwd_fused.hip, RZ): two MFMAs hand every lane the 8 consecutive channels of its pixel as exact f32.  So the ring
// carries the 16 / 24-channel expand input instead of z_in (one 16-byte vector per lane, as before -- a sixth of the HBM bytes, shared by
// all quads), and the activation mask / sum g*xhat (EPI) and the weight gradient's a = act(bn(z_in)) (WG) work on the recomputed value.
template <typename T, int ST, bool WG, bool EPI, bool DG = true, bool XP = false>
__global__ __launch_bounds__(256, 2) void dwr_bwd_kernel(const spb_dw_args_t a, const Geo g) {
  spb_publish_entry(a.entry_flag, a.entry_val);
  static_assert(DG || (WG && !EPI), "the weight-gradient-only instance has no epilogue");
  static_assert(!XP || ((WG || EPI) && sizeof(T) == 2), "expand recompute: 16-bit storage, and only where the conv input is needed");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = ST == 1 ? 14 : 15;
  constexpr int LH = ST == 1 ? 1 : 0;                    // stride 2 only needs the right neighbour
  constexpr bool IN = WG || EPI;                         // the conv-input tensor is read
  constexpr int NS = 2 + (IN ? (ST == 1 ? 1 : 4) : 0);
  constexpr int RD = RingDepth<T, NS>::v;
  constexpr int SLOT = NS * (sizeof(T) == 2 ? 1 : 2) * 1024;
  float* cfs = reinterpret_cast<float*>(smem);
  float* xtab = cfs + 4 * CFN;                           // XP: scale | shift of the Ce <= 32 expand-input channels
  char* rings = smem + 4 * CFN * sizeof(float) + (XP ? 64 * sizeof(float) : 0);
  const int lane = threadIdx.x & 63, l16 = lane & 15, row = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int C = a.C, H = a.H, W = a.W, ncg = C >> 3;
  const int OH = (H - 1) / ST + 1, OW = (W - 1) / ST + 1;
  // A wave reads / writes 64 contiguous bytes per pixel (4 channel groups) = HALF a 128-byte L2 line; the other half belongs
  // to the neighbouring channel quad.  With quad-major block ids the two halves were fetched by workgroups on different XCDs
  // (different L2s): rocprofv3 FETCH_SIZE / WRITE_SIZE showed 2.1x the algorithmic bytes.  Here the dispatcher's round-robin
  // (block b -> XCD b % 8) is undone so that the workgroups of one task range and all channel quads share an XCD and start
  // together: the second half of every line is an L2 hit.
  int quad, bq;
  if (g.xcd) {
    const int nquads = gridDim.x / g.nb, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    quad = j % nquads; bq = (j / nquads) * 8 + x;
  } else { quad = blockIdx.x / g.nb; bq = blockIdx.x % g.nb; }
  const int cgi = quad * 4 + row;
  const bool cgok = cgi < ncg;
  const int c0 = (cgok ? cgi : ncg - 1) * 8;
  float* cf = cfs + row * CFN;
  if (!(SPB_ABL & 128) && wave == 0 && l16 < 8) {
    const int c = c0 + l16;
    float p0, p1, p2, sc, sh, mu, is;
    bn_bwd_epi_coef(a.pro, a.epi, IN && a.epi.gamma != nullptr, c, p0, p1, p2, sc, sh, mu, is);
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k * 8 + l16] = a.Wd[(size_t)c * 9 + k];
    cf[72 + l16] = p0; cf[80 + l16] = p1; cf[88 + l16] = p2;
    cf[96 + l16] = sc; cf[104 + l16] = sh; cf[112 + l16] = mu; cf[120 + l16] = is;
  }
  const int CX = XP ? a.Ce : C;                          // channels of the tensor behind Zo
  const bool xbn = XP && a.xe.gamma != nullptr;          // (uniform)
  const float xhi = act_hi(a.xe.act), xns = act_ns(a.xe.act, a.xe.slope);
  const bool xact = a.xe.act != SPB_ACT_NONE;
  uint4 wz0 = make_uint4(0, 0, 0, 0), wz1 = make_uint4(0, 0, 0, 0);
  const int xk0 = 8 * row < CX ? 8 * row : CX - 8;       // this lane's 8 input channels (clamped: the weight fragment is zero past Ce)
  if constexpr (XP) {
    // A operands: fragment t, row l16  <->  expand-weight row (= channel of z_in) quad * 32 + (l16 >> 2) * 8 + t * 4 + (l16 & 3), input channels
    // 8 row .. 8 row + 7.  The C layout then gives lane (l16, row) channels quad * 32 + 8 row + t * 4 + i: with t = 0, 1 its own octet.
    const bf16_t* We = reinterpret_cast<const bf16_t*>(a.We);
    const int n0 = quad * 32 + (l16 >> 2) * 8 + (l16 & 3), n1 = n0 + 4;
    wz0 = *reinterpret_cast<const uint4*>(We + (size_t)(n0 < C ? n0 : C - 1) * CX + xk0);
    wz1 = *reinterpret_cast<const uint4*>(We + (size_t)(n1 < C ? n1 : C - 1) * CX + xk0);
    if (n0 >= C || 8 * row >= CX) wz0 = make_uint4(0, 0, 0, 0);
    if (n1 >= C || 8 * row >= CX) wz1 = make_uint4(0, 0, 0, 0);
    if (wave == 1 && lane < 32) {                        // (wave 0 builds the coefficient rows above)
      float s_ = 1.f, h_ = 0.f;
      if (xbn) bn_fwd_coef(a.xe, lane < CX ? lane : CX - 1, s_, h_);
      xtab[lane] = s_; xtab[32 + lane] = h_;
    }
  }
  __syncthreads();
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Zo = reinterpret_cast<const T*>(XP ? a.Xe : a.Zout);
  const int zc0 = XP ? xk0 : c0;                         // channel offset of this lane's vector in the tensor behind Zo
  const T* Rg = reinterpret_cast<const T*>(a.res);
  T* Y = reinterpret_cast<T*>(a.Y);
  const int eact = a.epi.act; const float eslope = a.epi.slope;
  // HOIST (input-gradient-only instances): the loop-invariant coefficient vectors of the lane's 8 channels -- 9 tap weights, the three
  // BN-backward coefficients of dz, scale / shift of the input-side BN -- live in registers for the whole kernel instead of being read
  // back from LDS for every element (30 ds_read_b128 per element behind the ring's memory clobbers; the instance runs two waves per
  // SIMD, 256 registers each: 112 more fit)
  // (the weight-gradient-only instance of the side stream reads no tap weights: it keeps the 40 coefficient values; the fused instance,
  // at 236-256 registers already, keeps nothing)
  constexpr bool HOIST = SPB_DW_HOIST && DG && !WG;         // tap weights
  constexpr bool HOISTP = SPB_DW_HOIST && !(DG && WG);      // BN-backward coefficients of dz, scale / shift of the input side
  float wr[HOIST ? 72 : 1], pr[HOISTP ? 24 : 1], er[HOISTP ? 16 : 1];
  if constexpr (HOIST) {
#pragma unroll
    for (int k = 0; k < 9; ++k) ld_lds8(cf + k * 8, wr + k * 8);
  }
  if constexpr (HOISTP) {
    ld_lds8(cf + 72, pr); ld_lds8(cf + 80, pr + 8); ld_lds8(cf + 88, pr + 16);
    ld_lds8(cf + 96, er); ld_lds8(cf + 104, er + 8);
  }
  char* ring = rings + wave * RD * SLOT;
  const unsigned ring_lds = lds_addr(ring);
  const int wq = bq * 4 + wave, nwq = g.nb * 4;
  const int ntw = wq < g.ntasks ? (g.ntasks - wq + nwq - 1) / nwq : 0;
  const int N = (SPB_ABL & 32) ? 0 : ntw * g.E;
  const bool lane_prod = cgok && l16 >= LH && l16 < LH + NP;
  float s[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }
  float aw[WG ? 9 : 1][8];
#pragma unroll
  for (int k = 0; k < (WG ? 9 : 1); ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[k][j] = 0.f;

  auto issue = [&](const Cursor& c, int slot) {
    const unsigned sl = ring_lds + (unsigned)(slot * SLOT);
    const int q = c.sx * NP + l16 - LH;
    const int qc = clampi(q, 0, OW - 1);
    if (ST == 1) {
      const int y = c.sy * g.R - 1 + c.i;
      const size_t go = ((size_t)(c.b * OH + clampi(y, 0, OH - 1)) * OW + qc) * C + c0;
      dma_vec<T>(G + go, sl, 0); dma_vec<T>(Z + go, sl, 1);
      if (IN) dma_vec<T>(Zo + ((size_t)(c.b * H + clampi(y - 1, 0, H - 1)) * W + qc) * CX + zc0, sl, 2);
    } else {
      const int y = c.sy * g.R + c.i, r = y - 1;
      const size_t go = ((size_t)(c.b * OH + clampi(y, 0, OH - 1)) * OW + qc) * C + c0;
      dma_vec<T>(G + go, sl, 0); dma_vec<T>(Z + go, sl, 1);
      if (IN) {
        const int ya = clampi(2 * r, 0, H - 1), yb = clampi(2 * r + 1, 0, H - 1);
        const int xa = clampi(2 * q, 0, W - 1), xb = clampi(2 * q + 1, 0, W - 1);
        dma_vec<T>(Zo + ((size_t)(c.b * H + ya) * W + xa) * CX + zc0, sl, 2);
        dma_vec<T>(Zo + ((size_t)(c.b * H + ya) * W + xb) * CX + zc0, sl, 3);
        dma_vec<T>(Zo + ((size_t)(c.b * H + yb) * W + xa) * CX + zc0, sl, 4);
        dma_vec<T>(Zo + ((size_t)(c.b * H + yb) * W + xb) * CX + zc0, sl, 5);
      }
    }
  };
  auto dzf = [&](const Raw8<T>& gr, const Raw8<T>& zr, bool ok, float o[8]) {
    float gf[8], zf[8], p0[8], p1[8], p2[8];
    cvt8(gr, gf); cvt8(zr, zf);
    if constexpr (HOISTP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { p0[j] = pr[j]; p1[j] = pr[HOISTP ? 8 + j : 0]; p2[j] = pr[HOISTP ? 16 + j : 0]; }
    } else { ld_lds8(cf + 72, p0); ld_lds8(cf + 80, p1); ld_lds8(cf + 88, p2); }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ok ? gf[j] * p0[j] + zf[j] * p1[j] + p2[j] : 0.f;
  };
  // z_in of this lane's pixel and channel octet: the ring's vector as it is, or (XP) W x on the matrix cores.  Called from wave-uniform
  // control flow only (all 64 lanes feed the matrix core).
  auto zin = [&](const char* slot, int piece, float zf[8]) __attribute__((always_inline)) {
    if constexpr (XP) {
      Raw8<bf16_t> xr;
      xr.u = *reinterpret_cast<const uint4*>(slot + (piece << 10) + lane * 16);
      if (xbn) {                                         // the expand convolution's operand: round16(act(bn(x)))
        float xv[8], xs[8], xh[8];
        cvt8(xr, xv);
        ld_lds8(xtab + xk0, xs); ld_lds8(xtab + 32 + xk0, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xv[j] * xs[j] + xh[j];
        if (xact) {                                        // (uniform; MobileNetV2's block inputs are linear)
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[j] = __builtin_amdgcn_fmed3f(xv[j], 0.f, xhi) + xns * fminf(xv[j], 0.f);
        }
        xr.u.x = pack_bf16x2(xv[0], xv[1]); xr.u.y = pack_bf16x2(xv[2], xv[3]); xr.u.z = pack_bf16x2(xv[4], xv[5]); xr.u.w = pack_bf16x2(xv[6], xv[7]);
      }
      const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xr.u);
      const f32x4_t z0 = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wz0), xb, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
      const f32x4_t z1 = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wz1), xb, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
      for (int i = 0; i < 4; ++i) { zf[i] = z0[i]; zf[4 + i] = z1[i]; }
    } else cvt8(ring_vec<T>(slot, piece, lane), zf);
  };
  // a[p] of the conv input from its raw z (WG), 0 for lanes/pixels that do not exist
  auto apf = [&](const float zf[8], bool ok, float o[8]) {
    float sc[8], sh[8];
    if constexpr (HOISTP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = er[j]; sh[j] = er[HOISTP ? 8 + j : 0]; }
    } else { ld_lds8(cf + 96, sc); ld_lds8(cf + 104, sh); }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ok ? act_fwd(zf[j] * sc[j] + sh[j], eact, eslope) : 0.f;
  };
  // finish one input pixel: residual, activation mask, rounding, BN-backward sums, store.  The second sum is kept as
  // sum g*z and turned into sum g*xhat = invstd * (sum g*z - mean * sum g) once per block (two coefficient vectors
  // less in the loop; the subtraction costs log2(|mean|/std) bits of the f32 partial, a few at most).
  auto fin = [&](float acc[8], const float zf[8], bool ok, size_t off) {
    if (EPI) {
      float sc[8], sh[8];
      if constexpr (HOISTP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = er[j]; sh[j] = er[HOISTP ? 8 + j : 0]; }
      } else { ld_lds8(cf + 96, sc); ld_lds8(cf + 104, sh); }
      if (Rg) {
        float rf[8];
        cvt8(ldraw<T>(Rg + off), rf);
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
    if (ok && !(SPB_ABL & 1024)) st8<T>(Y + off, acc);
  };

  Cursor pc, cc;
  pc.k = wq; pc.i = 0; decode_task(pc, g);
  cc = pc;
  int pe = 0;
  for (; pe < RD - 1 && pe < N; ++pe) { issue(pc, pe % RD); advance(pc, g, nwq); }
  float dm[8], d0[8];   // ST1: dz rows y-2, y-1;  ST2: d0 = dz row r (dm unused)
#pragma unroll
  for (int j = 0; j < 8; ++j) { dm[j] = 0.f; d0[j] = 0.f; }
  for (int e = 0; e < N; ++e) {
    if (pe < N) { if (!(SPB_ABL & 512)) issue(pc, pe % RD); advance(pc, g, nwq); ++pe; }
    if (N - 1 - e >= RD - 1) wait_vmcnt<NS * (sizeof(T) == 2 ? 1 : 2) * (RD - 1)>();
    else wait_vmcnt<0>();
    if (SPB_ABL & 256) { advance(cc, g, nwq); continue; }
    const char* slot = ring + (e % RD) * SLOT;
    const int q = cc.sx * NP + l16 - LH;
    const bool qok = q >= 0 && q < OW;
    if constexpr (ST == 1) {
      const int y = cc.sy * g.R - 1 + cc.i;      // dz row of this element
      float dp[8];
      dzf(ring_vec<T>(slot, 0, lane), ring_vec<T>(slot, 1, lane), qok && y >= 0 && y < OH, dp);
      if (cc.i >= 2) {
        const int r = y - 1;                     // input row produced now
        const bool ok = lane_prod && qok && r < H;
        float zf[8], acc[8], ap[8];
        if (IN) zin(slot, 2, zf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (WG) apf(zf, ok, ap);
        // taps: input pixel (r, q) <- dz(r + 1 - ky, q + 1 - kx)
#define DWR_ROW(ky, DR)                                                                    \
        {                                                                                  \
          if (WG) asm volatile("" ::: "memory"); /* one weight row in flight at a time */  \
          float w0[8], w1[8], w2[8];                                                       \
          if constexpr (HOIST) {                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                \
              w0[j] = wr[HOIST ? ((ky) * 3 + 0) * 8 + j : 0]; w1[j] = wr[HOIST ? ((ky) * 3 + 1) * 8 + j : 0]; \
              w2[j] = wr[HOIST ? ((ky) * 3 + 2) * 8 + j : 0];                              \
            }                                                                              \
          } else {                                                                         \
            ld_lds8(cf + ((ky) * 3 + 0) * 8, w0); ld_lds8(cf + ((ky) * 3 + 1) * 8, w1);    \
            ld_lds8(cf + ((ky) * 3 + 2) * 8, w2);                                          \
          }                                                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                  \
            const float vr = from_right(DR[j]), vl = from_left(DR[j]);                     \
            if (DG) acc[j] += vr * w0[j] + DR[j] * w1[j] + vl * w2[j];                     \
            if (WG) {                                                                      \
              aw[WG ? (ky) * 3 + 0 : 0][j] += vr * ap[j];                                  \
              aw[WG ? (ky) * 3 + 1 : 0][j] += DR[j] * ap[j];                               \
              aw[WG ? (ky) * 3 + 2 : 0][j] += vl * ap[j];                                  \
            }                                                                              \
          }                                                                                \
        }
        DWR_ROW(0, dp) DWR_ROW(1, d0) DWR_ROW(2, dm)
#undef DWR_ROW
        if (DG) fin(acc, zf, ok, ((size_t)(cc.b * H + clampi(r, 0, H - 1)) * W + clampi(q, 0, W - 1)) * C + c0);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { dm[j] = d0[j]; d0[j] = dp[j]; }
    } else {
      const int y = cc.sy * g.R + cc.i;          // dz row of this element = row r+1 of step r
      float d1[8];
      dzf(ring_vec<T>(slot, 0, lane), ring_vec<T>(slot, 1, lane), qok && y < OH, d1);
      if (cc.i >= 1) {
        const int r = y - 1;
        const int ca = 2 * q, cb = 2 * q + 1;
        const bool oka = lane_prod && ca < W && 2 * r < H, okb = lane_prod && cb < W && 2 * r < H;
        const bool row1 = 2 * r + 1 < H;
        const size_t base = (size_t)cc.b * H * W;
        const size_t pa0 = ((base + (size_t)clampi(2 * r, 0, H - 1) * W + clampi(ca, 0, W - 1)) * C) + c0;
        const size_t pb0 = ((base + (size_t)clampi(2 * r, 0, H - 1) * W + clampi(cb, 0, W - 1)) * C) + c0;
        const size_t pa1 = ((base + (size_t)clampi(2 * r + 1, 0, H - 1) * W + clampi(ca, 0, W - 1)) * C) + c0;
        const size_t pb1 = ((base + (size_t)clampi(2 * r + 1, 0, H - 1) * W + clampi(cb, 0, W - 1)) * C) + c0;
        // one input pixel at a time (keeps acc/ap at 16 registers)
#define DWR_PIX(PI, OKP, OFF, BODY)                                                        \
        {                                                                                  \
          if (WG || EPI) asm volatile("" ::: "memory");                                    \
          const bool ok = (OKP);                                                           \
          float zf[8], ap[8], acc[8];                                                      \
          if (IN) zin(slot, 2 + (PI), zf);                                                 \
          if (WG) apf(zf, ok, ap);                                                         \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = 0.f;                      \
          BODY                                                                             \
          if (DG) fin(acc, zf, ok, (OFF));                                                 \
        }
#define DWR_TAP(k, V)                                                                      \
        {                                                                                  \
          float w[8];                                                                      \
          if constexpr (HOIST) {                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) w[j] = wr[HOIST ? (k) * 8 + j : 0]; \
          } else ld_lds8(cf + (k) * 8, w);                                                 \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                  \
            const float v = (V);                                                           \
            if (DG) acc[j] += v * w[j];                                                    \
            if (WG) aw[WG ? (k) : 0][j] += v * ap[j];                                      \
          }                                                                                \
        }
        DWR_PIX(0, oka, pa0, DWR_TAP(4, d0[j]))
        DWR_PIX(1, okb, pb0, DWR_TAP(3, from_right(d0[j])) DWR_TAP(5, d0[j]))
        DWR_PIX(2, oka && row1, pa1, DWR_TAP(1, d1[j]) DWR_TAP(7, d0[j]))
        DWR_PIX(3, okb && row1, pb1,
                DWR_TAP(0, from_right(d1[j])) DWR_TAP(2, d1[j]) DWR_TAP(6, from_right(d0[j])) DWR_TAP(8, d0[j]))
#undef DWR_PIX
#undef DWR_TAP
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) d0[j] = d1[j];
    }
    advance(cc, g, nwq);
  }
  if (EPI) {
    float mu[8], is[8];
    ld_lds8(cf + 112, mu); ld_lds8(cf + 120, is);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[1][j] = is[j] * (s[1][j] - mu[j] * s[0][j]);
  }
  __syncthreads();  // rings are idle: reuse as reduction scratch
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rowok[r] = quad * 4 + r < ncg;
  if (!(SPB_ABL & 64) && EPI) {
    float* dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = a.osums + (size_t)(blockIdx.x % a.oR) * 2 * C + (size_t)min(quad * 4 + r, ncg - 1) * 8;
    push_block<2>(reinterpret_cast<float*>(rings), s, wave, lane, rowok, dst, 1, C);
  }
  if constexpr (WG && !(SPB_ABL & 64)) {
    float* dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = a.dW + (size_t)min(quad * 4 + r, ncg - 1) * 8 * 9;
    push_block<9>(reinterpret_cast<float*>(rings), aw, wave, lane, rowok, dst, 9, 1);
  }
}

// Task shape.  ~2 blocks per CU; rows per task R as long as possible (2/R of the rows are re-read as halo, 1/R for the
// stride-2 backward) while the tasks still split evenly over the waves of a channel quad.
int g_rows_override = 0;
int g_xcd_pair = 1;
int g_blocks_override = 0;
Geo make_geo(int B, int C, int lane_rows, int lane_cols, int NP, int halo, int target_blocks) {
  Geo g;
  const int ncg = C >> 3, nquads = (ncg + 3) / 4;
  g.nstrip = (lane_cols + NP - 1) / NP;
  if (g_blocks_override > 0) target_blocks = g_blocks_override;
  g.nb = target_blocks / nquads;   // never more blocks than are resident at once: a partial second round doubles small layers
  const long long per_seg = (long long)B * g.nstrip;
  if ((long long)g.nb * 4 > per_seg * lane_rows) g.nb = (int)((per_seg * lane_rows + 3) / 4);
  if (g.nb < 1) g.nb = 1;
  g.xcd = 0;
  if (g_xcd_pair && g.nb >= 8 && nquads > 1) { g.nb &= ~7; g.xcd = 1; }
  const int nw = g.nb * 4;
  int bestR = lane_rows; double best = 1e30;
  for (int R = 1; R <= lane_rows; ++R) {
    const int nseg = (lane_rows + R - 1) / R;
    const long long nt = per_seg * nseg;
    const long long rounds = (nt + nw - 1) / nw;
    const double cost = (double)rounds * (R + halo) + 0.25 * rounds;   // per-task switch overhead ~ a quarter row
    if (cost < best * 0.999) { best = cost; bestR = R; }
  }
  if (g_rows_override > 0) bestR = g_rows_override < lane_rows ? g_rows_override : lane_rows;
  g.R = bestR; g.E = bestR + halo;
  g.nseg = (lane_rows + bestR - 1) / bestR;
  g.ntasks = (int)(per_seg * g.nseg);
  return g;
}

int ring_depth_host(int ns, int es) {   // == RingDepth<T, NS>::v
  const int bytes = ns * es * 4096;
  return SPB_DW_RING_BUDGET / bytes >= SPB_DW_RING_CAP ? SPB_DW_RING_CAP : (SPB_DW_RING_BUDGET / bytes >= 2 ? SPB_DW_RING_BUDGET / bytes : 2);
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

// rows > 0: rows per task; rows < 0: -rows = target block count (experiments)
static int g_dw_wgrad_blocks = 512;   // workgroups of the weight-gradient-only instance (side stream); spb_debug_set_dw_wgrad_blocks
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_wgrad_blocks(int n) { if (n > 0) g_dw_wgrad_blocks = n; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_xcd(int on) { g_xcd_pair = on; return 0; }
#endif
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_rows(int rows) { if (rows >= 0) g_rows_override = rows; else g_blocks_override = -rows; return 0; }
#endif

int spb_dwr_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const Geo g = make_geo(a->B, a->C, OH, OW, st == 1 ? 14 : 15, st == 1 ? 2 : 1, st == 1 ? 1024 : 768);
  const int nquads = ((a->C >> 3) + 3) / 4;
  const dim3 grid((unsigned)(nquads * g.nb));
  const int es = dtype == SPB_BF16 ? 1 : 2;
  const int rd = ring_depth_host(st == 1 ? 1 : 4, es);
  size_t lds = 4 * CFN * sizeof(float) + (size_t)4 * rd * (st == 1 ? 1 : 4) * es * 1024;
  const size_t red = (size_t)16 * 16 * sizeof(float) + 4 * CFN * sizeof(float);
  if (lds < red) lds = red;
#define L_(T_, ST_)                                                                     \
  {                                                                                     \
    static bool once = false;                                                           \
    if (!once) { allow_lds(dwr_fwd_kernel<T_, ST_>, 160 * 1024); once = true; }         \
    hipLaunchKernelGGL((dwr_fwd_kernel<T_, ST_>), grid, dim3(256), lds, s, *a, g);      \
  }
  if (dtype == SPB_BF16) { if (st == 1) L_(bf16_t, 1) else L_(bf16_t, 2) }
  else { if (st == 1) L_(float, 1) else L_(float, 2) }
#undef L_
  return 0;
}

// weight gradient alone (a->Zout / a->epi name the convolution's input tensor and its BN + activation, a->Y is unused)
int spb_dwr_wgrad(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const Geo g = make_geo(a->B, a->C, OH, OW, st == 1 ? 14 : 15, st == 1 ? 2 : 1, g_dw_wgrad_blocks);
  const int nquads = ((a->C >> 3) + 3) / 4;
  const dim3 grid((unsigned)(nquads * g.nb));
  const int es = dtype == SPB_BF16 ? 1 : 2;
  const int ns = 2 + (st == 1 ? 1 : 4);
  const int rd = ring_depth_host(ns, es);
  const bool xp = a->Xe != nullptr;
  if (xp && dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  size_t lds = 4 * CFN * sizeof(float) + (xp ? 256 : 0) + (size_t)4 * rd * ns * es * 1024;
  const size_t red = (size_t)16 * 72 * sizeof(float) + 4 * CFN * sizeof(float) + (xp ? 256 : 0);
  if (lds < red) lds = red;
#define W_(T_, ST_, XP_)                                                                                        \
  {                                                                                                             \
    static bool once = false;                                                                                   \
    if (!once) { allow_lds(dwr_bwd_kernel<T_, ST_, true, false, false, XP_>, 160 * 1024); once = true; }        \
    hipLaunchKernelGGL((dwr_bwd_kernel<T_, ST_, true, false, false, XP_>), grid, dim3(256), lds, s, *a, g);     \
  }
  if (xp) { if (st == 1) W_(bf16_t, 1, true) else W_(bf16_t, 2, true) }
  else if (dtype == SPB_BF16) { if (st == 1) W_(bf16_t, 1, false) else W_(bf16_t, 2, false) }
  else { if (st == 1) W_(float, 1, false) else W_(float, 2, false) }
#undef W_
  return 0;
}

int spb_dwr_bwd(int dtype, const spb_dw_args_t* a, hipStream_t s) {
  const int st = a->stride;
  const int OH = (a->H - 1) / st + 1, OW = (a->W - 1) / st + 1;
  const Geo g = make_geo(a->B, a->C, OH, OW, st == 1 ? 14 : 15, st == 1 ? 2 : 1, 512);
  const int nquads = ((a->C >> 3) + 3) / 4;
  const dim3 grid((unsigned)(nquads * g.nb));
  const bool wg = a->dW != nullptr, epi = a->epi_mode == 2;
  const bool xp = a->Xe != nullptr && (wg || epi);      // (a plain input gradient never looks at the conv input)
  if (xp && dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  const int es = dtype == SPB_BF16 ? 1 : 2;
  const int ns = 2 + ((wg || epi) ? (st == 1 ? 1 : 4) : 0);
  const int rd = ring_depth_host(ns, es);
  size_t lds = 4 * CFN * sizeof(float) + (xp ? 256 : 0) + (size_t)4 * rd * ns * es * 1024;
  const size_t red = (size_t)16 * 72 * sizeof(float) + 4 * CFN * sizeof(float) + (xp ? 256 : 0);   // reduction scratch reuses the rings
  if (lds < red) lds = red;
  if (xp) {
#define X_(ST_, WG_, EPI_)                                                                                          \
  {                                                                                                                 \
    static bool once = false;                                                                                       \
    if (!once) { allow_lds(dwr_bwd_kernel<bf16_t, ST_, WG_, EPI_, true, true>, 160 * 1024); once = true; }          \
    hipLaunchKernelGGL((dwr_bwd_kernel<bf16_t, ST_, WG_, EPI_, true, true>), grid, dim3(256), lds, s, *a, g);       \
  }
    if (st == 1) { if (wg) { if (epi) X_(1, true, true) else X_(1, true, false) } else X_(1, false, true) }
    else { if (wg) { if (epi) X_(2, true, true) else X_(2, true, false) } else X_(2, false, true) }
#undef X_
    return 0;
  }
#define L_(T_, ST_, WG_, EPI_)                                                                   \
  {                                                                                              \
    static bool once = false;                                                                    \
    if (!once) { allow_lds(dwr_bwd_kernel<T_, ST_, WG_, EPI_>, 160 * 1024); once = true; }       \
    hipLaunchKernelGGL((dwr_bwd_kernel<T_, ST_, WG_, EPI_>), grid, dim3(256), lds, s, *a, g);    \
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

// ---- C-ABI of the depthwise convolution (include/spb_hip.h): shape checks and the choice between the two kernel
// families.  Small maps (width <= 28: the 28x28 / 14x14 / 7x7 tensors of KRN) go to the one-round-trip plane kernels
// (dwconv_plane.hip), everything else and the stand-alone weight gradient to the row-unit kernels above.
int spb_dwp_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s);   // dwconv_plane.hip; SPB_E_UNSUPPORTED: not covered
int spb_dwt_fwd(int dtype, const spb_dw_args_t* a, hipStream_t s);   // dwconv_tile.hip (bf16, large maps); SPB_E_UNSUPPORTED: not covered
int spb_dwt_dgrad(int dtype, const spb_dw_args_t* a, hipStream_t s);
int spb_dwp_bwd(int dtype, const spb_dw_args_t* a, hipStream_t s);
static int g_dw_mode = 1;   // 1: plane kernels where they apply (default); 0: row-unit kernels only
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_dw_mode(int mode) { g_dw_mode = mode; return 0; }
#endif

static int dw_check(const spb_dw_args_t* a, bool fwd = false) {
  if (!a || !a->Wd) return SPB_E_ARG;
  if (!a->X && !(fwd && a->Xe)) return SPB_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 7)) return SPB_E_SHAPE;
  if (a->stride != 1 && a->stride != 2) return SPB_E_SHAPE;
  if (a->Xe) {   // expand recompute (spb_dw_args_t::Xe)
    if (!a->We) return SPB_E_ARG;
    if (a->Ce < 8 || a->Ce > 32 || (a->Ce & 7)) return SPB_E_SHAPE;
  }
  return 0;
}

extern "C" int spb_dwconv_fwd(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a, true);
  if (e) return e;
  if (!a->Y || (a->epi_mode == 1 && (!a->osums || a->oR < 1))) return SPB_E_ARG;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  if (a->Xe) {   // the input tensor does not exist: the tile kernel's expand-recompute instance or nothing
    if (spb_dwt_fwd(dtype, a, (hipStream_t)stream) == SPB_E_UNSUPPORTED) return SPB_E_UNSUPPORTED;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (spb_dwt_fwd(dtype, a, (hipStream_t)stream) == SPB_E_UNSUPPORTED) {
    if (g_dw_mode != 1 || spb_dwp_fwd(dtype, a, (hipStream_t)stream) == SPB_E_UNSUPPORTED)
      spb_dwr_fwd(dtype, a, (hipStream_t)stream);
  }
  SPB_CHECK_LAUNCH();
  return 0;
}

// dW != NULL fuses the weight gradient into the same pass: then Zout must be the input tensor of the convolution and
// `epi` its BN/activation (what spb_dwconv_wgrad takes as Xin / pro_in), also when epi_mode == 0.
extern "C" int spb_dwconv_dgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y) return SPB_E_ARG;
  if (a->epi_mode == 2 && (!a->osums || a->oR < 1 || (!a->Zout && !a->Xe))) return SPB_E_ARG;
  if (a->dW != nullptr && !a->Zout && !a->Xe) return SPB_E_ARG;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  spb_dw_args_t k = *a;
  if (!k.X2) k.X2 = k.X;  // no BN behind the convolution: p1 == 0, the kernel still reads a (finite) second operand
  if (k.Xe && (k.epi_mode == 2 || k.dW)) {   // the conv input is recomputed: the row-unit kernel's expand-recompute instances
    if (spb_dwr_bwd(dtype, &k, (hipStream_t)stream) == SPB_E_UNSUPPORTED) return SPB_E_UNSUPPORTED;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (spb_dwt_dgrad(dtype, &k, (hipStream_t)stream) == SPB_E_UNSUPPORTED) {
    if (g_dw_mode != 1 || spb_dwp_bwd(dtype, &k, (hipStream_t)stream) == SPB_E_UNSUPPORTED)
      spb_dwr_bwd(dtype, &k, (hipStream_t)stream);
  }
  SPB_CHECK_LAUNCH();
  return 0;
}

// weight gradient alone (row-unit kernel, weight-gradient-only instance: the input tensor travels as Zout / epi there)
extern "C" int spb_dwconv_wgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->dW || (!a->Xin && !a->Xe)) return SPB_E_ARG;
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  spb_dw_args_t k = *a;
  k.Zout = a->Xin; k.epi = a->pro_in; k.Y = nullptr; k.epi_mode = 0; k.res = nullptr; k.entry_flag = nullptr;
  if (!k.X2) k.X2 = k.X;
  if (spb_dwr_wgrad(dtype, &k, (hipStream_t)stream) == SPB_E_UNSUPPORTED) return SPB_E_UNSUPPORTED;
  SPB_CHECK_LAUNCH();
  return 0;
}
