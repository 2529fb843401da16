// The two dense convolutions of KRN that are not 1x1/depthwise:
//   stem  Conv2d(3,32,3,stride 2,pad 1,bias=False) on the NCHW f32 image  (torchvision mobilenet_v2.features[0],
//         reached through reference park2019.py:107-108)
//   head  Conv2d(1024,2K,kernel 7) on the 7x7 map, i.e. one fully-connected layer over (h,w,c), followed by the
//         (x,y) de-interleave and the summed per-keypoint MSE (park2019.py:121,139-162).
#include "common.h"
#include <type_traits>

#ifndef SPB_TS_DECL       // register-held phase timestamps (scratch/ubench_head.hip); compiled out in the product build
#define SPB_TS_DECL
#define SPB_TSR(i)
#define SPB_TS_FLUSH
#endif

namespace {

// ------------------------------------------------------------------------------------------------ stem forward
// thread = (output pixel, group of 8 output channels); weights live in LDS as [tap][co] so the 4 channel groups
// of a pixel read 4 distinct 32-byte rows.  Output is written as NHWC 16-byte vectors (64 B contiguous per pixel).
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       T* __restrict__ y, float* osums, int oR, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float wl[27 * 32];
  __shared__ float red[64];
  const int t = threadIdx.x;
  for (int i = t; i < 27 * 32; i += 256) {
    const int tap = i >> 5, co = i & 31;
    wl[i] = w[co * 27 + tap];
  }
  if (t < 64) red[t] = 0.f;
  __syncthreads();
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long P = (long long)B * OH * OW;
  const int cg = t & 3, pl = t >> 2;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  for (long long p = (long long)blockIdx.x * 64 + pl; p < P; p += (long long)gridDim.x * 64) {
    // the 27x8 weights this thread uses are loop invariant; without this barrier the compiler keeps all 216 of them in
    // VGPRs (282 registers, 1 wave per SIMD, 280 GB/s).  Re-reading them from LDS per pixel costs 54 ds_read_b128.
    asm volatile("" ::: "memory");
    const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), b = (int)(p / ((long long)OW * OH));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci)  // not unrolled: 9 taps (72 weights) live at a time instead of 27 (216)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ih = oh * 2 - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iw = ow * 2 - 1 + kx;
          const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
          float xv = x[((size_t)(b * 3 + ci) * H + ihc) * W + iwc];  // clamped address: no branch around the load
          xv = (ih == ihc && iw == iwc) ? xv : 0.f;
          const float* wr = wl + (ci * 9 + ky * 3 + kx) * 32 + cg * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += xv * wr[j];
        }
      }
    rnd8<T>(acc);
    st8<T>(y + (size_t)p * 32 + cg * 8, acc);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
  }
  if (osums) {
#ifdef SPB_DET   // reproducible twin (common.h): no LDS float atomics; every thread adds exactly
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + cg * 8 + j, s1[j]);
      atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + 32 + cg * 8 + j, s2[j]);
    }
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&red[cg * 8 + j], s1[j]);
      atomicAdd(&red[32 + cg * 8 + j], s2[j]);
    }
    __syncthreads();
    if (t < 64) atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + t, red[t]);
#endif
  }
}

// dW[co,ci,ky,kx] += sum_pixels dz[pixel,co] * x[pixel -> (ci,ky,kx)];  thread = (pixel, 8 co, ci) -> 72 partials
template <typename T>
__global__ __launch_bounds__(192) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ G,
                                                         const T* __restrict__ Z, const spb_bnref_t pro, float* dW,
                                                         int B, int H, int W) {
  __shared__ float red[32 * 27];
  const int t = threadIdx.x;
  for (int i = t; i < 32 * 27; i += 192) red[i] = 0.f;
  __syncthreads();
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long P = (long long)B * OH * OW;
  const int sub = t % 12, pl = t / 12;
  const int cg = sub & 3, ci = sub >> 2;
  float p0[8], p1[8], p2[8], aw[9][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bn_bwd_coef(pro, cg * 8 + j, p0[j], p1[j], p2[j]);
#pragma unroll
    for (int k = 0; k < 9; ++k) aw[k][j] = 0.f;
  }
  for (long long p = (long long)blockIdx.x * 16 + pl; p < P; p += (long long)gridDim.x * 16) {
    const int ow = (int)(p % OW), oh = (int)((p / OW) % OH), b = (int)(p / ((long long)OW * OH));
    float g[8], z[8], dz[8];
    ld8<T>(G + (size_t)p * 32 + cg * 8, g);
    ld8<T>(Z + (size_t)p * 32 + cg * 8, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) dz[j] = g[j] * p0[j] + z[j] * p1[j] + p2[j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ih = oh * 2 - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iw = ow * 2 - 1 + kx;
        const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        float xv = x[((size_t)(b * 3 + ci) * H + ihc) * W + iwc];
        xv = (ih == ihc && iw == iwc) ? xv : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) aw[ky * 3 + kx][j] += xv * dz[j];
      }
    }
  }
#ifdef SPB_DET   // reproducible twin (common.h): no LDS float atomics; every thread adds exactly
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(dW + (cg * 8 + j) * 27 + ci * 9 + k, aw[k][j]);
#else
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&red[(cg * 8 + j) * 27 + ci * 9 + k], aw[k][j]);
  __syncthreads();
  for (int i = t; i < 32 * 27; i += 192) atomicAdd(dW + i, red[i]);
#endif
}

// ------------------------------------------------------------------------------------------------ head forward
// out[b,j] = sum_k relu(bn(z))[b,k] * Wp[j,k],  K = HW*C = 50 176.  Skinny GEMM (M = batch, N = 2K keypoints): split K over all waves
// of the grid, fragments straight from HBM (each operand element is used by one wave).  ONE launch: a wave issues every load of
// its K slice before the first matrix step (U steps of 32: one memory round trip; the loop with its loads under `if (b < B)` made one per
// step), the four waves of a workgroup add their tiles in LDS, and the workgroup adds its tile to the [B][Jp] accumulator in FIXED POINT
// (64-bit integer atomics, 2^-36 resolution: integer addition is associative, so the result does not depend on the order in which the
// workgroups arrive -- run-to-run identical, unlike float atomics) and takes a ticket; the workgroup that draws the last ticket reads the
// accumulator, adds the bias and writes predictions, loss scalars and d loss / d pred (park2019.py:142-156).  Round 3 ran this as two
// launches (19.6 + 8 us and a launch boundary); a last-arriver that re-reads per-workgroup partial tiles instead is bound by what one CU
// can pull (0.5 MB: 50 us, measured).  `partial` (S*B*Jp floats) holds the accumulator and, behind it, the ticket word: zero before the
// first call, restored to zero by every call.
constexpr int HEAD_U = 4;
constexpr float HEAD_FIX = 68719476736.f;   // 2^36: |partial| < 2^26 keeps the 64-bit sum of 2^16 terms exact
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const spb_head_args_t a, int kchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [C]
  float* sh = sc + a.C;
  float* tile = sh + a.C;                      // [4][64][32] per-wave accumulator tiles
  __shared__ int last_flag;
  __shared__ float lred[2][4];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  SPB_TS_DECL;
  SPB_TSR(0);
  const int KH = a.HW * a.C;
  const int wave = blockIdx.x * 4 + w;
  const int kbeg = wave * kchunk, kend = min(KH, kbeg + kchunk);
  const T* Z = reinterpret_cast<const T*>(a.Z);
  const T* Wp = reinterpret_cast<const T*>(a.Wp);
  const int NJ = a.Jp / 16;  // <= 2
  unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(a.partial);              // [B][Jp]
  unsigned* ticket = reinterpret_cast<unsigned*>(acc64 + (size_t)a.B * a.Jp);
  unsigned* poison = ticket + 1;     // count of partial sums the fixed-point accumulator cannot hold (NaN, Inf, |v| >= 2^26): the last arriver then reports NaN
  // every load of a K slice first, on clamped indices (HEAD_U steps of 32)
  uint4 wr[HEAD_U][2], zr[HEAD_U][4];
  auto fetch = [&](int bb, int k0) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < HEAD_U; ++u) {
        const int kk = k0 + u * 32 + lq * 8;
        const int kc = kk < KH ? kk : KH - 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) wr[u][j] = *reinterpret_cast<const uint4*>(Wp + (size_t)((j < NJ ? j : 0) * 16 + li) * KH + kc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = bb + i * 16 + li;
          zr[u][i] = *reinterpret_cast<const uint4*>(Z + (size_t)(b < a.B ? b : a.B - 1) * KH + kc);
        }
      }
    }
  };
  // the first slice is requested BEFORE the coefficient table: loads return in order, so the table's own loads ride the same round trip
  fetch(0, kbeg);
  bn_fwd_table<4>(a.pro, a.C, sc, sh, t, 256);
  __syncthreads();
  SPB_TSR(1);
  unsigned long long seen = 0;
  for (int bb = 0; bb < a.B; bb += 64) {
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      for (int k0 = kbeg; k0 < kend; k0 += 32 * HEAD_U) {
        if (bb != 0 || k0 != kbeg) fetch(bb, k0);
#pragma unroll
        for (int u = 0; u < HEAD_U; ++u) {
          const int kk = k0 + u * 32 + lq * 8;
          const bool kok = kk < kend;             // beyond the slice: zero weights (the activations stay finite)
          const int c0 = (kk < KH ? kk : KH - 8) % a.C;
          bf16x8_t bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = __builtin_bit_cast(bf16x8_t, (kok && j < NJ) ? wr[u][j] : make_uint4(0, 0, 0, 0));
          const float4 s0 = *reinterpret_cast<const float4*>(sc + c0), s1 = *reinterpret_cast<const float4*>(sc + c0 + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(sh + c0), h1 = *reinterpret_cast<const float4*>(sh + c0 + 4);
          const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (bb + i * 16 >= a.B) continue;     // (uniform) tile rows beyond the batch
            Raw8<T> zq; zq.u = zr[u][i];
            float v[8];
            cvt8(zq, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e] * scv[e] + shv[e], a.pro.act, a.pro.slope);
            uint4 q;
            q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, q);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = SPB_MFMA16(af, bf[j], acc[i][j]);
          }
        }
      }
    } else {
      // f32: each lane owns 4 consecutive k; MFMA step s pairs element s of every lane (same k set on both sides)
      for (int k = kbeg; k < kend; k += 16) {
        const int kk = k + lq * 4;
        const int c0 = kk % a.C;
        float bw[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < NJ) u = *reinterpret_cast<const float4*>(Wp + (size_t)(j * 16 + li) * KH + kk);
          bw[j][0] = u.x; bw[j][1] = u.y; bw[j][2] = u.z; bw[j][3] = u.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = bb + i * 16 + li;
          float av[4] = {0.f, 0.f, 0.f, 0.f};
          if (b < a.B) {
            const float4 u = *reinterpret_cast<const float4*>(Z + (size_t)b * KH + kk);
            av[0] = u.x; av[1] = u.y; av[2] = u.z; av[3] = u.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = act_fwd(av[e] * sc[c0 + e] + sh[c0 + e], a.pro.act, a.pro.slope);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[j][s], acc[i][j], 0, 0, 0);
        }
      }
    }
    SPB_TSR(2);
    // the four waves' tiles -> one partial tile of the workgroup (C layout: col = lane & 15, rows (lane >> 4) * 4 + r)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(w * 64 + i * 16 + lq * 4 + r) * 32 + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    for (int i = t; i < 64 * 32; i += 256) {
      const int b = bb + (i >> 5), col = i & 31;
      if (b < a.B && col < a.J) {
        const float v = (tile[i] + tile[2048 + i]) + (tile[4096 + i] + tile[6144 + i]);
        // RETURNING atomics, their results consumed below before the ticket is taken: the return value exists only once the addition has
        // been performed where all XCDs see it.  With fire-and-forget atomics + __threadfence() the ticket of a workgroup could overtake
        // one of its additions on the way to memory (different addresses, different channels): about once in a thousand launches the
        // last arriver read an accumulator that lacked a tile -- a wrong prediction, a loss spike of 100-500 and a wrecked update
        // (tests/test_parity_conditioned_gpu.py stopped settling).
        if (fabsf(v) < 67108864.f) seen += atomicAdd(acc64 + (size_t)b * a.Jp + col, (unsigned long long)__float2ll_rn(v * HEAD_FIX));
        else seen += atomicAdd(poison, 1u);      // a diverged network must read NaN / Inf like the reference's, not a saturated finite number
      }
    }
    __syncthreads();
  }
  // ---- ticket: the workgroup that arrives last finishes the job
  SPB_TSR(3);
  if (seen == 0x7fffffffffffff01ull) tile[0] = 1.f;   // (never true) a use of every returned value ahead of the barrier
  __threadfence();
  __syncthreads();
  SPB_TSR(4);
  if (t == 0) last_flag = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  SPB_TSR(5);
  if (!last_flag) { SPB_TS_FLUSH; return; }
  __threadfence();
  const bool bad = __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  float lx = 0.f, ly = 0.f;
  const int nout = a.B * a.J;
  for (int i0 = t; i0 < nout; i0 += 256 * 8) {
    unsigned long long fx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {          // all loads in flight (agent scope: the sums were made at the memory side by other XCDs' atomics)
      const int idx = i0 + 256 * u, ic = idx < nout ? idx : nout - 1;
      fx[u] = __hip_atomic_load(acc64 + (size_t)(ic / a.J) * a.Jp + ic % a.J, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0 + 256 * u;
      if (idx >= nout) continue;
      const int b = idx / a.J, j = idx % a.J;
      __hip_atomic_store(acc64 + (size_t)b * a.Jp + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next call
      const float s = bad ? __int_as_float(0x7fc00000) : (float)((double)(long long)fx[u] * (1.0 / (double)HEAD_FIX)) + (a.bias ? a.bias[j] : 0.f);
      a.pred[idx] = s;
      if (a.target) {
        const int nK = a.J / 2;
        const float d = s - a.target[(size_t)b * a.J + (j & 1) * nK + (j >> 1)];
        if (j & 1) ly += d * d; else lx += d * d;
        a.dout[idx] = 2.f * d / (float)a.B;
      }
    }
  }
  lx = wave_sum(lx); ly = wave_sum(ly);
  if (l == 0) { lred[0][w] = lx; lred[1][w] = ly; }
  lds_barrier();
  if (t == 0) {
    if (a.target && a.scalars) {
      const float tx = ((lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3])) / (float)a.B;
      const float ty = ((lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3])) / (float)a.B;
      a.scalars[0] = tx + ty; a.scalars[1] = tx; a.scalars[2] = ty;
    }
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(poison, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SPB_TSR(6);
  SPB_TS_FLUSH;
}

// ------------------------------------------------------------------------------------------------ head backward
// dA = dout * Wp, masked by relu'(bn(z)), plus sum(g), sum(g*xhat) for the BN below (the weight gradient is head_wgrad_kernel)
// HBK consecutive k (8-element vectors of one (position, channel) run) per workgroup.  256 (round 2; was 512): 196 workgroups
// instead of 98 on 256 CUs, 8 instead of 4 batch subsets per workgroup, and the [J][HBK] weight slab is fetched with all of a
// thread's loads in flight (the load -> LDS store loop cost one memory round trip per iteration, six of them): 32 -> ~18 us.
constexpr int HBK = 256;
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const spb_head_bwd_args_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KV = HBK / 8, NSUB = 256 / KV, WLD = 4;   // k vectors, batch subsets, weight vectors per thread (J * KV <= 1024: J <= 32)
  float* ds = reinterpret_cast<float*>(smem);     // [J][NSUB][MB] upstream gradient * gscale of the current 64 images, slot-major (below)
  float* wl = ds + (size_t)a.J * 64;              // [J][HBK] weight slab
  float* red = wl + (size_t)a.J * HBK;            // [4 waves][2][HBK]
  float* tab = red + 8 * HBK;                     // [4][HBK] scale, shift, mean, inverse std of the workgroup's entries
  static_assert(HBK == 256, "one table entry per thread");
  const int t = threadIdx.x;
  const float gs_ = a.gscale * (a.gscale_dev ? *a.gscale_dev : 1.f);     // (the float16 recipe's device-side loss scale)
  SPB_TS_DECL;
  SPB_TSR(0);
  const int KH = a.HW * a.C;
  const int kbase = blockIdx.x * HBK;
  const T* Z = reinterpret_cast<const T*>(a.Z);
  const T* Wp = reinterpret_cast<const T*>(a.Wp);
  // weight slab and this thread's z vectors: every load first (clamped indices), then the LDS stores.  The z loads used to sit in the batch
  // loop below, one memory round trip per image of the thread's subset (six at bs=48: 23.6 us for 15 MB).
  constexpr int MB = 8;                     // images per thread and pass: NSUB * MB = 64
  const int k8 = t % KV, sub = t / KV;
  const int k = kbase + k8 * 8;
  const bool kok = k < KH;
  const int kc = kok ? k : KH - 8;
  Raw8<T> wr[WLD], zq[MB];
#pragma unroll
  for (int u = 0; u < WLD; ++u) {
    const int i = t + 256 * u;
    const int ic = i < a.J * KV ? i : a.J * KV - 1;
    const int j = ic / KV, v8 = ic % KV;
    const int kk = kbase + v8 * 8;
    wr[u] = ldraw<T>(Wp + (size_t)j * KH + (kk < KH ? kk : KH - 8));
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int b = sub + i * NSUB;
    zq[i] = ldraw<T>(Z + (size_t)(b < a.B ? b : a.B - 1) * KH + kc);
  }
  // BatchNorm coefficients of this workgroup's HBK consecutive (position, channel) entries: ONE channel per thread, so the moment loads of
  // all of them are one memory round trip (eight bn_moments calls per thread, each with its uniform branches, were eight: 5 us of prologue)
  {
    const int kk = kbase + t < KH ? kbase + t : KH - 1;
    const int c = kk % a.C;
    float m, v;
    const float gm = a.pro.gamma[c], bt = a.pro.beta[c];      // requested with the sums, not after them
    bn_moments(a.pro, c, m, v);
    const float g = gm * v;
    tab[t] = g; tab[HBK + t] = bt - m * g; tab[2 * HBK + t] = m; tab[3 * HBK + t] = v;
  }
#pragma unroll
  for (int u = 0; u < WLD; ++u) {
    const int i = t + 256 * u;
    if (i < a.J * KV) {
      const int j = i / KV, v8 = i % KV;
      float v[8];
      cvt8(wr[u], v);
      const bool ok = kbase + v8 * 8 < KH;
#pragma unroll
      for (int e = 0; e < 8; ++e) wl[j * HBK + v8 * 8 + e] = ok ? v[e] : 0.f;
    }
  }
  T* G = reinterpret_cast<T*>(a.G);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // dA[b][k] = sum_j d[b][j] * w[j][k] for this thread's 8 k and its MB images (b = b0 + sub + i * NSUB): j outermost, so the two weight
  // vectors and the MB upstream gradients of a j are four 16-byte LDS reads feeding 8 * MB independent FMAs (the image-major loop read
  // three LDS values per 8 FMAs behind one another with one wave per SIMD to hide it: 12.8 of the kernel's 18 us).
  for (int b0 = 0; b0 < a.B; b0 += NSUB * MB) {
    if (b0 > 0) {          // batches beyond 64 images: the next images' z vectors, landed by the barrier below (nothing is pending inside
      __syncthreads();     // the per-image blocks then, so their stores are not waited for one by one)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int b = b0 + sub + i * NSUB;
        zq[i] = ldraw<T>(Z + (size_t)(b < a.B ? b : a.B - 1) * KH + kc);
      }
    }
    {   // upstream gradients of these 64 images, slot-major; all of a thread's loads first (J <= 32: at most 8)
      float dl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = t + 256 * u, ic = i < a.J * 64 ? i : a.J * 64 - 1;
        const int b = b0 + ((ic >> 3) & 7) + (ic & 7) * NSUB;
        dl[u] = a.dout[(b < a.B ? b : a.B - 1) * a.J + (ic >> 6)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = t + 256 * u;
        const int b = b0 + ((i >> 3) & 7) + (i & 7) * NSUB;
        if (i < a.J * 64) ds[i] = b < a.B ? dl[u] * gs_ : 0.f;
      }
    }
    __syncthreads();
    if (b0 == 0) { SPB_TSR(1); }
    if (kok) {
      float sc[8], sh[8], mu[8], is[8];
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        *reinterpret_cast<float4*>(sc + e) = *reinterpret_cast<const float4*>(tab + k8 * 8 + e);
        *reinterpret_cast<float4*>(sh + e) = *reinterpret_cast<const float4*>(tab + HBK + k8 * 8 + e);
        *reinterpret_cast<float4*>(mu + e) = *reinterpret_cast<const float4*>(tab + 2 * HBK + k8 * 8 + e);
        *reinterpret_cast<float4*>(is + e) = *reinterpret_cast<const float4*>(tab + 3 * HBK + k8 * 8 + e);
      }
      // straight-line per slot count (a `continue` per slot inside the j loop made eight scalar branches per j and an LDS wait at every one)
      auto images = [&](auto ns) {
        constexpr int NS = decltype(ns)::value;
        float da[NS][8];
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) da[i][e] = 0.f;
#pragma unroll 2
        for (int j = 0; j < a.J; ++j) {
          const float4 w0 = *reinterpret_cast<const float4*>(wl + j * HBK + k8 * 8);
          const float4 w1 = *reinterpret_cast<const float4*>(wl + j * HBK + k8 * 8 + 4);
          const float4 d0 = *reinterpret_cast<const float4*>(ds + (j * NSUB + sub) * MB);
          const float4 d1 = *reinterpret_cast<const float4*>(ds + (j * NSUB + sub) * MB + 4);
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
          for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) da[i][e] += dv[i] * wv[e];
        }
        if (b0 == 0) { SPB_TSR(4); }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int b = b0 + sub + i * NSUB;
          const bool live = b < a.B;            // per lane only when the batch is not a multiple of NSUB
          const float lv = live ? 1.f : 0.f;
          float z[8];
          cvt8(zq[i], z);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float u = z[e] * sc[e] + sh[e];
            da[i][e] = rnd<T>(da[i][e] * (lv * act_grad(u, a.pro.act, a.pro.slope)));
            s1[e] += da[i][e];
            s2[e] += da[i][e] * ((z[e] - mu[e]) * is[e]);
          }
          if (live) st8<T>(G + (size_t)b * KH + k, da[i]);
        }
      };
      switch (min(MB, (a.B - b0 + NSUB - 1) / NSUB)) {
        case 1: images(std::integral_constant<int, 1>()); break;
        case 2: images(std::integral_constant<int, 2>()); break;
        case 3: images(std::integral_constant<int, 3>()); break;
        case 4: images(std::integral_constant<int, 4>()); break;
        case 5: images(std::integral_constant<int, 5>()); break;
        case 6: images(std::integral_constant<int, 6>()); break;
        case 7: images(std::integral_constant<int, 7>()); break;
        default: images(std::integral_constant<int, 8>()); break;
      }
    }
  }
  SPB_TSR(5);
  // the two batch subsets of a wave (lanes l, l ^ 32) on the vector ALU, the four waves through LDS with plain stores: word e * KV + k8, so
  // consecutive lanes hit consecutive banks (float LDS atomics on red[k8 * 8 + e] were 16-way bank conflicts: 3.5 us)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v1 = xor32_sum(kok ? s1[e] : 0.f), v2 = xor32_sum(kok ? s2[e] : 0.f);
    if ((t & 63) < 32) {
      red[((t >> 6) * 2 + 0) * HBK + e * KV + k8] = v1;
      red[((t >> 6) * 2 + 1) * HBK + e * KV + k8] = v2;
    }
  }
  SPB_TSR(6);
  lds_barrier();      // LDS only: the stores of G above need not have landed before the sums go out
  SPB_TSR(2);
  for (int i = t; i < 2 * HBK; i += 256) {
    const int which = i / HBK, r = i % HBK;
    const int kk = kbase + (r % KV) * 8 + r / KV;
    if (kk < KH) {
      const int rep = blockIdx.x % a.oR;
      const float v = (red[which * HBK + r] + red[(2 + which) * HBK + r]) + (red[(4 + which) * HBK + r] + red[(6 + which) * HBK + r]);
      atomicAdd(a.osums + (size_t)rep * 2 * a.C + (size_t)which * a.C + (kk % a.C), v);
    }
  }
  SPB_TSR(3);
  SPB_TS_FLUSH;
}

// Weight gradient of the 7x7 head convolution: dW[j][c][hw] += sum_b dout[b][j] * relu(bn(z))[b][hw][c], dbias[j] += sum_b dout.
// One workgroup owns 8 channels and every spatial position: a thread keeps (hw, channel) pairs and all J outputs in registers,
// the 8 channels of a position are 16 contiguous bytes of z, and each dW[j][c0..c0+7][*] block is one contiguous 8*HW-float
// region written by this workgroup alone.  (The earlier mapping -- 512 consecutive k = one position, 512 channels -- scattered
// 4-byte read-modify-writes 196 bytes apart: 60 us and 2.4x the algorithmic HBM traffic.)
template <typename T>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const spb_head_bwd_args_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ds = reinterpret_cast<float*>(smem);                 // [B][32] upstream gradient * gscale, zero beyond J: the inner
                                                              // loop is then branch-free with 16-byte LDS reads (a runtime
                                                              // `j < J` test per term cost a branch + a dependent read each)
  T* zs = reinterpret_cast<T*>(smem + (size_t)a.B * 32 * 4);  // [B][HW][8] this workgroup's slab of z
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * 8, KH = a.HW * a.C, npair = a.HW * 8;
  const T* Z = reinterpret_cast<const T*>(a.Z);
  // the whole slab in one burst of independent 16-byte (bf16) loads: one memory round trip instead of one per image
  for (int i = t; i < a.B * a.HW; i += 256) {
    const int b = i / a.HW, hw = i % a.HW;
    const Raw8<T> r = ldraw<T>(Z + (size_t)b * KH + (size_t)hw * a.C + c0);
    *reinterpret_cast<Raw8<T>*>(zs + (size_t)i * 8) = r;
  }
  const float gs_ = a.gscale * (a.gscale_dev ? *a.gscale_dev : 1.f);
  for (int i = t; i < a.B * 32; i += 256) ds[i] = (i & 31) < a.J ? a.dout[(i >> 5) * a.J + (i & 31)] * gs_ : 0.f;
  const int ci = t & 7;
  float mu, is;
  const float gm = a.pro.gamma[c0 + ci], bt = a.pro.beta[c0 + ci];
  bn_moments(a.pro, c0 + ci, mu, is);
  const float sc = gm * is, sh = bt - mu * sc;
  __syncthreads();
  if (blockIdx.x == 0 && t < a.J) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += ds[b * 32 + t];
    a.dbias[t] += s;
  }
  for (int p = t; p < npair; p += 256) {          // pair = (hw = p >> 3, channel c0 + (p & 7)); p & 7 == t & 7 since 256 % 8 == 0
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (int b = 0; b < a.B; ++b) {
      const float v = act_fwd(to_f<T>(zs[(size_t)b * npair + p]) * sc + sh, a.pro.act, a.pro.slope);
      const float4* d = reinterpret_cast<const float4*>(ds + b * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 dv = d[q];
        acc[4 * q] += dv.x * v; acc[4 * q + 1] += dv.y * v; acc[4 * q + 2] += dv.z * v; acc[4 * q + 3] += dv.w * v;
      }
    }
    // accumulate into dW: all J loads first (the compiler cannot prove the J addresses distinct, and a load / add / store
    // chain per j is J serialised memory round trips: 88 of this kernel's first 115 us)
    const int hw = p >> 3;
    float* dwp = a.dW + (size_t)(c0 + ci) * a.HW + hw;
    const size_t js = (size_t)a.C * a.HW;
    float old[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) old[j] = j < a.J ? dwp[j * js] : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < a.J) dwp[j * js] = old[j] + acc[j];
  }
}

}  // namespace

int spb_stem_fwd_mfma(const float* x, const float* w, void* y, float* osums, int oR, int B, int H, int W, hipStream_t s);
int spb_stem_wgrad_mfma(const float* x, const void* G, const void* Z, const spb_bnref_t* pro, float* dW, int B, int H, int W,
                        hipStream_t s);
static int g_stem_mfma = 1;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_stem_mfma(int on) { g_stem_mfma = on; return 0; }
#endif

extern "C" int spb_stem_fwd(int dtype, const float* x, const float* w, void* y, float* osums, int oR, int B, int H,
                            int W, spb_stream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return SPB_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  long long P = (long long)B * OH * OW;
  int grid = (int)((P + 64 * 8 - 1) / (64 * 8));
  if (grid > 2048) grid = 2048;
  if (dtype == SPB_BF16 && g_stem_mfma) {   // implicit GEMM on the matrix cores (stem_mfma.hip)
    spb_stem_fwd_mfma(x, w, y, osums, oR, B, H, W, (hipStream_t)stream);
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == SPB_BF16)
    hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, (bf16_t*)y, osums, oR, B, H, W);
  else if (dtype == SPB_F32)
    hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, (float*)y, osums, oR, B, H, W);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_stem_wgrad(int dtype, const float* x, const void* G, const void* Z, const spb_bnref_t* pro,
                              float* dW, int B, int H, int W, spb_stream_t stream) {
  if (!x || !G || !Z || !pro || !dW) return SPB_E_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  long long P = (long long)B * OH * OW;
  int grid = (int)((P + 16 * 32 - 1) / (16 * 32));
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  if (dtype == SPB_BF16 && g_stem_mfma && (OW & 7) == 0) {
    spb_stem_wgrad_mfma(x, G, Z, pro, dW, B, H, W, (hipStream_t)stream);
    SPB_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == SPB_BF16)
    hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(grid), dim3(192), 0, (hipStream_t)stream, x, (const bf16_t*)G, (const bf16_t*)Z, *pro, dW, B, H, W);
  else if (dtype == SPB_F32)
    hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(grid), dim3(192), 0, (hipStream_t)stream, x, (const float*)G, (const float*)Z, *pro, dW, B, H, W);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_head_fwd(int dtype, const spb_head_args_t* a, spb_stream_t stream) {
  if (!a || !a->Z || !a->Wp || !a->partial || !a->pred) return SPB_E_ARG;
  if (a->B <= 0 || a->J <= 0 || a->J > 32 || (a->J & 1) || a->Jp < a->J || (a->Jp & 15) || a->Jp > 32) return SPB_E_SHAPE;
  if ((a->C & 7) || a->S <= 0 || (a->S & 3)) return SPB_E_SHAPE;
  if (a->target && (!a->dout || !a->scalars)) return SPB_E_ARG;
  const int KH = a->HW * a->C;
  const int kchunk = spb_ceil_div(spb_ceil_div(KH, a->S), 32) * 32;
  const size_t lds = (size_t)2 * a->C * sizeof(float) + (size_t)4 * 64 * 32 * sizeof(float);
  if (dtype == SPB_BF16)
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(a->S / 4), dim3(256), lds, (hipStream_t)stream, *a, kchunk);
  else if (dtype == SPB_F32)
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(a->S / 4), dim3(256), lds, (hipStream_t)stream, *a, kchunk);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_head_bwd(int dtype, const spb_head_bwd_args_t* a, spb_stream_t stream) {
  if (!a || !a->Z || !a->Wp || !a->dout || !a->G || !a->osums || !a->dW || !a->dbias || !a->pro.gamma) return SPB_E_ARG;
  if (a->B <= 0 || a->J <= 0 || a->J > 32 || (a->C & 7) || a->oR < 1) return SPB_E_SHAPE;
  const int KH = a->HW * a->C;
  const int gx = spb_ceil_div(KH, HBK);
  const size_t lds = ((size_t)a->J * 64 + (size_t)a->J * HBK + 12 * HBK) * sizeof(float);
  if (a->J * (HBK / 8) > 1024) return SPB_E_SHAPE;
  if (lds > 160 * 1024) return SPB_E_SHAPE;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (dtype != SPB_BF16 && dtype != SPB_F32) return SPB_E_ARG;
  hipStream_t hs = (hipStream_t)stream;
  if (a->roles != 2) {     // input gradient + BatchNorm sums
    if (dtype == SPB_BF16) hipLaunchKernelGGL(head_bwd_kernel<bf16_t>, dim3(gx, 1), dim3(256), lds, hs, *a);
    else hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(gx, 1), dim3(256), lds, hs, *a);
  }
  if (a->roles != 1) {     // weight + bias gradient: one workgroup per 8 channels, all HW positions
    const size_t l2 = (size_t)a->B * 32 * sizeof(float) + (size_t)a->B * a->HW * 8 * (dtype == SPB_BF16 ? 2 : 4);
    if (l2 > 160 * 1024) return SPB_E_SHAPE;
    static bool wg_attr = false;
    if (!wg_attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&head_wgrad_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&head_wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      wg_attr = true;
    }
    if (dtype == SPB_BF16) hipLaunchKernelGGL(head_wgrad_kernel<bf16_t>, dim3(a->C / 8), dim3(256), l2, hs, *a);
    else hipLaunchKernelGGL(head_wgrad_kernel<float>, dim3(a->C / 8), dim3(256), l2, hs, *a);
  }
  SPB_CHECK_LAUNCH();
  return 0;
}
