// The two dense convolutions of KRN that are not 1x1/depthwise:
//   stem  Conv2d(3,32,3,stride 2,pad 1,bias=False) on the NCHW f32 image  (torchvision mobilenet_v2.features[0],
//         reached through reference park2019.py:107-108)
//   head  Conv2d(1024,2K,kernel 7) on the 7x7 map, i.e. one fully-connected layer over (h,w,c), followed by the
//         (x,y) de-interleave and the summed per-keypoint MSE (park2019.py:121,139-162).
#include "common.h"

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
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&red[cg * 8 + j], s1[j]);
      atomicAdd(&red[32 + cg * 8 + j], s2[j]);
    }
    __syncthreads();
    if (t < 64) atomicAdd(osums + (size_t)(blockIdx.x % oR) * 64 + t, red[t]);
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
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&red[(cg * 8 + j) * 27 + ci * 9 + k], aw[k][j]);
  __syncthreads();
  for (int i = t; i < 32 * 27; i += 192) atomicAdd(dW + i, red[i]);
}

// ------------------------------------------------------------------------------------------------ head forward
// out[b,j] = sum_k relu(bn(z))[b,k] * Wp[j,k],  K = HW*C = 50 176.  Skinny GEMM (M = batch, N = 2K keypoints):
// split K over all waves of the grid, fragments straight from HBM (each operand element is used by one wave),
// partial [wave][B][Jp] tiles reduced by the loss kernel.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const spb_head_args_t a, int kchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [C]
  float* sh = sc + a.C;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  bn_fwd_table<4>(a.pro, a.C, sc, sh, t, 256);
  // the loss scalars head_reduce_kernel accumulates into: zeroed here (that kernel starts after this one has finished) instead of by a
  // memset launch between the two (7 us on the launch stream for 12 bytes)
  if (blockIdx.x == 0 && t < 3 && a.target && a.scalars) a.scalars[t] = 0.f;
  __syncthreads();
  const int KH = a.HW * a.C;
  const int wave = blockIdx.x * 4 + w;
  const int kbeg = wave * kchunk, kend = min(KH, kbeg + kchunk);
  const T* Z = reinterpret_cast<const T*>(a.Z);
  const T* Wp = reinterpret_cast<const T*>(a.Wp);
  const int NJ = a.Jp / 16;  // <= 2
  for (int bb = 0; bb < a.B; bb += 64) {
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      for (int k = kbeg; k < kend; k += 32) {
        const int kk = k + lq * 8;
        const int c0 = kk % a.C;
        bf16x8_t bf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint4 u = make_uint4(0, 0, 0, 0);
          if (j < NJ) u = *reinterpret_cast<const uint4*>(Wp + (size_t)(j * 16 + li) * KH + kk);
          bf[j] = __builtin_bit_cast(bf16x8_t, u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = bb + i * 16 + li;
          float v[8];
          if (b < a.B) {
            ld8<T>(Z + (size_t)b * KH + kk, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e] * sc[c0 + e] + sh[c0 + e], a.pro.act, a.pro.slope);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          }
          uint4 u;
          u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
          u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
          const bf16x8_t af = __builtin_bit_cast(bf16x8_t, u);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = SPB_MFMA16(af, bf[j], acc[i][j]);
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = bb + i * 16 + lq * 4 + r, col = j * 16 + li;
          if (b < a.B && col < a.Jp) a.partial[((size_t)wave * a.B + b) * a.Jp + col] = acc[i][j][r];
        }
  }
}

// pred = sum of split-K partials + bias; loss = sum_k mean_b (x-tx)^2 + (y-ty)^2 (park2019.py:142-156).
// 64 outputs per workgroup, 4 lane groups each summing a quarter of the partials with independent (pipelined) loads.
__global__ __launch_bounds__(256) void head_reduce_kernel(const spb_head_args_t a) {
  __shared__ float red[4][64];
  __shared__ float rl[2][64];
  const int t = threadIdx.x, i = t & 63, q = t >> 6;
  const int idx = blockIdx.x * 64 + i;
  const bool ok = idx < a.B * a.J;
  const int b = ok ? idx / a.J : 0, j = ok ? idx % a.J : 0;
  const int per = a.S / 4;
  const size_t stride = (size_t)a.B * a.Jp;
  const float* p = a.partial + ((size_t)q * per * a.B + b) * a.Jp + j;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = 0;
  for (; w + 4 <= per; w += 4) {
    s0 += p[(size_t)w * stride]; s1 += p[(size_t)(w + 1) * stride];
    s2 += p[(size_t)(w + 2) * stride]; s3 += p[(size_t)(w + 3) * stride];
  }
  for (; w < per; ++w) s0 += p[(size_t)w * stride];
  red[q][i] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0) {
    float lx = 0.f, ly = 0.f;
    if (ok) {
      const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i] + (a.bias ? a.bias[j] : 0.f);
      a.pred[idx] = s;
      if (a.target) {
        const int nK = a.J / 2;
        const float d = s - a.target[(size_t)b * a.J + (j & 1) * nK + (j >> 1)];
        if (j & 1) ly = d * d; else lx = d * d;
        a.dout[idx] = 2.f * d / (float)a.B;
      }
    }
    lx = wave_sum(lx); ly = wave_sum(ly);
    if (i == 0 && a.target && a.scalars) {
      lx /= (float)a.B; ly /= (float)a.B;
      atomicAdd(a.scalars + 0, lx + ly); atomicAdd(a.scalars + 1, lx); atomicAdd(a.scalars + 2, ly);
    }
  }
  (void)rl;
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
  float* ds = reinterpret_cast<float*>(smem);     // [B][J] upstream gradient * gscale
  float* wl = ds + ((a.B * a.J + 3) & ~3);        // [J][HBK] weight slab
  float* red = wl + (size_t)a.J * HBK;            // [2][HBK]
  const int t = threadIdx.x;
  const int KH = a.HW * a.C;
  const int kbase = blockIdx.x * HBK;
  const T* Z = reinterpret_cast<const T*>(a.Z);
  const T* Wp = reinterpret_cast<const T*>(a.Wp);
  // weight slab: every load of this thread first (clamped indices), then the LDS stores
  Raw8<T> wr[WLD];
#pragma unroll
  for (int u = 0; u < WLD; ++u) {
    const int i = t + 256 * u;
    const int ic = i < a.J * KV ? i : a.J * KV - 1;
    const int j = ic / KV, v8 = ic % KV;
    const int kk = kbase + v8 * 8;
    wr[u] = ldraw<T>(Wp + (size_t)j * KH + (kk < KH ? kk : KH - 8));
  }
  for (int i = t; i < a.B * a.J; i += 256) ds[i] = a.dout[i] * a.gscale;
  const int k8 = t % KV, sub = t / KV;
  const int k = kbase + k8 * 8;
  const bool kok = k < KH;
  const int c0 = (kok ? k : 0) % a.C;
  float sc[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bn_moments(a.pro, c0 + j, mu[j], is[j]);
    sc[j] = a.pro.gamma[c0 + j] * is[j];
    sh[j] = a.pro.beta[c0 + j] - mu[j] * sc[j];
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
  for (int i = t; i < 2 * HBK; i += 256) red[i] = 0.f;
  __syncthreads();
  T* G = reinterpret_cast<T*>(a.G);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (kok) {
    for (int b = sub; b < a.B; b += NSUB) {
      float da[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) da[e] = 0.f;
      for (int j = 0; j < a.J; ++j) {
        const float d = ds[b * a.J + j];
        const float4 w0 = *reinterpret_cast<const float4*>(wl + j * HBK + k8 * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + j * HBK + k8 * 8 + 4);
        da[0] += d * w0.x; da[1] += d * w0.y; da[2] += d * w0.z; da[3] += d * w0.w;
        da[4] += d * w1.x; da[5] += d * w1.y; da[6] += d * w1.z; da[7] += d * w1.w;
      }
      float z[8];
      ld8<T>(Z + (size_t)b * KH + k, z);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float u = z[e] * sc[e] + sh[e];
        da[e] = rnd<T>(da[e] * act_grad(u, a.pro.act, a.pro.slope));
        s1[e] += da[e];
        s2[e] += da[e] * ((z[e] - mu[e]) * is[e]);
      }
      st8<T>(G + (size_t)b * KH + k, da);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      atomicAdd(&red[k8 * 8 + e], s1[e]);
      atomicAdd(&red[HBK + k8 * 8 + e], s2[e]);
    }
  }
  __syncthreads();
  for (int i = t; i < 2 * HBK; i += 256) {
    const int which = i / HBK, kk = kbase + (i % HBK);
    if (kk < KH) {
      const int rep = blockIdx.x % a.oR;
      atomicAdd(a.osums + (size_t)rep * 2 * a.C + (size_t)which * a.C + (kk % a.C), red[i]);
    }
  }
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
  for (int i = t; i < a.B * 32; i += 256) ds[i] = (i & 31) < a.J ? a.dout[(i >> 5) * a.J + (i & 31)] * a.gscale : 0.f;
  const int ci = t & 7;
  float mu, is;
  bn_moments(a.pro, c0 + ci, mu, is);
  const float sc = a.pro.gamma[c0 + ci] * is, sh = a.pro.beta[c0 + ci] - mu * sc;
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
extern "C" int spb_debug_set_stem_mfma(int on) { g_stem_mfma = on; return 0; }

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
  const size_t lds = (size_t)2 * a->C * sizeof(float);
  if (dtype == SPB_BF16)
    hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(a->S / 4), dim3(256), lds, (hipStream_t)stream, *a, kchunk);
  else if (dtype == SPB_F32)
    hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(a->S / 4), dim3(256), lds, (hipStream_t)stream, *a, kchunk);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  hipLaunchKernelGGL(head_reduce_kernel, dim3(spb_ceil_div(a->B * a->J, 64)), dim3(256), 0, (hipStream_t)stream, *a);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_head_bwd(int dtype, const spb_head_bwd_args_t* a, spb_stream_t stream) {
  if (!a || !a->Z || !a->Wp || !a->dout || !a->G || !a->osums || !a->dW || !a->dbias || !a->pro.gamma) return SPB_E_ARG;
  if (a->B <= 0 || a->J <= 0 || a->J > 32 || (a->C & 7) || a->oR < 1) return SPB_E_SHAPE;
  const int KH = a->HW * a->C;
  const int gx = spb_ceil_div(KH, HBK);
  const size_t lds = ((size_t)((a->B * a->J + 3) & ~3) + (size_t)a->J * HBK + 2 * HBK) * sizeof(float);
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
