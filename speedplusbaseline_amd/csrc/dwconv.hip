// Depthwise 3x3 convolution (pad 1, stride 1|2), NHWC, forward / input-gradient / weight-gradient.
// Replaces nn.Conv2d(C,C,3,groups=C) + BN + ReLU(6) at reference park2019.py:47-49 and in the torchvision
// MobileNetV2 inverted-residual blocks (park2019.py:107-108).
//
// 9 MACs per element: no matrix-core work here, the kernels are HBM/L2 streaming.  A thread owns 8 consecutive
// channels (one 16-byte vector) and walks output pixels; a workgroup owns a 64-channel slab so the per-channel BN
// sums / weight gradients reduce inside LDS and cost one global atomic per channel per workgroup.  The previous
// layer's BN+activation (forward) or BN-backward (gradients) is applied on the fly to every loaded vector, so the
// normalised tensors never exist in HBM.
#include "common.h"

namespace {

struct DwGeom {
  int cgl_n, cgl_shift, npl;  // cg lanes per workgroup (4 or 8) and pixel lanes
};

__device__ __forceinline__ void dw_thread(const spb_dw_args_t& a, int& cg, int& pl, int& npl, bool& valid) {
  const int CG = a.C >> 3;
  const int cgl_n = CG < 8 ? 4 : 8;
  const int t = threadIdx.x;
  const int cgl = t % cgl_n;
  pl = t / cgl_n;
  npl = 256 / cgl_n;
  cg = blockIdx.y * cgl_n + cgl;
  valid = cg < CG;
}

// reduce per-thread channel sums over the workgroup and push them to the global accumulator
__device__ __forceinline__ void dw_push_sums(float* red /*[2][64]*/, const float s1[8], const float s2[8], int cgl_n,
                                             int C, float* osums, int oR, bool valid) {
  const int t = threadIdx.x;
  if (t < 128) red[t] = 0.f;
  __syncthreads();
  const int cgl = t % cgl_n;
  if (valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&red[cgl * 8 + j], s1[j]);
      atomicAdd(&red[64 + cgl * 8 + j], s2[j]);
    }
  }
  __syncthreads();
  if (t < 128) {
    const int which = t >> 6, cl = t & 63;
    const int c = blockIdx.y * cgl_n * 8 + cl;
    if (cl < cgl_n * 8 && c < C) {
      const int rep = (blockIdx.x + blockIdx.y) % oR;
      atomicAdd(osums + (size_t)rep * 2 * C + (size_t)which * C + c, red[which * 64 + cl]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const spb_dw_args_t a) {
  __shared__ float red[128];
  int cg, pl, npl; bool valid;
  dw_thread(a, cg, pl, npl, valid);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const long long P = (long long)a.B * OH * OW;
  const int c0 = cg * 8;
  const T* X = reinterpret_cast<const T*>(a.X);
  T* Y = reinterpret_cast<T*>(a.Y);

  float wt[9][8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f; }
  if (valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bn_fwd_coef(a.pro, c0 + j, sc[j], sh[j]);
#pragma unroll
      for (int k = 0; k < 9; ++k) wt[k][j] = a.Wd[(size_t)(c0 + j) * 9 + k];
    }
    for (long long p = (long long)blockIdx.x * npl + pl; p < P; p += (long long)gridDim.x * npl) {
      const int ow = (int)(p % OW);
      const int oh = (int)((p / OW) % OH);
      const int b = (int)(p / ((long long)OW * OH));
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ih = oh * st - 1 + ky;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iw = ow * st - 1 + kx;
          if (iw < 0 || iw >= W) continue;
          float x[8];
          ld8<T>(X + ((size_t)(b * H + ih) * W + iw) * C + c0, x);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[j] += act_fwd(x[j] * sc[j] + sh[j], a.pro.act, a.pro.slope) * wt[ky * 3 + kx][j];
        }
      }
      rnd8<T>(acc);
      st8<T>(Y + (size_t)p * C + c0, acc);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
    }
  }
  if (a.epi_mode == 1) dw_push_sums(red, s1, s2, (C >> 3) < 8 ? 4 : 8, C, a.osums, a.oR, valid);
}

// dA[b,ih,iw,c] = sum_{ky,kx} dz[b,oh,ow,c] * w[c,ky,kx]  with  oh*stride - 1 + ky = ih
template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const spb_dw_args_t a) {
  __shared__ float red[128];
  int cg, pl, npl; bool valid;
  dw_thread(a, cg, pl, npl, valid);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const long long P = (long long)a.B * H * W;
  const int c0 = cg * 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Rg = reinterpret_cast<const T*>(a.res);
  const T* Zo = reinterpret_cast<const T*>(a.Zout);
  T* Y = reinterpret_cast<T*>(a.Y);

  float wt[9][8], p0[8], p1[8], p2[8], s1[8], s2[8];
  float e_sc[8], e_sh[8], e_mu[8], e_is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bn_bwd_coef(a.pro, c0 + j, p0[j], p1[j], p2[j]);
#pragma unroll
      for (int k = 0; k < 9; ++k) wt[k][j] = a.Wd[(size_t)(c0 + j) * 9 + k];
      e_sc[j] = 1.f; e_sh[j] = 0.f; e_mu[j] = 0.f; e_is[j] = 0.f;
      if (a.epi_mode == 2 && a.epi.gamma != nullptr) {
        bn_moments(a.epi, c0 + j, e_mu[j], e_is[j]);
        e_sc[j] = a.epi.gamma[c0 + j] * e_is[j];
        e_sh[j] = a.epi.beta[c0 + j] - e_mu[j] * e_sc[j];
      }
    }
    for (long long p = (long long)blockIdx.x * npl + pl; p < P; p += (long long)gridDim.x * npl) {
      const int iw = (int)(p % W);
      const int ih = (int)((p / W) % H);
      const int b = (int)(p / ((long long)W * H));
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ty = ih + 1 - ky;
        if (ty < 0 || (st == 2 && (ty & 1))) continue;
        const int oh = ty / st;
        if (oh >= OH) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tx = iw + 1 - kx;
          if (tx < 0 || (st == 2 && (tx & 1))) continue;
          const int ow = tx / st;
          if (ow >= OW) continue;
          const size_t o = ((size_t)(b * OH + oh) * OW + ow) * C + c0;
          float g[8], z[8];
          ld8<T>(G + o, g);
          if (Z) ld8<T>(Z + o, z);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = 0.f;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (g[j] * p0[j] + z[j] * p1[j] + p2[j]) * wt[ky * 3 + kx][j];
        }
      }
      const size_t o = (size_t)p * C + c0;
      if (a.epi_mode == 2) {
        float z[8];
        ld8<T>(Zo + o, z);
        if (Rg) {
          float rr[8];
          ld8<T>(Rg + o, rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += rr[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float u = z[j] * e_sc[j] + e_sh[j];
          acc[j] = rnd<T>(acc[j] * act_grad(u, a.epi.act, a.epi.slope));
          s1[j] += acc[j];
          s2[j] += acc[j] * ((z[j] - e_mu[j]) * e_is[j]);
        }
      }
      st8<T>(Y + o, acc);
    }
  }
  if (a.epi_mode == 2) dw_push_sums(red, s1, s2, (C >> 3) < 8 ? 4 : 8, C, a.osums, a.oR, valid);
}

// dW[c,ky,kx] += sum_{b,oh,ow} dz[b,oh,ow,c] * act(bn(x))[b, oh*s-1+ky, ow*s-1+kx, c]
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const spb_dw_args_t a) {
  __shared__ float red[64 * 9];
  int cg, pl, npl; bool valid;
  dw_thread(a, cg, pl, npl, valid);
  const int C = a.C, H = a.H, W = a.W, st = a.stride;
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  const long long P = (long long)a.B * OH * OW;
  const int c0 = cg * 8;
  const T* G = reinterpret_cast<const T*>(a.X);
  const T* Z = reinterpret_cast<const T*>(a.X2);
  const T* Xin = reinterpret_cast<const T*>(a.Xin);
  const int t = threadIdx.x;
  const int cgl_n = (C >> 3) < 8 ? 4 : 8;

  for (int i = t; i < 64 * 9; i += 256) red[i] = 0.f;
  __syncthreads();
  if (valid) {
    float p0[8], p1[8], p2[8], sc[8], sh[8], aw[9][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bn_bwd_coef(a.pro, c0 + j, p0[j], p1[j], p2[j]);
      bn_fwd_coef(a.pro_in, c0 + j, sc[j], sh[j]);
#pragma unroll
      for (int k = 0; k < 9; ++k) aw[k][j] = 0.f;
    }
    for (long long p = (long long)blockIdx.x * npl + pl; p < P; p += (long long)gridDim.x * npl) {
      const int ow = (int)(p % OW);
      const int oh = (int)((p / OW) % OH);
      const int b = (int)(p / ((long long)OW * OH));
      float g[8], z[8], dz[8];
      ld8<T>(G + (size_t)p * C + c0, g);
      if (Z) ld8<T>(Z + (size_t)p * C + c0, z);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) dz[j] = g[j] * p0[j] + z[j] * p1[j] + p2[j];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ih = oh * st - 1 + ky;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iw = ow * st - 1 + kx;
          if (iw < 0 || iw >= W) continue;
          float x[8];
          ld8<T>(Xin + ((size_t)(b * H + ih) * W + iw) * C + c0, x);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            aw[ky * 3 + kx][j] += dz[j] * act_fwd(x[j] * sc[j] + sh[j], a.pro_in.act, a.pro_in.slope);
        }
      }
    }
    const int cgl = t % cgl_n;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&red[(cgl * 8 + j) * 9 + k], aw[k][j]);
  }
  __syncthreads();
  for (int i = t; i < cgl_n * 8 * 9; i += 256) {
    const int c = blockIdx.y * cgl_n * 8 + i / 9;
    if (c < C) atomicAdd(a.dW + (size_t)c * 9 + (i % 9), red[i]);
  }
}

dim3 dw_grid(const spb_dw_args_t& a, long long P) {
  const int CG = a.C >> 3;
  const int cgl_n = CG < 8 ? 4 : 8;
  const int npl = 256 / cgl_n;
  const int gy = (CG + cgl_n - 1) / cgl_n;
  long long gx = (P + (long long)npl * 8 - 1) / ((long long)npl * 8);
  const long long cap = 4096 / gy > 1 ? 4096 / gy : 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

int dw_check(const spb_dw_args_t* a) {
  if (!a || !a->X || !a->Wd) return SPB_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 7)) return SPB_E_SHAPE;
  if (a->stride != 1 && a->stride != 2) return SPB_E_SHAPE;
  return 0;
}

}  // namespace

extern "C" int spb_dwconv_fwd(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y || (a->epi_mode == 1 && (!a->osums || a->oR < 1))) return SPB_E_ARG;
  const int OH = (a->H - 1) / a->stride + 1, OW = (a->W - 1) / a->stride + 1;
  const dim3 grid = dw_grid(*a, (long long)a->B * OH * OW);
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dwconv_dgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->Y) return SPB_E_ARG;
  if (a->epi_mode == 2 && (!a->osums || a->oR < 1 || !a->Zout)) return SPB_E_ARG;
  const dim3 grid = dw_grid(*a, (long long)a->B * a->H * a->W);
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_dgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_dwconv_wgrad(int dtype, const spb_dw_args_t* a, spb_stream_t stream) {
  int e = dw_check(a);
  if (e) return e;
  if (!a->dW || !a->Xin) return SPB_E_ARG;
  const int OH = (a->H - 1) / a->stride + 1, OW = (a->W - 1) / a->stride + 1;
  dim3 grid = dw_grid(*a, (long long)a->B * OH * OW);
  if (dtype == SPB_BF16) hipLaunchKernelGGL(dw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(dw_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}
