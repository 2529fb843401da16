// Style decoder (Ghiasi), REFERENCE-PRECISION mode: float32 storage and arithmetic.
// The reference runs `styleAugmentor(images)` outside autocast, i.e. in fp32 (trainer.py:68-69, ghiasi.py:106-136); the fast path of this
// build is bf16 on the matrix cores (csrc/ghiasi.hip, ghiasi_wide.hip; mean |d| 3e-3 on a [0,1] image).  Ghiasi(precision="fp32") runs the
// same layer sequence through the three kernels below instead: direct convolution on the vector units, every tensor float32, the same
// "raw conv output + per-(image, channel) sums" representation of instance-normalised tensors.  It exists for parity (tests hold it to the
// float32 CPU restatement of ghiasi.py at 1e-4), not for speed: ~40 ms per 48 images against 1.3 ms.
// C-ABI: spb_gconv(SPB_F32, ...) (same argument struct: X / Y float32 NHWC, W float32 [Cout][KH*KH][Cin], `coef` table only),
// spb_in_apply_f32, spb_final_sigmoid_f32 (include/spb_hip.h).
#include "common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int u, int n) { return u < 0 ? -u : (u >= n ? 2 * (n - 1) - u : u); }

// thread: one output pixel x 8 consecutive output channels; lanes of a wave are 64 consecutive pixels of one (image, channel group), so the
// weight reads are wave-uniform broadcasts and the statistics reduce over the wave before they reach memory (Hout * Wout % 64 == 0)
__global__ __launch_bounds__(256) void gconv_f32_kernel(const spb_gconv_args_t g) {
  const int K = g.KH, pad = K / 2, up = g.upsample, st = g.stride;
  const int Hup = g.Hin * up, Wup = g.Win * up;
  const int Hout = Hup / st, Wout = Wup / st;
  const int ncg = (g.Cout + 7) / 8;
  const long long hw = (long long)Hout * Wout;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)g.B * ncg * hw;
  const long long ic = idx < total ? idx : total - 1;
  const int p = (int)(ic % hw), cg = (int)((ic / hw) % ncg), b = (int)(ic / (hw * ncg));
  const int oy = p / Wout, ox = p % Wout, co0 = cg * 8;
  const float* X = reinterpret_cast<const float*>(g.X) + (size_t)b * g.Hin * g.Win * g.Cin;
  const float* W = reinterpret_cast<const float*>(g.W);
  const float* cf = g.coef ? g.coef + (size_t)b * g.Cin * 2 : nullptr;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < K; ++ky) {
    const int iy = reflect_idx(oy * st + ky - pad, Hup) / up;
    for (int kx = 0; kx < K; ++kx) {
      const int ix = reflect_idx(ox * st + kx - pad, Wup) / up;
      const float* xp = X + ((size_t)iy * g.Win + ix) * g.Cin;
      const float* wp = W + (size_t)(ky * K + kx) * g.Cin;
      for (int ci = 0; ci < g.Cin; ++ci) {
        float xv = xp[ci];
        if (cf) { xv = xv * cf[2 * ci] + cf[2 * ci + 1]; xv = g.relu ? fmaxf(xv, 0.f) : xv; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int co = co0 + j < g.Cout ? co0 + j : g.Cout - 1;
          acc[j] = fmaf(xv, wp[(size_t)co * K * K * g.Cin + ci], acc[j]);
        }
      }
    }
  }
  float* Y = reinterpret_cast<float*>(g.Y);
  const bool live = idx < total;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool cok = co0 + j < g.Cout;
    const float v = acc[j] + ((g.bias && cok) ? g.bias[co0 + j] : 0.f);
    if (live && cok) Y[((size_t)b * hw + p) * g.ldc + co0 + j] = v;
    if (g.stats) {   // (uniform) the wave's 64 pixels belong to one image and channel group
      const float m = (live && cok) ? v : 0.f;
      const float s1 = wave_sum(m), s2 = wave_sum(m * m);
      if ((threadIdx.x & 63) == 0 && cok) {
        atomicAdd(g.stats + ((size_t)b * g.Cout + co0 + j) * 2, s1);
        atomicAdd(g.stats + ((size_t)b * g.Cout + co0 + j) * 2 + 1, s2);
      }
    }
  }
}

// Y = [res +] act(X * scale + shift), NHWC float32
__global__ void in_apply_f32_kernel(const float* X, const float* coef, const float* res, float* Y, long long hw, int C, int relu, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long long b = i / ((long long)C * hw);
  const float* cf = coef + ((size_t)b * C + c) * 2;
  float u = X[i] * cf[0] + cf[1];
  u = relu ? fmaxf(u, 0.f) : u;
  Y[i] = res ? u + res[i] : u;
}

// out (NCHW, 3 channels) = sigmoid(Z * scale + shift), Z NHWC float32 with channel stride ldc
__global__ void final_sigmoid_f32_kernel(const float* Z, const float* coef, float* out, int B, long long hw, int ldc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * hw) return;
  const long long b = i / hw, p = i % hw;
  for (int c = 0; c < 3; ++c) {
    const float u = Z[i * ldc + c] * coef[((size_t)b * 3 + c) * 2] + coef[((size_t)b * 3 + c) * 2 + 1];
    out[((size_t)b * 3 + c) * hw + p] = 1.f / (1.f + expf(-u));
  }
}

}  // namespace

int spb_gconv_f32(const spb_gconv_args_t* a, hipStream_t stream) {
  if (a->in_stats != nullptr) return SPB_E_UNSUPPORTED;       // the f32 mode takes the coefficient table (spb_in_coef)
  const int Hout = a->Hin * a->upsample / a->stride, Wout = a->Win * a->upsample / a->stride;
  if (((long long)Hout * Wout) & 63) return SPB_E_SHAPE;      // a wave = 64 pixels of one image and channel group
  const long long total = (long long)a->B * ((a->Cout + 7) / 8) * Hout * Wout;
  hipLaunchKernelGGL(gconv_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, *a);
  return 0;
}

extern "C" int spb_in_apply_f32(const float* X, const float* coef, const float* res, float* Y, int B, long long hw, int C, int relu,
                                spb_stream_t stream) {
  if (!X || !coef || !Y || B <= 0 || hw <= 0 || C <= 0) return SPB_E_ARG;
  const long long n = (long long)B * hw * C;
  hipLaunchKernelGGL(in_apply_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, coef, res, Y, hw, C, relu, n);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_final_sigmoid_f32(const float* Z, const float* coef, float* out, int B, long long hw, int ldc, spb_stream_t stream) {
  if (!Z || !coef || !out || B <= 0 || hw <= 0 || ldc < 3) return SPB_E_ARG;
  const long long n = (long long)B * hw;
  hipLaunchKernelGGL(final_sigmoid_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Z, coef, out, B, hw, ldc);
  SPB_CHECK_LAUNCH();
  return 0;
}
