// GPU input pipeline for the KRN / SPN datasets (reference src/datasets/transforms.py:38-244, called per sample from
// src/datasets/Park2019KRNDataset.py:81-109): resize of the cropped region of interest to the network input, ToTensor, and the
// four augmentations, for a whole batch at once.  SURVEY.md section 8(f) rank 1: at >10^4 images/s per GPU the reference's
// per-sample PIL / torchvision pipeline in DataLoader workers is the wall.
//
// What runs where: the host decodes the frame and CROPS it (a slice); the crops of a batch arrive as one packed uint8 buffer.
//   pp_coeffs   per image and axis, the separable filter of PIL's Image.resize(BILINEAR) -- torchvision 0.9's
//               resized_crop on a PIL image is crop + resize (transforms.py:158,183) -- exactly as Pillow's Resample.c
//               computes it: support scaled by the shrink factor (antialiasing), double-precision triangle weights normalised
//               to 1, then 22-bit fixed point (round half away from zero).  No FMA contraction in here.
//   pp_hpass    horizontal pass over every source row into a uint8 intermediate (32-bit accumulate, +2^21, >>22, clip8)
//   pp_finish   vertical pass, ToTensor (v / 255, f32), quarter-turn rotation and flip as index permutations, brightness /
//               contrast clamp(a*x + b, 0, 1), Gaussian noise clamp(x + n*std, 0, 1); writes NCHW float32.
// Bit-exact against Pillow + the reference's tensor arithmetic given the same random draws (tests/test_preproc_gpu.py).
#include "common.h"

namespace {

constexpr int KMAX = 24;           // filter taps per output pixel: ceil(shrink factor) * 2 + 1 <= 24  (shrink <= 11.5x)
constexpr int PRECISION_BITS = 22; // 32 - 8 - 2, Pillow Resample.c

struct PPImage {                   // one row of the per-image table (int32 x 8), see spb_preproc_args_t
  int off_lo, off_hi, h, w, rot, flip, flags, pad;
};

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// grid (B, 2): axis 0 = horizontal (in size w), axis 1 = vertical (in size h); thread xx < S
__global__ void pp_coeffs_kernel(const int* __restrict__ tab, int* __restrict__ bounds, int* __restrict__ kk, int S) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, axis = blockIdx.y;
  const int in_size = axis == 0 ? tab[b * 8 + 3] : tab[b * 8 + 2];
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
    const double in0 = 0.0, in1 = (double)in_size;
    const double scale = (in1 - in0) / (double)S;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;          // bilinear: support 1
    const double center = in0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KMAX) xmax = KMAX;                        // host rejects shapes that would need more taps
    double w[KMAX];
    double ww = 0.0;
    for (int x = 0; x < KMAX; ++x) {
      double v = 0.0;
      if (x < xmax) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        v = a < 1.0 ? 1.0 - a : 0.0;
      }
      w[x] = v;
      ww += v;
    }
    int* ko = kk + ((size_t)(b * 2 + axis) * S + xx) * KMAX;
    for (int x = 0; x < KMAX; ++x) {
      double v = w[x];
      if (x < xmax && ww != 0.0) v /= ww;
      const double s = v * (double)(1 << PRECISION_BITS);
      ko[x] = v < 0 ? (int)(-0.5 + s) : (int)(0.5 + s);
    }
    bounds[((size_t)(b * 2 + axis) * S + xx) * 2] = xmin;
    bounds[((size_t)(b * 2 + axis) * S + xx) * 2 + 1] = xmax;
  }
}

// grid (ceil(maxH * S / 256), B): thread = (row y, output column xx); all C channels of the pixel
template <int C>
__global__ __launch_bounds__(256) void pp_hpass_kernel(const unsigned char* __restrict__ src, const int* __restrict__ tab,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       unsigned char* __restrict__ tmp, int S, int maxH) {
  const int b = blockIdx.y;
  const int h = tab[b * 8 + 2], w = tab[b * 8 + 3];
  const long long off = ((long long)tab[b * 8 + 1] << 32) | (unsigned)tab[b * 8];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int y = i / S, xx = i % S;
  if (y >= h) return;
  const int* bd = bounds + ((size_t)(b * 2) * S + xx) * 2;
  const int xmin = bd[0], n = bd[1];
  const int* k = kk + ((size_t)(b * 2) * S + xx) * KMAX;
  const unsigned char* row = src + off + ((size_t)y * w + xmin) * C;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)row[x * C + c] * kv;
  }
  unsigned char* o = tmp + (((size_t)b * maxH + y) * S + xx) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = clip8(acc[c]);
}

// grid (ceil(S*S/256), B): thread = output pixel (yo, xo) of the AUGMENTED image; pulls its source pixel (y0, x0)
template <int C>
__global__ __launch_bounds__(256) void pp_finish_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ tab,
                                                        const float* __restrict__ ftab, const int* __restrict__ bounds,
                                                        const int* __restrict__ kk, const float* __restrict__ noise,
                                                        float* __restrict__ out, int S, int maxH, float noise_std) {
#pragma clang fp contract(off)   // a*x + b and x + n*std are two rounded float32 operations each in the reference (no FMA)
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * S) return;
  const int yo = i / S, xo = i % S;
  const int rot = tab[b * 8 + 4], flip = tab[b * 8 + 5], flags = tab[b * 8 + 6];
  // undo Flip (transforms.py:56-68), then undo Rotate (transforms.py:38-54: T.rotate is counter-clockwise, k quarter turns)
  int y1 = yo, x1 = xo;
  if (flip == 1) x1 = S - 1 - xo;            // horizontal
  else if (flip == 2) y1 = S - 1 - yo;       // vertical
  int y0 = y1, x0 = x1;
  if (rot == 1) { y0 = x1; x0 = S - 1 - y1; }             // new[i][j] = old[j][S-1-i]
  else if (rot == 2) { y0 = S - 1 - y1; x0 = S - 1 - x1; }
  else if (rot == 3) { y0 = S - 1 - x1; x0 = y1; }
  const int* bd = bounds + ((size_t)(b * 2 + 1) * S + y0) * 2;
  const int ymin = bd[0], n = bd[1];
  const int* k = kk + ((size_t)(b * 2 + 1) * S + y0) * KMAX;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
  const unsigned char* col = tmp + (((size_t)b * maxH + ymin) * S + x0) * C;
  for (int y = 0; y < n; ++y) {
    const int kv = k[y];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)col[(size_t)y * S * C + c] * kv;
  }
  const float a = ftab[b * 2], bb = ftab[b * 2 + 1];
#pragma unroll
  for (int c3 = 0; c3 < 3; ++c3) {
    const int c = C == 1 ? 0 : c3;                        // a grey frame is 'RGB' with three equal bands (convert('RGB'))
    float v = (float)clip8(acc[c]) / 255.0f;              // ToTensor (transforms.py:192-196)
    if (flags & 1) { const float m = a * v; v = fminf(fmaxf(m + bb, 0.f), 1.f); }           // BrightnessContrast, :70-92
    if (flags & 2) {                                                                         // GaussianNoise, :94-105
      const float nz = noise[(((size_t)b * 3 + c3) * S + yo) * S + xo] * noise_std;
      v = fminf(fmaxf(v + nz, 0.f), 1.f);
    }
    out[(((size_t)b * 3 + c3) * S + yo) * S + xo] = v;
  }
}

}  // namespace

extern "C" int spb_preproc_max_taps(void) { return KMAX; }

extern "C" int spb_preproc_batch(const spb_preproc_args_t* a, spb_stream_t stream) {
  if (!a || !a->src || !a->table || !a->ftable || !a->out || !a->bounds || !a->coeffs || !a->tmp) return SPB_E_ARG;
  if (a->B <= 0 || a->S <= 0 || a->max_h <= 0 || (a->C != 1 && a->C != 3)) return SPB_E_ARG;
  if ((a->flags_any & 2) && !a->noise) return SPB_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pp_coeffs_kernel, dim3(a->B, 2), dim3(256), 0, s, a->table, a->bounds, a->coeffs, a->S);
  const dim3 gh((unsigned)(((long long)a->max_h * a->S + 255) / 256), a->B);
  const dim3 gf((unsigned)((a->S * a->S + 255) / 256), a->B);
  if (a->C == 3) {
    hipLaunchKernelGGL(pp_hpass_kernel<3>, gh, dim3(256), 0, s, (const unsigned char*)a->src, a->table, a->bounds, a->coeffs,
                       (unsigned char*)a->tmp, a->S, a->max_h);
    hipLaunchKernelGGL(pp_finish_kernel<3>, gf, dim3(256), 0, s, (const unsigned char*)a->tmp, a->table, a->ftable, a->bounds, a->coeffs,
                       a->noise, a->out, a->S, a->max_h, a->noise_std);
  } else {
    hipLaunchKernelGGL(pp_hpass_kernel<1>, gh, dim3(256), 0, s, (const unsigned char*)a->src, a->table, a->bounds, a->coeffs,
                       (unsigned char*)a->tmp, a->S, a->max_h);
    hipLaunchKernelGGL(pp_finish_kernel<1>, gf, dim3(256), 0, s, (const unsigned char*)a->tmp, a->table, a->ftable, a->bounds, a->coeffs,
                       a->noise, a->out, a->S, a->max_h, a->noise_std);
  }
  SPB_CHECK_LAUNCH();
  return 0;
}
