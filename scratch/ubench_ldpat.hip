// how fast can ~150 workgroups pull a 48-96 KB operand tile each?  load patterns of the small-M GEMMs (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../speedplusbaseline_amd/csrc/common.h"
// MODE 0: fragment layout (lane (i,q): row i, 16 B at k = 32c + 8q): 64 rows x K per workgroup, 16 rows per wave
// MODE 1: coalesced rows (lane l: row l>>2 (+16 per step), 16 B piece l&3 of the 64-byte chunk c)   -> registers
// MODE 2: the same addresses by LDS-DMA
// MODE 3: fully linear 16 B per lane over the tile (row-major, K*2 bytes per row contiguous)          -> registers
template <int MODE, int KC>
__global__ __launch_bounds__(256) void k(const bf16_t* A, float* out, int M, int K) {
  extern __shared__ char smem[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, li = l & 15, lq = l >> 4;
  const int m0 = blockIdx.x * 64;
  uint4 r[KC];
  float acc = 0.f;
  if (MODE == 0) {
    const size_t row = (size_t)min(m0 + w * 16 + li, M - 1) * K;
#pragma unroll
    for (int c = 0; c < KC; ++c) r[c] = *reinterpret_cast<const uint4*>(A + row + c * 32 + lq * 8);
  } else if (MODE == 1) {
    const size_t row = (size_t)min(m0 + w * 16 + (l >> 2), M - 1) * K;
#pragma unroll
    for (int c = 0; c < KC; ++c) r[c] = *reinterpret_cast<const uint4*>(A + row + c * 32 + (l & 3) * 8);
  } else if (MODE == 2) {
    const size_t row = (size_t)min(m0 + w * 16 + (l >> 2), M - 1) * K;
    const unsigned base = lds_addr(smem) + (unsigned)__builtin_amdgcn_readfirstlane(w * KC * 1024);
#pragma unroll
    for (int c = 0; c < KC; ++c) dma16(A + row + c * 32 + (l & 3) * 8, base + (unsigned)(c << 10));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < KC; ++c) r[c] = *reinterpret_cast<const uint4*>(smem + (w * KC + c) * 1024 + l * 16);
  } else {
    const char* base = reinterpret_cast<const char*>(A) + (size_t)m0 * K * 2 + (size_t)w * 16 * K * 2;
#pragma unroll
    for (int c = 0; c < KC; ++c) r[c] = *reinterpret_cast<const uint4*>(base + (size_t)(c * 64 + l) * 16);
  }
#pragma unroll
  for (int c = 0; c < KC; ++c) acc += __uint_as_float(r[c].x) + __uint_as_float(r[c].w);
  if (acc == 123.456f) out[t] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE, int KC> float run(const bf16_t* A, float* out, int M, int K, int lds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, KC>), dim3((M + 63) / 64), dim3(256), lds, 0, A, out, M, K);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, KC>), dim3((M + 63) / 64), dim3(256), lds, 0, A, out, M, K);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 50;
}
int main() {
  bf16_t* A; float* out; CK(hipMalloc(&A, 64 << 20)); CK(hipMalloc(&out, 4096)); CK(hipMemset(A, 0, 64 << 20));
  const int Ms[] = {9408, 37632};
  for (int M : Ms) {
    printf("M %d K 384 (12 chunks, 48 KB per workgroup, %d workgroups): frag %.2f us  rows %.2f us  dma %.2f us  linear %.2f us\n", M, (M + 63) / 64,
           run<0, 12>(A, out, M, 384, 0), run<1, 12>(A, out, M, 384, 0), run<2, 12>(A, out, M, 384, 4 * 12 * 1024), run<3, 12>(A, out, M, 384, 0));
    printf("M %d K 768 (24 chunks, 96 KB per workgroup): frag %.2f us  rows %.2f us  dma %.2f us  linear %.2f us\n", M,
           run<0, 24>(A, out, M, 768, 0), run<1, 24>(A, out, M, 768, 0), run<2, 24>(A, out, M, 768, 4 * 24 * 1024), run<3, 24>(A, out, M, 768, 0));
  }
  printf("empty-ish: %.2f us\n", run<3, 1>(A, out, 9408, 32, 0));
  return 0;
}
