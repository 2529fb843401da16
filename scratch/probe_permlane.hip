// semantics probe: v_permlane32_swap / v_permlane16_swap based xor-32 / xor-16 reductions vs __shfl_xor (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xsum32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float xsum16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float asum32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float asum16(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k2(const float* in, float* out) {
  const float v = in[threadIdx.x];
  out[threadIdx.x] = asum32(v);
  out[64 + threadIdx.x] = asum16(v);
}
__global__ void k(const float* in, float* out) {
  const float v = in[threadIdx.x];
  out[threadIdx.x] = xsum32(v);
  out[64 + threadIdx.x] = xsum16(v);
  out[128 + threadIdx.x] = v + __shfl_xor(v, 32, 64);
  out[192 + threadIdx.x] = v + __shfl_xor(v, 16, 64);
}
int main() {
  float h[64], o[256], *di, *dout;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 20)) + i * 0.001f;
  hipMalloc(&di, 256); hipMalloc(&dout, 1024);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
  int bad32 = 0, bad16 = 0;
  for (int i = 0; i < 64; ++i) { bad32 += o[i] != o[128 + i]; bad16 += o[64 + i] != o[192 + i]; }
  printf("permlane32_swap xor-sum mismatches %d, permlane16_swap mismatches %d\n", bad32, bad16);
  for (int i = 0; i < 4; ++i) printf("lane %d: p32 %g ref %g | p16 %g ref %g\n", i * 17, o[i * 17], o[128 + i * 17], o[64 + i * 17], o[192 + i * 17]);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  bad32 = bad16 = 0;
  for (int i = 0; i < 64; ++i) { bad32 += o[i] != o[128 + i]; bad16 += o[64 + i] != o[192 + i]; }
  printf("inline asm: permlane32_swap xor-sum mismatches %d, permlane16_swap mismatches %d\n", bad32, bad16);
  for (int i = 0; i < 4; ++i) printf("lane %d: p32 %g ref %g | p16 %g ref %g\n", i * 17, o[i * 17], o[128 + i * 17], o[64 + i * 17], o[192 + i * 17]);
  return 0;
}
