// micro-benchmark / phase timing of the depthwise forward kernel and plain copy kernels (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ unsigned long long* g_ts;
#define SPB_TS(i) do { if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0) g_ts_buf[(blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
__device__ unsigned long long g_ts_buf[64 * 8];
#include "../speedplusbaseline_amd/csrc/dwconv.hip"

__global__ void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // copy bandwidth at several sizes / grids
  for (size_t mb : {8, 32, 128, 512}) {
    size_t bytes = mb << 20; void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    for (int grid : {256, 1024, 4096, 16384}) {
      hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
      CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("copy %4zu MB grid %6d: %8.2f us  %7.1f GB/s (r+w)\n", mb, grid, ms * 100, 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  // dw forward on KRN shapes
  struct Sh { int B, H, C, st; } shapes[] = {{48, 112, 96, 2}, {48, 56, 144, 1}, {48, 14, 384, 1}, {48, 7, 960, 1}};
  for (auto sh : shapes) {
    const int OH = (sh.H - 1) / sh.st + 1;
    size_t nin = (size_t)sh.B * sh.H * sh.H * sh.C, nout = (size_t)sh.B * OH * OH * sh.C;
    void *x, *y; float *w, *sums, *gam, *bet, *osums;
    CK(hipMalloc(&x, nin * 2)); CK(hipMalloc(&y, nout * 2)); CK(hipMalloc(&w, sh.C * 9 * 4)); CK(hipMalloc(&sums, 16 * sh.C * 4));
    CK(hipMalloc(&gam, sh.C * 4)); CK(hipMalloc(&bet, sh.C * 4)); CK(hipMalloc(&osums, 16 * sh.C * 4));
    CK(hipMemset(x, 0, nin * 2)); CK(hipMemset(w, 0, sh.C * 36)); CK(hipMemset(sums, 0, 64 * sh.C)); CK(hipMemset(gam, 0, sh.C * 4)); CK(hipMemset(bet, 0, sh.C * 4)); CK(hipMemset(osums, 0, 64 * sh.C));
    spb_dw_args_t a; std::memset(&a, 0, sizeof(a));
    a.X = x; a.Wd = w; a.Y = y; a.osums = osums; a.oR = 8; a.epi_mode = 1; a.B = sh.B; a.H = sh.H; a.W = sh.H; a.C = sh.C; a.stride = sh.st;
    a.pro.sums = sums; a.pro.gamma = gam; a.pro.beta = bet; a.pro.inv_n = 1.f; a.pro.eps = 1e-5f; a.pro.C = sh.C; a.pro.R = 8; a.pro.act = SPB_ACT_RELU6;
    for (int r = 0; r < 3; ++r) spb_dwconv_fwd(SPB_BF16, &a, 0);
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) spb_dwconv_fwd(SPB_BF16, &a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ts[64 * 8]; CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts_buf), sizeof(ts)));
    printf("dw_fwd B%d H%d C%d s%d: %8.2f us  %7.1f GB/s | wg0 phases (100MHz ticks): prologue %llu, first tile load %llu, compute %llu, last->end %llu, total %llu\n",
           sh.B, sh.H, sh.C, sh.st, ms * 100, (nin + nout) * 2.0 / (ms / 10 * 1e-3) / 1e9, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[4] - ts[0]);
  }
  return 0;
}
