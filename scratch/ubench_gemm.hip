// phase timing of the pointwise GEMM kernel on KRN layer shapes (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ unsigned long long g_ts_buf[64 * 8];
#define SPB_TS(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_ts_buf[(blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#include "../speedplusbaseline_amd/csrc/gemm_pw.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Sh { int M, K, N, pro, epi; const char* what; } shapes[] = {
    {9408, 384, 64, 1, 1, "fwd 384->64 @14"}, {9408, 64, 384, 1, 1, "fwd 64->384 @14"}, {2352, 960, 160, 1, 1, "fwd 960->160 @7"},
    {2352, 160, 960, 1, 1, "fwd 160->960 @7"}, {2352, 1280, 1024, 1, 1, "fwd 1280->1024 @7"}, {602112, 16, 96, 1, 1, "fwd 16->96 @112"},
    {150528, 144, 24, 1, 1, "fwd 144->24 @56"},
    {2352, 1024, 1280, 2, 2, "dgrad 1280<-1024 @7"}, {9408, 64, 384, 2, 2, "dgrad 384<-64 @14"}, {602112, 96, 16, 2, 2, "dgrad 16<-96 @112"}};
  for (auto sh : shapes) {
    size_t na = (size_t)sh.M * sh.K, ny = (size_t)sh.M * sh.N;
    void *A, *A2, *W, *Y, *Zo; float *sums, *gam, *bet, *osums, *esums;
    CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&A2, na * 2)); CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&Y, ny * 2)); CK(hipMalloc(&Zo, ny * 2));
    int C = sh.K > sh.N ? sh.K : sh.N;
    CK(hipMalloc(&sums, 32 * C * 4)); CK(hipMalloc(&gam, C * 4)); CK(hipMalloc(&bet, C * 4)); CK(hipMalloc(&osums, 32 * C * 4)); CK(hipMalloc(&esums, 32 * C * 4));
    CK(hipMemset(A, 0, na * 2)); CK(hipMemset(A2, 0, na * 2)); CK(hipMemset(W, 0, (size_t)sh.N * sh.K * 2)); CK(hipMemset(Zo, 0, ny * 2));
    CK(hipMemset(sums, 0, 128 * C)); CK(hipMemset(gam, 0, C * 4)); CK(hipMemset(bet, 0, C * 4)); CK(hipMemset(osums, 0, 128 * C)); CK(hipMemset(esums, 0, 128 * C));
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = A; g.A2 = sh.pro == 2 ? A2 : nullptr; g.Bw = W; g.Y = Y; g.Zout = sh.epi == 2 ? Zo : nullptr; g.osums = osums; g.oR = sh.M >= 32768 ? 8 : 1;
    g.M = sh.M; g.K = sh.K; g.N = sh.N; g.pro_mode = sh.pro; g.epi_mode = sh.epi; g.out_scale = 1.f;
    g.pro.sums = sums; g.pro.bsums = sums; g.pro.gamma = gam; g.pro.beta = bet; g.pro.inv_n = 1.f; g.pro.eps = 1e-5f; g.pro.C = sh.K; g.pro.R = g.oR; g.pro.act = SPB_ACT_RELU6;
    g.epi = g.pro; g.epi.sums = esums; g.epi.C = sh.N;
    for (int r = 0; r < 3; ++r) { int e = spb_pwconv_gemm(SPB_BF16, &g, 0); if (e) { printf("launch err %d\n", e); return 1; } }
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) spb_pwconv_gemm(SPB_BF16, &g, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long ts[64 * 8]; CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts_buf), sizeof(ts)));
    double bytes = (double)(na * (sh.pro == 2 ? 2 : 1) + ny * (sh.epi == 2 ? 2 : 1)) * 2;
    printf("%-22s M%6d K%5d N%5d: %8.2f us %7.1f GB/s | wg0 ticks(10ns): prologue %llu, K-loop(last tile) %llu, epilogue %llu, tail %llu, total %llu\n",
           sh.what, sh.M, sh.K, sh.N, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9, ts[1] - ts[0], ts[2] - (ts[3] > ts[1] && ts[3] < ts[2] ? ts[3] : ts[1]), ts[3] - ts[2], ts[4] - ts[3], ts[4] - ts[0]);
    hipFree(A); hipFree(A2); hipFree(W); hipFree(Y); hipFree(Zo); hipFree(sums); hipFree(gam); hipFree(bet); hipFree(osums); hipFree(esums);
  }
  return 0;
}
