// pw_big experiments: timing + phase timeline for chosen (M, K, N) (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBIG_EXP_NOMFMA] [-DBIG_EXP_NOXF] scratch/ubench_big.hip -o scratch/ubench_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_ts;
#define SPB_TS_DECL unsigned long long ts_r[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SPB_TSR(i) ts_r[i] = wall_clock64()
#define SPB_TS_FLUSH do { if (threadIdx.x == 0 && g_ts && blockIdx.x < 4096) for (int i_ = 0; i_ < 8; ++i_) g_ts[blockIdx.x * 8 + i_] = ts_r[i_]; } while (0)
#define SPB_STAT_BEGIN unsigned long long st_b = __builtin_readcyclecounter()
#define SPB_STAT_END(i) ts_r[i] += __builtin_readcyclecounter() - st_b
#define SPB_STAT_BEGIN2 unsigned long long st_c = __builtin_readcyclecounter()
#define SPB_STAT_END2(i) ts_r[i] += __builtin_readcyclecounter() - st_c
#include "../speedplusbaseline_amd/csrc/gemm_big.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Sh { int M, K, N; };
  std::vector<Sh> shapes;
  for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
  if (shapes.empty()) shapes = {{2352, 320, 1024}, {2352, 1024, 1024}, {2352, 1280, 1024}, {1176, 1024, 1024}, {4704, 1024, 1024}, {2352, 1024, 512}};
  unsigned long long* ts; CK(hipMalloc(&ts, 4096 * 64));
  for (int dir = 0; dir < 2; ++dir)
  for (auto f : shapes) {
    Sh sh = f;
    if (dir == 1) { sh.K = f.N; sh.N = f.K; }
    const int pro = dir == 0 ? 1 : 2, epi = dir == 0 ? 1 : 2;
    size_t na = (size_t)sh.M * sh.K, ny = (size_t)sh.M * sh.N;
    void *A, *A2, *W, *Y, *Zo; float *sums, *gam, *bet, *osums, *esums;
    CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&A2, na * 2)); CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&Y, ny * 2)); CK(hipMalloc(&Zo, ny * 2));
    int C = sh.K > sh.N ? sh.K : sh.N;
    CK(hipMalloc(&sums, 32 * C * 4)); CK(hipMalloc(&gam, C * 4)); CK(hipMalloc(&bet, C * 4)); CK(hipMalloc(&osums, 32 * C * 4)); CK(hipMalloc(&esums, 32 * C * 4));
    CK(hipMemset(A, 0, na * 2)); CK(hipMemset(A2, 0, na * 2)); CK(hipMemset(W, 0, (size_t)sh.N * sh.K * 2)); CK(hipMemset(Zo, 0, ny * 2));
    CK(hipMemset(sums, 0, 128 * C)); CK(hipMemset(gam, 0, C * 4)); CK(hipMemset(bet, 0, C * 4)); CK(hipMemset(osums, 0, 128 * C)); CK(hipMemset(esums, 0, 128 * C));
    spb_gemm_args_t g; std::memset(&g, 0, sizeof(g));
    g.A = A; g.A2 = pro == 2 ? A2 : nullptr; g.Bw = W; g.Y = Y; g.Zout = epi == 2 ? Zo : nullptr; g.osums = osums; g.oR = 1;
    g.M = sh.M; g.K = sh.K; g.N = sh.N; g.pro_mode = pro; g.epi_mode = epi; g.out_scale = 1.f;
    g.pro.sums = sums; g.pro.bsums = sums; g.pro.gamma = gam; g.pro.beta = bet; g.pro.inv_n = 1.f; g.pro.eps = 1e-5f; g.pro.C = sh.K; g.pro.R = 1; g.pro.act = SPB_ACT_RELU6;
    g.epi = g.pro; g.epi.sums = esums; g.epi.C = sh.N;
    bool bad = false; for (int r = 0; r < 3; ++r) { int e = spb_gemm_big(&g, 0); if (e) { printf("%s M%6d K%5d N%5d: unsupported (%d)\n", dir ? "dgrad" : "fwd  ", sh.M, sh.K, sh.N, e); bad = true; break; } }
    if (bad) continue;
    CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) spb_gemm_big(&g, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * sh.M * sh.K * sh.N;
    printf("%s M%6d K%5d N%5d: %7.2f us %6.0f TFLOP/s", dir ? "dgrad" : "fwd  ", sh.M, sh.K, sh.N, ms * 50, fl / (ms / 20 * 1e-3) / 1e12);
    CK(hipMemset(ts, 0, 4096 * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
    spb_gemm_big(&g, 0); CK(hipDeviceSynchronize());
    static unsigned long long h[4096 * 8]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &nul, sizeof(nul)));
    unsigned long long t0 = ~0ull; int nb = 0;
    for (int b = 0; b < 4096; ++b) if (h[b * 8]) { if (h[b * 8] < t0) t0 = h[b * 8]; nb++; }
    if (nb) {
      double avg[6] = {0}, mx[6] = {0};
      for (int b = 0; b < 4096; ++b) if (h[b * 8]) for (int i = 0; i < 6; ++i) { double v = h[b * 8 + i] ? (double)(h[b * 8 + i] - t0) / 100.0 : 0.0; avg[i] += v / nb; if (v > mx[i]) mx[i] = v; }
      printf(" | %4d wgs: start %.1f/%.1f  prologue %.1f/%.1f  kloop %.1f/%.1f  epi %.1f/%.1f  end %.1f/%.1f", nb, avg[0], mx[0], avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4]);
      double w5 = 0, w6 = 0, w7 = 0; for (int b = 0; b < 4096; ++b) if (h[b * 8]) { w5 += (double)h[b * 8 + 5] / nb; w6 += (double)h[b * 8 + 6] / nb; w7 += (double)h[b * 8 + 7] / nb; }
      printf("  [wave 0 cycles waiting: dma %.0f  lds %.0f  barrier %.0f]", w5, w6, w7);
    }
    printf("\n");
    hipFree(A); hipFree(A2); hipFree(W); hipFree(Y); hipFree(Zo); hipFree(sums); hipFree(gam); hipFree(bet); hipFree(osums); hipFree(esums);
  }
  return 0;
}
