// phase timeline of the KRN head kernels (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/ubench_head.hip -o scratch/ubench_head
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_ts;
#define SPB_TS_DECL unsigned long long ts_r[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SPB_TSR(i) ts_r[i] = wall_clock64()
#define SPB_TS_FLUSH do { if (threadIdx.x == 0 && g_ts && blockIdx.x < 4096) for (int i_ = 0; i_ < 8; ++i_) g_ts[blockIdx.x * 8 + i_] = ts_r[i_]; } while (0)
#include "../speedplusbaseline_amd/csrc/stem_head.hip"
int spb_stem_fwd_mfma(const float*, const float*, void*, float*, int, int, int, int, hipStream_t) { return 0; }
int spb_stem_wgrad_mfma(const float*, const void*, const void*, const spb_bnref_t*, float*, int, int, int, hipStream_t) { return 0; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static void timeline(unsigned long long* ts, const char* what) {
  static unsigned long long h[4096 * 8]; hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull; int nb = 0;
  for (int b = 0; b < 4096; ++b) if (h[b * 8]) { if (h[b * 8] < t0) t0 = h[b * 8]; nb++; }
  double avg[8] = {0}, mx[8] = {0};
  for (int b = 0; b < 4096; ++b) if (h[b * 8]) for (int i = 0; i < 8; ++i) { double v = h[b * 8 + i] ? (double)(h[b * 8 + i] - t0) / 100.0 : 0.0; avg[i] += v / nb; if (v > mx[i]) mx[i] = v; }
  printf("  %s %d wgs:", what, nb);
  for (int i = 0; i < 8; ++i) printf(" t%d %.1f/%.1f", i, avg[i], mx[i]);
  printf("\n");
}
int main() {
  const int B = 48, J = 22, Jp = 32, HW = 49, C = 1024, KH = HW * C;
  const int S = getenv("S") ? atoi(getenv("S")) : 512;
  void *Z, *Wp, *G; float *bias, *target, *partial, *pred, *dout, *scal, *sums, *gam, *bet, *osums, *dW, *dbias;
  CK(hipMalloc(&Z, (size_t)B * KH * 2)); CK(hipMalloc(&Wp, (size_t)Jp * KH * 2)); CK(hipMalloc(&G, (size_t)B * KH * 2));
  CK(hipMalloc(&bias, 128)); CK(hipMalloc(&target, B * J * 4)); CK(hipMalloc(&partial, (size_t)S * B * Jp * 4)); CK(hipMalloc(&pred, B * J * 4));
  CK(hipMalloc(&dout, B * J * 4)); CK(hipMalloc(&scal, 16)); CK(hipMalloc(&sums, 2 * C * 4)); CK(hipMalloc(&gam, C * 4)); CK(hipMalloc(&bet, C * 4));
  CK(hipMalloc(&osums, 2 * C * 4)); CK(hipMalloc(&dW, (size_t)J * KH * 4)); CK(hipMalloc(&dbias, 128));
  CK(hipMemset(Z, 0, (size_t)B * KH * 2)); CK(hipMemset(Wp, 0, (size_t)Jp * KH * 2)); CK(hipMemset(bias, 0, 128)); CK(hipMemset(target, 0, B * J * 4));
  CK(hipMemset(partial, 0, (size_t)S * B * Jp * 4)); CK(hipMemset(sums, 0, 2 * C * 4)); CK(hipMemset(gam, 0, C * 4)); CK(hipMemset(bet, 0, C * 4));
  CK(hipMemset(osums, 0, 2 * C * 4)); CK(hipMemset(dW, 0, (size_t)J * KH * 4)); CK(hipMemset(dbias, 0, 128)); CK(hipMemset(dout, 0, B * J * 4));
  spb_head_args_t a; std::memset(&a, 0, sizeof(a));
  a.Z = Z; a.Wp = Wp; a.bias = bias; a.target = target; a.partial = partial; a.pred = pred; a.dout = dout; a.scalars = scal;
  a.pro.sums = sums; a.pro.bsums = sums; a.pro.gamma = gam; a.pro.beta = bet; a.pro.inv_n = 1.f / (B * HW); a.pro.eps = 1e-5f; a.pro.C = C; a.pro.R = 1; a.pro.act = SPB_ACT_RELU;
  a.B = B; a.J = J; a.Jp = Jp; a.HW = HW; a.C = C; a.S = S;
  spb_head_bwd_args_t b; std::memset(&b, 0, sizeof(b));
  b.Z = Z; b.Wp = Wp; b.dout = dout; b.G = G; b.osums = osums; b.dW = dW; b.dbias = dbias; b.pro = a.pro; b.gscale = 1.f;
  b.B = B; b.J = J; b.Jp = Jp; b.HW = HW; b.C = C; b.oR = 1; b.roles = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* ts; CK(hipMalloc(&ts, 4096 * 64));
  for (int which = 0; which < 3; ++which) {
    if (which == 2) b.roles = 2;
    auto run = [&]() { return which == 0 ? spb_head_fwd(SPB_BF16, &a, 0) : spb_head_bwd(SPB_BF16, &b, 0); };
    for (int r = 0; r < 3; ++r) { int e = run(); if (e) { printf("launch err %d\n", e); return 1; } }
    CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.2f us per call\n", which == 0 ? "head_fwd" : (which == 1 ? "head_bwd (input gradient)" : "head_bwd (weight gradient)"), ms * 50);
    CK(hipMemset(ts, 0, 4096 * 64)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
    run(); CK(hipDeviceSynchronize());
    timeline(ts, "timeline");
    unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &nul, sizeof(nul)));
  }
  return 0;
}
