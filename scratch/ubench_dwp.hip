// plane vs row-unit depthwise kernels on the KRN small-map layer shapes (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/ubench_dwp.hip -o scratch/ubench_dwp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
__device__ unsigned long long* g_ts;
#define SPB_PTS(i) { if (threadIdx.x == 0 && g_ts) g_ts[blockIdx.x * 8 + (i)] = wall_clock64(); }
#include "../speedplusbaseline_amd/csrc/dwconv_rows.hip"
#include "../speedplusbaseline_amd/csrc/dwconv_plane.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Sh { int B, H, C, st; } shapes[] = {{48, 28, 192, 1}, {48, 28, 192, 2}, {48, 14, 384, 1}, {48, 14, 576, 1}, {48, 14, 576, 2},
                                             {48, 7, 960, 1}, {48, 7, 320, 1}, {48, 7, 1024, 1}, {48, 7, 1280, 1}};
  const int B0 = getenv("UB") ? atoi(getenv("UB")) : 0;
  for (auto sh : shapes) {
    if (B0) sh.B = B0;
    const int OH = (sh.H - 1) / sh.st + 1;
    size_t nin = (size_t)sh.B * sh.H * sh.H * sh.C, nout = (size_t)sh.B * OH * OH * sh.C;
    void *g, *z, *zo, *rr, *y; float *w, *dw, *sums, *gam, *bet, *osums, *bsums;
    CK(hipMalloc(&g, nout * 2)); CK(hipMalloc(&z, nout * 2)); CK(hipMalloc(&zo, nin * 2)); CK(hipMalloc(&rr, nin * 2)); CK(hipMalloc(&y, nin * 2));
    CK(hipMalloc(&w, sh.C * 36)); CK(hipMalloc(&dw, sh.C * 36)); CK(hipMalloc(&sums, 64 * sh.C)); CK(hipMalloc(&bsums, 64 * sh.C));
    CK(hipMalloc(&gam, sh.C * 4)); CK(hipMalloc(&bet, sh.C * 4)); CK(hipMalloc(&osums, 64 * sh.C));
    CK(hipMemset(g, 0, nout * 2)); CK(hipMemset(z, 0, nout * 2)); CK(hipMemset(zo, 0, nin * 2)); CK(hipMemset(rr, 0, nin * 2));
    CK(hipMemset(w, 0, sh.C * 36)); CK(hipMemset(dw, 0, sh.C * 36)); CK(hipMemset(sums, 0, 64 * sh.C)); CK(hipMemset(bsums, 0, 64 * sh.C));
    CK(hipMemset(gam, 0, sh.C * 4)); CK(hipMemset(bet, 0, sh.C * 4)); CK(hipMemset(osums, 0, 64 * sh.C));
    spb_dw_args_t a; std::memset(&a, 0, sizeof(a));
    a.X = g; a.X2 = z; a.Wd = w; a.Y = y; a.dW = dw; a.Zout = zo; a.res = nullptr; a.osums = osums; a.oR = 1; a.epi_mode = 2;
    a.B = sh.B; a.H = sh.H; a.W = sh.H; a.C = sh.C; a.stride = sh.st;
    spb_bnref_t r; std::memset(&r, 0, sizeof(r));
    r.sums = sums; r.gamma = gam; r.beta = bet; r.bsums = bsums; r.inv_n = 1.f; r.eps = 1e-5f; r.C = sh.C; r.R = 1; r.act = SPB_ACT_RELU6;
    a.pro = r; a.epi = r; a.pro_in = r;
    spb_dw_args_t f; std::memset(&f, 0, sizeof(f));
    f.X = zo; f.Wd = w; f.Y = g; f.osums = osums; f.oR = 1; f.epi_mode = 1; f.B = sh.B; f.H = sh.H; f.W = sh.H; f.C = sh.C; f.stride = sh.st;
    f.pro = r;
    printf("B%d H%3d C%4d s%d:", sh.B, sh.H, sh.C, sh.st);
    for (int mode = 0; mode < 2; ++mode) {
      spb_debug_set_dw_mode(mode);
      float ms;
      for (int k = 0; k < 3; ++k) spb_dwconv_dgrad(SPB_BF16, &a, 0);
      CK(hipEventRecord(e0)); for (int k = 0; k < 20; ++k) spb_dwconv_dgrad(SPB_BF16, &a, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (2.0 * nout + 2.0 * nin) * 2.0;
      printf("  %s bwd %7.2f us %6.0f GB/s", mode ? "plane" : "rows ", ms * 50, bytes / (ms / 20 * 1e-3) / 1e9);
      { spb_dw_args_t d = a; d.dW = nullptr;   // input gradient alone (the launch-stream instance of the KRN plan)
        for (int k = 0; k < 3; ++k) spb_dwconv_dgrad(SPB_BF16, &d, 0);
        CK(hipEventRecord(e0)); for (int k = 0; k < 20; ++k) spb_dwconv_dgrad(SPB_BF16, &d, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf(" dgrad %6.2f", ms * 50);
        spb_dw_args_t wq = a; wq.Xin = a.Zout; wq.pro_in = a.epi;
        for (int k = 0; k < 3; ++k) spb_dwconv_wgrad(SPB_BF16, &wq, 0);
        CK(hipEventRecord(e0)); for (int k = 0; k < 20; ++k) spb_dwconv_wgrad(SPB_BF16, &wq, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf(" wgrad %6.2f", ms * 50); }
      for (int k = 0; k < 3; ++k) spb_dwconv_fwd(SPB_BF16, &f, 0);
      CK(hipEventRecord(e0)); for (int k = 0; k < 20; ++k) spb_dwconv_fwd(SPB_BF16, &f, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" fwd %7.2f us %6.0f GB/s |", ms * 50, (nin + nout) * 2.0 / (ms / 20 * 1e-3) / 1e9);
    }
    printf("\n");
    if (getenv("PTS")) {   // per-phase timeline of one backward launch (100 MHz wall clock -> us), over all workgroups
      unsigned long long* ts; CK(hipMalloc(&ts, 4096 * 64)); CK(hipMemset(ts, 0, 4096 * 64));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
      { spb_dw_args_t d = a; if (getenv("PTS_DG")) d.dW = nullptr; spb_dwconv_dgrad(SPB_BF16, &d, 0); } CK(hipDeviceSynchronize());
      static unsigned long long h[4096 * 8]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull; int nb = 0;
      for (int b = 0; b < 4096; ++b) if (h[b * 8]) { if (h[b * 8] < t0) t0 = h[b * 8]; nb++; }
      double avg[7] = {0}, mx[7] = {0};
      for (int b = 0; b < 4096; ++b) if (h[b * 8]) for (int i = 0; i < 7; ++i) { double v = (double)(h[b * 8 + i] - t0) / 100.0; avg[i] += v / nb; if (v > mx[i]) mx[i] = v; }
      printf("   bwd phases over %d workgroups (us since first start): ", nb);
      for (int i = 0; i < 7; ++i) printf(" p%d avg %.2f max %.2f |", i, avg[i], mx[i]);
      printf("\n");
      ts = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &ts, sizeof(ts)));
    }
    hipFree(g); hipFree(z); hipFree(zo); hipFree(rr); hipFree(y);
  }
  return 0;
}
