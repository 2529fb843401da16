// ablation timing of the wide decoder convolution (128->128 3x3 at 56x56, B=48); not part of the product
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DGABL=<bits> scratch/ubench_gconv.hip -o scratch/ubench_gconv_<bits>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define GC_TS 1
#include "../speedplusbaseline_amd/csrc/ghiasi.hip"
#define GWABL GABL
#define GW_TS 1
#include "../speedplusbaseline_amd/csrc/ghiasi_wide.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Sh { int B, H, Cin, Cout, K, st, up; } shapes[] = {{48, 56, 128, 128, 3, 1, 1}, {48, 56, 128, 64, 3, 1, 2}, {48, 112, 64, 32, 3, 1, 2}, {48, 224, 32, 3, 9, 1, 1}, {48, 224, 32, 64, 3, 2, 1}, {48, 112, 64, 128, 3, 2, 1}};
  if (getenv("PF")) spb_debug_set_gconv_slab_pf(atoi(getenv("PF")));
  if (getenv("SLAB")) spb_debug_set_gconv_slab(atoi(getenv("SLAB")));
  if (getenv("HPRE")) spb_debug_set_gconv_halo_prefetch(atoi(getenv("HPRE")));
  if (getenv("ROT")) spb_debug_set_gconv_wide_rotate(atoi(getenv("ROT")));
  if (getenv("DELAY")) spb_debug_set_gconv_wide_delay(atoi(getenv("DELAY")));
  if (getenv("WGS")) spb_debug_set_gconv_wide_wgs(atoi(getenv("WGS")));
  if (getenv("WPXG")) spb_debug_set_gconv_wlds_pxg(atoi(getenv("WPXG")));
  printf("GABL=%d PF=%s SLAB=%s WPXG=%s\n", GABL, getenv("PF") ? getenv("PF") : "-", getenv("SLAB") ? getenv("SLAB") : "-", getenv("WPXG") ? getenv("WPXG") : "-");
  for (auto sh : shapes) {
    if (getenv("UB")) sh.B = atoi(getenv("UB"));
    const int Hout = sh.H * sh.up / sh.st;
    size_t nin = (size_t)sh.B * sh.H * sh.H * sh.Cin, nout = (size_t)sh.B * Hout * Hout * (sh.Cout < 4 ? 4 : sh.Cout);
    void *x, *w, *y; float *coef, *stats, *bias;
    CK(hipMalloc(&x, nin * 2)); CK(hipMalloc(&w, (size_t)sh.Cout * sh.K * sh.K * sh.Cin * 2)); CK(hipMalloc(&y, nout * 2));
    CK(hipMalloc(&coef, sh.B * sh.Cin * 8)); CK(hipMalloc(&stats, sh.B * 128 * 8)); CK(hipMalloc(&bias, 512));
    CK(hipMemset(x, 0, nin * 2)); CK(hipMemset(w, 0, (size_t)sh.Cout * sh.K * sh.K * sh.Cin * 2)); CK(hipMemset(coef, 0, sh.B * sh.Cin * 8));
    CK(hipMemset(stats, 0, sh.B * 128 * 8)); CK(hipMemset(bias, 0, 512));
    spb_gconv_args_t a; std::memset(&a, 0, sizeof(a));
    a.X = x; a.W = w; a.bias = bias; a.coef = coef; a.Y = y; a.stats = stats; a.B = sh.B; a.Hin = sh.H; a.Win = sh.H; a.Cin = sh.Cin;
    a.Cout = sh.Cout; a.KH = sh.K; a.stride = sh.st; a.upsample = sh.up; a.relu = 1; a.ldc = sh.Cout < 4 ? 4 : sh.Cout;
    const bool up2 = getenv("UP2") && sh.up == 2 && sh.K == 3;
    if (up2) {
      for (int k = 0; k < 3; ++k) if (spb_gconv_up2(SPB_BF16, &a, 0)) { printf("up2 launch failed\n"); return 1; }
      CK(hipEventRecord(e0)); for (int k = 0; k < 10; ++k) spb_gconv_up2(SPB_BF16, &a, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
      printf("up2   %3d->%3d @%3d: %8.2f us\n", sh.Cin, sh.Cout, Hout, ms2 * 100);
      unsigned long long* ts; CK(hipMalloc(&ts, 1024 * 16 * 8)); CK(hipMemset(ts, 0, 1024 * 16 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &ts, sizeof(ts)));
      spb_gconv_up2(SPB_BF16, &a, 0); CK(hipDeviceSynchronize());
      static unsigned long long h[1024 * 16]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
      unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &nul, sizeof(nul)));
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < 1024 * 16; ++i) if (h[i] && h[i] < t0) t0 = h[i];
      for (int ti = 0; ti < 4; ++ti) {
        double av[3] = {0, 0, 0}; int n = 0;
        for (int w = 0; w < 1024; ++w) if (h[(w * 4 + ti) * 4]) { ++n; for (int i = 0; i < 3; ++i) av[i] += (double)(h[(w * 4 + ti) * 4 + i] - t0) / 100.0; }
        if (n) printf("  tile group %d (%4d wgs): top %.2f  committed %.2f  loop done %.2f us\n", ti, n, av[0] / n, av[1] / n, av[2] / n);
      }
      hipFree(ts);
    }
    if (getenv("GS") && sh.st == 2 && sh.Cin == 32) {
      unsigned long long* ts; CK(hipMalloc(&ts, 1024 * 16 * 8)); CK(hipMemset(ts, 0, 1024 * 16 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &ts, sizeof(ts)));
      spb_gconv(SPB_BF16, &a, 0); CK(hipDeviceSynchronize());
      static unsigned long long hs[1024 * 16]; CK(hipMemcpy(hs, ts, sizeof(hs), hipMemcpyDeviceToHost));
      unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &nul, sizeof(nul)));
      for (int ti = 0; ti < 4; ++ti) {
        double av[3] = {0, 0, 0}; int n = 0;
        for (int w = 0; w < 512; ++w) if (hs[(w * 4 + ti) * 4]) { ++n; for (int i = 0; i < 3; ++i) av[i] += (double)(hs[(w * 4 + ti) * 4 + i] - hs[(w * 4) * 4]) / 100.0; }
        if (n) printf("  32->64 s2 tile %d (%4d wgs, us from the workgroup's first stamp): top %.2f  committed %.2f  loop done %.2f\n", ti, n, av[0] / n, av[1] / n, av[2] / n);
      }
      hipFree(ts);
    }
    if (getenv("K9") && sh.K == 9) {
      unsigned long long* ts; CK(hipMalloc(&ts, 512 * 8 * 8)); CK(hipMemset(ts, 0, 512 * 8 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &ts, sizeof(ts)));
      spb_gconv(SPB_BF16, &a, 0); CK(hipDeviceSynchronize());
      static unsigned long long h9[512 * 8]; CK(hipMemcpy(h9, ts, sizeof(h9), hipMemcpyDeviceToHost));
      unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gc_ts), &nul, sizeof(nul)));
      double av[7] = {0}; int n = 0;
      for (int w = 0; w < 512; ++w) if (h9[w * 8]) { ++n; for (int i = 0; i < 7; ++i) av[i] += (double)(h9[w * 8 + i] - h9[w * 8]) / 100.0; }
      if (n) printf("  9x9 band kernel (%d wgs): weights %.2f  halo %.2f  taps %.2f  barrier %.2f  P written %.2f  sums+stores %.2f  end %.2f us\n", n, av[1] / n, av[2] / n, av[3] / n, av[3] / n, av[4] / n, av[5] / n, av[6] / n);
      hipFree(ts);
    }
    const bool wide = getenv("WIDE") && sh.Cin == 128 && sh.Cout == 128 && sh.up == 1;
    for (int k = 0; k < 3; ++k) if (wide ? spb_gconv_wide(SPB_BF16, &a, 0) : spb_gconv(SPB_BF16, &a, 0)) { printf("launch failed\n"); return 1; }
    CK(hipEventRecord(e0)); for (int k = 0; k < 10; ++k) { if (wide) spb_gconv_wide(SPB_BF16, &a, 0); else spb_gconv(SPB_BF16, &a, 0); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (wide) {
      unsigned long long* ts; CK(hipMalloc(&ts, 1024 * 4 * 5 * 8)); CK(hipMemset(ts, 0, 1024 * 4 * 5 * 8));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gw_ts), &ts, sizeof(ts)));
      spb_gconv_wide(SPB_BF16, &a, 0); CK(hipDeviceSynchronize());
      static unsigned long long h[1024 * 4 * 5]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
      unsigned long long* nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gw_ts), &nul, sizeof(nul)));
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < 1024 * 20; ++i) if (h[i] && h[i] < t0) t0 = h[i];
      for (int gi = 0; gi < 3; ++gi) {
        double av[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int w = 0; w < 1024; ++w) if (h[(w * 4 + gi) * 5]) { ++n; for (int i = 0; i < 5; ++i) av[i] += (double)(h[(w * 4 + gi) * 5 + i] - t0) / 100.0; }
        if (n) printf("  group %d (%4d wgs): start %.2f  committed %.2f  loop done %.2f  all waves done %.2f  end %.2f us\n", gi, n, av[0] / n, av[1] / n, av[2] / n, av[3] / n, av[4] / n);
      }
      for (int w = 0; w < 3; ++w) { printf("  wg %d:", w * 8); for (int gi = 0; gi < 3; ++gi) for (int i = 0; i < 5; ++i) printf(" %.2f", h[(w * 32 + gi) * 5 + i] ? (double)(h[(w * 32 + gi) * 5 + i] - t0) / 100.0 : 0.0); printf("\n"); }
      hipFree(ts);
    }
    const double fl = 2.0 * sh.B * Hout * Hout * sh.Cout * sh.Cin * sh.K * sh.K;
    printf("gconv %dx%d %3d->%3d s%d u%d @%3d: %8.2f us  %7.1f TFLOP/s\n", sh.K, sh.K, sh.Cin, sh.Cout, sh.st, sh.up, Hout, ms * 100, fl / (ms / 10 * 1e-3) / 1e12);
    hipFree(x); hipFree(w); hipFree(y); hipFree(coef); hipFree(stats); hipFree(bias);
  }
  return 0;
}
