// phase ablation of the row-unit depthwise backward kernels on the 112x112 / 56x56 KRN layers (not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSPB_ABL=<bits> scratch/ubench_dw3.hip speedplusbaseline_amd/csrc/dwconv_plane.hip -o scratch/ubench_dw3_<bits>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../speedplusbaseline_amd/csrc/dwconv_rows.hip"
extern "C" int spb_debug_set_dw_rows(int);
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Sh { int B, H, C, st; } shapes[] = {{48, 112, 32, 1}, {48, 112, 96, 2}, {48, 56, 144, 1}, {48, 56, 144, 2}};
  if (getenv("DW_ROWS")) spb_debug_set_dw_rows(atoi(getenv("DW_ROWS")));
  if (getenv("DW_BLOCKS")) spb_debug_set_dw_rows(-atoi(getenv("DW_BLOCKS")));
  const bool split = getenv("DW_NOWG") != nullptr;
  printf("SPB_ABL=%d blocks=%s rows=%s wg=%d\n", SPB_ABL, getenv("DW_BLOCKS") ? getenv("DW_BLOCKS") : "-", getenv("DW_ROWS") ? getenv("DW_ROWS") : "-", !split);
  for (auto sh : shapes) {
    const int OH = (sh.H - 1) / sh.st + 1;
    size_t nin = (size_t)sh.B * sh.H * sh.H * sh.C, nout = (size_t)sh.B * OH * OH * sh.C;
    void *g, *z, *zo, *y; float *w, *dw, *sums, *gam, *bet, *osums, *bsums;
    CK(hipMalloc(&g, nout * 2)); CK(hipMalloc(&z, nout * 2)); CK(hipMalloc(&zo, nin * 2)); CK(hipMalloc(&y, nin * 2));
    CK(hipMalloc(&w, sh.C * 36)); CK(hipMalloc(&dw, sh.C * 36)); CK(hipMalloc(&sums, 64 * sh.C)); CK(hipMalloc(&bsums, 64 * sh.C));
    CK(hipMalloc(&gam, sh.C * 4)); CK(hipMalloc(&bet, sh.C * 4)); CK(hipMalloc(&osums, 64 * sh.C));
    CK(hipMemset(g, 0, nout * 2)); CK(hipMemset(z, 0, nout * 2)); CK(hipMemset(zo, 0, nin * 2));
    CK(hipMemset(w, 0, sh.C * 36)); CK(hipMemset(dw, 0, sh.C * 36)); CK(hipMemset(sums, 0, 64 * sh.C)); CK(hipMemset(bsums, 0, 64 * sh.C));
    CK(hipMemset(gam, 0, sh.C * 4)); CK(hipMemset(bet, 0, sh.C * 4)); CK(hipMemset(osums, 0, 64 * sh.C));
    spb_dw_args_t a; std::memset(&a, 0, sizeof(a));
    a.X = g; a.X2 = z; a.Wd = w; a.Y = y; a.dW = split ? nullptr : dw; a.Zout = zo; a.osums = osums; a.oR = 8; a.epi_mode = 2;
    a.B = sh.B; a.H = sh.H; a.W = sh.H; a.C = sh.C; a.stride = sh.st;
    spb_bnref_t r; std::memset(&r, 0, sizeof(r));
    r.sums = sums; r.gamma = gam; r.beta = bet; r.bsums = bsums; r.inv_n = 1.f; r.eps = 1e-5f; r.C = sh.C; r.R = 8; r.act = SPB_ACT_RELU6;
    a.pro = r; a.epi = r; a.pro_in = r;
    for (int k = 0; k < 3; ++k) spb_dwr_bwd(SPB_BF16, &a, 0);
    CK(hipEventRecord(e0)); for (int k = 0; k < 10; ++k) spb_dwr_bwd(SPB_BF16, &a, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (2.0 * nout + 2.0 * nin) * 2.0;
    printf("dw_bwd B%d H%3d C%3d s%d: %8.2f us  %7.1f GB/s", sh.B, sh.H, sh.C, sh.st, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
    {
      spb_dw_args_t f; std::memset(&f, 0, sizeof(f));
      f.X = zo; f.Wd = w; f.Y = g; f.osums = osums; f.oR = 8; f.epi_mode = 1; f.B = sh.B; f.H = sh.H; f.W = sh.H; f.C = sh.C; f.stride = sh.st;
      f.pro = r;
      for (int k = 0; k < 3; ++k) spb_dwr_fwd(SPB_BF16, &f, 0);
      CK(hipEventRecord(e0)); for (int k = 0; k < 10; ++k) spb_dwr_fwd(SPB_BF16, &f, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("   | fwd %8.2f us  %7.1f GB/s\n", ms * 100, (nin + nout) * 2.0 / (ms / 10 * 1e-3) / 1e9);
    }
    hipFree(g); hipFree(z); hipFree(zo); hipFree(y);
  }
  return 0;
}
