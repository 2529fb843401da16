// how many workgroups of the tiled pointwise GEMM instances fit a CU (hipOccupancyMaxActiveBlocksPerMultiprocessor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../speedplusbaseline_amd/csrc/gemm_pw.hip"
extern "C" int spb_partial_reduce(const spb_red_job_t*, int, spb_stream_t) { return 0; }
int spb_gemm_os(const spb_gemm_args_t*, hipStream_t) { return SPB_E_UNSUPPORTED; }
int spb_gemm_sk(const spb_gemm_args_t*, hipStream_t) { return SPB_E_UNSUPPORTED; }
int spb_gemm_big(const spb_gemm_args_t*, hipStream_t) { return SPB_E_UNSUPPORTED; }
template <typename K> void probe(const char* name, K k, size_t lds) {
  int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
  hipFuncAttributes a; hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k));
  printf("%-40s lds %6zu -> %d blocks/CU (err %d)  regs %d  static lds %zu\n", name, lds, n, (int)e, a.numRegs, a.sharedSizeBytes);
}
int main() {
  probe("pw_gemm<1,64,32,2,2> K=96", pw_gemm_kernel<bf16_t, 1, 64, 32, 2, 2>, (3 * 96 + 128) * 4 + 16384);
  probe("pw_gemm<1,64,32,1,1> K=96", pw_gemm_kernel<bf16_t, 1, 64, 32, 1, 1>, (3 * 96) * 4 + 16384);
  probe("pw_gemm<2,32,32,1,1> K=32", pw_gemm_kernel<bf16_t, 2, 32, 32, 1, 1>, (3 * 32) * 4 + 16384);
  probe("pw_gemm<1,64,32,2,2> lds 8K", pw_gemm_kernel<bf16_t, 1, 64, 32, 2, 2>, 8192);
  return 0;
}
