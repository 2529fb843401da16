// occupancy of the depthwise plane / row kernels at their launch LDS sizes (hipOccupancyMaxActiveBlocksPerMultiprocessor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../speedplusbaseline_amd/csrc/dwconv_rows.hip"
#include "../speedplusbaseline_amd/csrc/dwconv_plane.hip"
template <typename K> void probe(const char* name, K k, size_t lds) {
  int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
  hipFuncAttributes a; hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k));
  printf("%-48s lds %6zu -> %d blocks/CU (err %d)  regs %d  static lds %zu\n", name, lds, n, (int)e, a.numRegs, a.sharedSizeBytes);
}
int main() {
  const size_t tb = (7 * 32 + 9 * 32 + 16 * 16) * 4;
  probe("dwp_bwd<1,false,true> 14x14 (16x16 tile)", dwp_bwd_kernel<bf16_t, 1, false, true>, tb + 256 * 128);
  probe("dwp_bwd<1,false,true> 7x7 NB=4 (4x9x9)", dwp_bwd_kernel<bf16_t, 1, false, true>, tb + 4 * 81 * 128);
  probe("dwp_fwd<1,5> 14x14", dwp_fwd_kernel<bf16_t, 1, 5>, (64 + 288 + 256) * 4 + 256 * 128);
  probe("dwr_bwd<1,false,true,true> ring 48K", dwr_bwd_kernel<bf16_t, 1, false, true, true>, 2048 + 49152);
  probe("dwr_bwd<2,false,true,true> ring 48K", dwr_bwd_kernel<bf16_t, 2, false, true, true>, 2048 + 49152);
  probe("dwr_fwd<1> ring 16K", dwr_fwd_kernel<bf16_t, 1>, 2048 + 16384);
  probe("dwr_fwd<2> ring 64K", dwr_fwd_kernel<bf16_t, 2>, 2048 + 49152);
  return 0;
}
