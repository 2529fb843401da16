/* TUNING BUILD ONLY -- not part of the product ABI.
 *
 * libspb_hip.so (and its twins libspb_hip_f16.so / libspb_hip_det.so) hold NO mutable tuning state: kernel selection thresholds, grid
 * sizes and A/B switches are constants there, and none of the functions below is exported.  They exist in libspb_hip_tune.so only --
 * the same sources compiled with -DSPB_TUNING (speedplusbaseline_amd/build.py) -- which the measurement scripts (bench.py with
 * SPB_DEBUG=..., scratch/) and the kernel-variant tests (tests/test_kernels_gpu.py: a variant the dispatcher would not pick for a shape)
 * load through speedplusbaseline_amd._lib.tuning().  Every setter changes a PROCESS-GLOBAL of that library and returns 0.
 */
#ifndef SPB_HIP_TUNING_H
#define SPB_HIP_TUNING_H
#include "spb_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int spb_debug_set_conv9_band(int on); /* decoder's last 9x9 layer: band-staged kernel (1, default) or the generic 8x8-tile kernel */
int spb_debug_set_launch_events(int on); /* side-stream forks wait on the preceding GEMM launch's completion event (1) or on a recorded event (0) */
int spb_debug_set_dw_split(int hw);        /* depthwise layers on maps up to `hw` columns wide run their weight gradient on the side stream (default 112: every depthwise layer; 0: always fused) */
int spb_debug_set_domain_tail_rows(int on); /* RevGrad forward: row-parallel AvgPool2d(7) + Conv2d(1280,1,1) tail (1, default) or the walking kernel (0) */
int spb_debug_set_join_fused(int on);      /* KRN plan: residual adds formed by the next expand convolution (1, default) or by bn_apply launches (0) */
int spb_debug_set_wgrad_parts(int on);     /* KRN plan: weight gradients as partial sums + spb_partial_reduce (1, default) or f32 atomics (0) */
int spb_debug_set_wgrad_min_flush(int n); /* fork at a depthwise backward kernel only when >= n weight gradients are queued */
int spb_debug_set_wgrad_tile(int mode, int wide_target); /* pointwise weight gradient tiles: 0 = 64 x 64 only (default), 1 = by the count of operand re-derivations, 2..5 = force 64x64 / 64x128 / 128x64 / 128x128; wide_target > 0: workgroup target of launches with 128-wide tiles (default 384) */
int spb_debug_set_wgrad_target(int wgs); /* pointwise weight gradient: row splits chosen for about this many workgroups per launch (every split adds N*K f32 atomics); wgs < 0: the same for the partial-sum form (default 512; every split adds an N*K slab) */
int spb_debug_set_wgrad_batch(int n); /* pointwise weight-gradient GEMMs handed to the side stream per fork event */
int spb_debug_set_gemm_bk64_dgrad_min_k(int k); /* backward-type small-M GEMMs with K >= k: 64x32 tiles with 64-deep chunks */
int spb_debug_set_replica_rows(long long rows); /* BatchNorm batch sums get 8 atomic replicas for tensors with at least this many rows (contexts created afterwards) */
int spb_debug_set_dw_xcd(int on); /* depthwise row kernels: channel quads of one task range on one XCD (1, default) or quad-major ids */
int spb_debug_set_stem_grid(int fwd, int wgrad); /* workgroup caps of the stem forward / weight-gradient launches (A/B) */
int spb_debug_set_gemm_plain_dma(int on); /* pro_mode 0 bf16 GEMMs: LDS-DMA ring kernel (1, default) or the register-prefetch kernel */
int spb_debug_set_optim(int vec, int per_thread, int nontemporal); /* optimizer launch shape A/B: lanes of 1|4 floats, 1|2|4 per thread, nt accesses */
int spb_debug_set_gemm_dma(int on); /* 1: small-M bf16 pointwise GEMMs use the LDS-DMA ring kernel (default 0) */
int spb_debug_set_dw_mode(int mode); /* depthwise fwd/dgrad: 1 plane kernels on maps up to spb_debug_set_dw_plane_max_w columns wide (14x14 and 7x7 by default), row-unit kernels elsewhere (default); 0 row-unit kernels only */
int spb_debug_set_fused_pw_bwd(int on); /* 0: the KRN plan never uses spb_pwconv_bwd_fused */
int spb_debug_set_stem_mfma(int on); /* 0: bf16 stem uses the scalar kernels instead of the MFMA implicit GEMM */
int spb_debug_set_dw_wgrad_blocks(int n); /* workgroups of the depthwise weight-gradient-only launches (A/B) */
int spb_debug_set_side_priority(int on); /* 1: contexts created afterwards put their weight-gradient side stream at the lowest stream priority (A/B) */
int spb_debug_set_stem_tile(int on); /* 0: the bf16 stem kernels gather their taps from global memory instead of an LDS tile (A/B); 1 (default): LDS tiles, forward bands of 8 output rows; 2: bands of 4 */
int spb_debug_set_stem_wgrad_tile(int rows); /* output rows per workgroup of the LDS-tile stem weight gradient (8 | 16; 0: gather kernel) */
int spb_debug_set_softce_split(int min_classes); /* spb_softce / _scaled: rows of at least this many classes (default 2048) use the class-split pair of launches (8 workgroups per row) */
int spb_debug_set_dw_tile(int min_width, int workgroups); /* bf16 depthwise forward on maps at least min_width wide: LDS-tile kernels (dwconv_tile.hip; default 28, 0 = never); workgroups per launch in the low 16 bits (0 = resident estimate); bit 16: the stride-1 input gradient too (off: measured slower) */
int spb_debug_set_pwb(int chunk_rows, int recompute_z, int max_waves); /* fused pointwise backward: rows per chunk (16 default | 32), z of the expand layers recomputed on the matrix cores instead of read (1 default; -1 keeps), waves per workgroup (8) */
int spb_debug_set_gemm_rs(int on, int min_m); /* bf16 GEMMs with K <= 96, N = 192 | 384 | 576, M >= min_m (4096): one-round-trip row-slab kernel (on=1, default) */
int spb_debug_set_gemm_big(int on, int min_n, int min_k); /* small-M bf16 GEMMs with N >= min_n (512), K >= min_k (256): 128 x 128 tile kernel (on=1, default) */
int spb_debug_set_bn_bwd_prep_rows(int on); /* spb_bn_bwd_prep: row-parallel kernel (1, default) or the walking kernel (0) */
int spb_debug_set_gemm_wg_cap(int n); /* tiled pointwise GEMM: most workgroups per launch (default 1024 = what is resident at once); beyond it workgroups loop over M tiles */
int spb_debug_set_gemm_wide_min_n(int n); /* small-M forward-type bf16 GEMMs with N >= n and a long reduction: 64 x 128 tiles */
int spb_debug_set_gemm_bk64_min_k(int k); /* small-M bf16 GEMMs with K >= k use 64-wide reduction chunks (default 256) */
int spb_debug_set_gconv_slab(int mode); /* wide decoder convs: 0 per-wave weight streaming; 1 slab kernel with 4 tiles (1 workgroup per CU); 2 (default) 2 tiles, 2 per CU */
int spb_debug_set_side_wgrad(int on); /* 0: pointwise weight gradients stay on the launch stream */
int spb_debug_set_dw_rows(int rows); /* rows per row unit (0: automatic) */
int spb_debug_set_dw_plane_max_w(int w); /* depthwise plane kernels: widest feature map they take (default 14; at most 28) */
int spb_debug_set_dw_plane_min_wgs(int n); /* depthwise plane kernels: fewer images per workgroup while the launch has fewer workgroups than n (default 384) */
int spb_debug_set_im2col_rgb_band(int on); /* SPN conv1 column matrix: 1 = band kernel (image rows through LDS), 0 = per-element gather */
int spb_debug_set_conv9_wgs(int n); /* 9x9 32->3 conv: persistent workgroups (default 512) */
int spb_debug_set_gconv_up2_wreg(int on); /* 64 -> 32 phase conv: a wave keeps its phase weights in registers for all its tile groups (1, default) */
int spb_debug_set_gconv_up2_prefetch(int on); /* phase (upsampling) convs: prefetch the next tile group's halo into registers (1, default) */
int spb_debug_set_gconv_wide_wgs(int n); /* residual-block convs (ghiasi_wide.hip): persistent workgroups (default 512 = two per CU) */
int spb_debug_set_gconv_wide_rotate(int on); /* ... workgroups enter the 36-step weight cycle at staggered steps (1, default; no measured effect) */
int spb_debug_set_gconv_wide_delay(int n); /* ... experiment: second workgroup of a CU starts n x 0.43 us late (0, default) */
int spb_debug_set_gconv_slab_pf(int n); /* wide decoder convs: weight slabs in flight per workgroup (3 | 6, default 6) */
int spb_debug_set_gconv_halo_prefetch(int on); /* decoder convs with LDS-resident weights: prefetch the next tile's halo (1, default) */
int spb_debug_set_gconv_wlds_pxg(int n); /* decoder convs with LDS-resident weights: 8x8 tiles per workgroup side by side (1 | 2) */
int spb_debug_set_gemm_os(int on, int min_k, int max_n, int min_m); /* small-map bf16 GEMMs with min_k <= K <= 576, N <= max_n (<= 96), M >= min_m: one-shot kernel (on=1, default; 0 arguments keep the defaults 160 / 96 / 4096) */
int spb_debug_set_gemm_sk(int on, int min_k, int rf); /* small-M bf16 GEMMs with K >= min_k (default 192): split-K-over-waves kernel (on=1, default); rf > 0 forces 16*rf-row tiles */
int spb_debug_set_gemm_st(int on, int min_m, int wgs); /* forward 1x1 GEMMs with M >= min_m (100000: the 112x112 / 56x56 maps), K <= 160, N in {16, 24, 32, 48k}: streaming kernel (gemm_st.hip; on=1, default); wgs: persistent workgroups (1280) */
int spb_debug_set_fuse_expand(int min_width); /* KRN plan: expand -> depthwise fusion with recompute for inverted-residual blocks whose map is at least min_width wide and whose expand input has <= 32 channels (56, default: blocks 2-4; 28: blocks 2-7 need the pointwise backward's recompute too and are not wired; 0: off) */
int spb_debug_set_router_side(int on); /* KRN plan: the RouterV2 branch (forward: 1x1 + reorg; backward: un-reorg + input gradient) on the side stream beside the 7x7 chain (1, default) or inside the chain (0) */
#ifdef __cplusplus
}
#endif
#endif
