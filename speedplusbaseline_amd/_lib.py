"""ctypes binding of libspb_hip.so (the C-ABI declared in include/spb_hip.h).

There is deliberately no fallback: if the HIP library is missing or a symbol is absent, importing/using the ops raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspb_hip.so")
if os.environ.get("SPB_LIB_VARIANT"):   # kernel experiments only (scratch/build_variant.py): libspb_hip.<variant>.so, same C-ABI
    LIB_PATH = os.path.join(_HERE, "libspb_hip.%s.so" % os.environ["SPB_LIB_VARIANT"])

F32, BF16 = 0, 1
AMP_STATE, AMP_SCALE, AMP_INV_SCALE, AMP_TRACKER, AMP_FOUND_INF, AMP_STEPS, AMP_LR, AMP_SKIP = 12, 0, 1, 2, 3, 4, 5, 8   # spb_hip.h SPB_AMP_*
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY = 0, 1, 2, 3

vp = C.c_void_p
fp = C.c_void_p  # float* passed as raw device address
i32 = C.c_int
i64 = C.c_longlong
f32 = C.c_float


class BNRef(C.Structure):
    _fields_ = [("sums", vp), ("gamma", vp), ("beta", vp), ("bsums", vp), ("inv_n", f32), ("eps", f32),
                ("slope", f32), ("C", i32), ("R", i32), ("act", i32), ("moments", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("A2", vp), ("Bw", vp), ("Y", vp), ("res", vp), ("Zout", vp), ("bias", vp), ("osums", vp),
                ("pro", BNRef), ("epi", BNRef), ("M", i32), ("K", i32), ("N", i32), ("pro_mode", i32),
                ("epi_mode", i32), ("out_act", i32), ("oR", i32), ("out_scale", f32), ("lda", i32), ("ldc", i32), ("stop_event", vp),
                ("pro2", BNRef), ("Ymat", vp)]


class RedJob(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("stride", i64), ("n", i32), ("nparts", i32)]


class WgradArgs(C.Structure):
    _fields_ = [("G", vp), ("Zn", vp), ("X", vp), ("dW", vp), ("pro_dz", BNRef), ("pro_a", BNRef), ("M", i32),
                ("K", i32), ("N", i32), ("ldg", i32), ("ldx", i32), ("part", vp), ("part_cap", i64), ("job_out", vp)]


class PwBwdArgs(C.Structure):
    _fields_ = [("G", vp), ("Zn", vp), ("Wt", vp), ("X", vp), ("Zout", vp), ("res", vp), ("Y", vp), ("dW", vp),
                ("osums", vp), ("pro_dz", BNRef), ("pro_a", BNRef), ("epi", BNRef), ("M", i32), ("K", i32), ("N", i32),
                ("oR", i32), ("part", vp), ("part_cap", i64), ("job_out", vp)]


AMP_SEGS = 8


class AmpSegs(C.Structure):
    _fields_ = [("ptr", vp * AMP_SEGS), ("n", i64 * AMP_SEGS), ("is16", i32 * AMP_SEGS), ("nseg", i32)]


class GconvArgs(C.Structure):
    _fields_ = [("X", vp), ("W", vp), ("bias", vp), ("coef", vp), ("Y", vp), ("stats", vp), ("B", i32), ("Hin", i32),
                ("Win", i32), ("Cin", i32), ("Cout", i32), ("KH", i32), ("stride", i32), ("upsample", i32), ("relu", i32),
                ("ldc", i32), ("in_stats", vp), ("in_gamma", vp), ("in_beta", vp), ("in_ld", i32), ("in_inv_n", C.c_float),
                ("in_eps", C.c_float)]


class DwArgs(C.Structure):
    _fields_ = [("X", vp), ("X2", vp), ("Xin", vp), ("Wd", vp), ("Y", vp), ("dW", vp), ("res", vp), ("Zout", vp),
                ("osums", vp), ("pro", BNRef), ("pro_in", BNRef), ("epi", BNRef), ("B", i32), ("H", i32), ("W", i32),
                ("C", i32), ("stride", i32), ("epi_mode", i32), ("oR", i32), ("part", vp), ("part_cap", i64), ("job_out", vp),
                ("entry_flag", vp), ("entry_val", C.c_uint32), ("Xe", vp), ("We", vp), ("xe", BNRef), ("Ce", i32)]


class BnApplyArgs(C.Structure):
    _fields_ = [("Z", vp), ("res", vp), ("Y", vp), ("bn", BNRef), ("bn_res", BNRef), ("B", i32), ("H", i32),
                ("W", i32), ("C", i32), ("ldc", i32), ("coff", i32), ("reorg", i32)]


class BnBwdArgs(C.Structure):
    _fields_ = [("dY", vp), ("Z", vp), ("G", vp), ("osums", vp), ("bn", BNRef), ("B", i32), ("H", i32), ("W", i32),
                ("C", i32), ("ldc", i32), ("coff", i32), ("reorg", i32), ("oR", i32)]


class HeadArgs(C.Structure):
    _fields_ = [("Z", vp), ("Wp", vp), ("bias", vp), ("target", vp), ("partial", vp), ("pred", vp), ("dout", vp),
                ("scalars", vp), ("pro", BNRef), ("B", i32), ("J", i32), ("Jp", i32), ("HW", i32), ("C", i32),
                ("S", i32)]


class HeadBwdArgs(C.Structure):
    _fields_ = [("Z", vp), ("Wp", vp), ("dout", vp), ("G", vp), ("osums", vp), ("dW", vp), ("dbias", vp),
                ("pro", BNRef), ("gscale", f32), ("B", i32), ("J", i32), ("Jp", i32), ("HW", i32), ("C", i32),
                ("oR", i32), ("roles", i32), ("gscale_dev", vp)]


class BnUpdEntry(C.Structure):
    _fields_ = [("sums_off", i64), ("bsums_off", i64), ("rm_off", i64), ("gamma_off", i64), ("beta_off", i64),
                ("C", i32), ("R", i32), ("bn_index", i32), ("inv_n", f32), ("unbias", f32)]


class PrepEntry(C.Structure):
    _fields_ = [("src_off", i64), ("dst_off", i64), ("rows", i32), ("cols", i32), ("mode", i32), ("aux", i32),
                ("aux2", i32), ("tile0", i32)]


class OptimArgs(C.Structure):
    _fields_ = [("params", vp), ("grads", vp), ("m", vp), ("v", vp), ("sqnorm", vp), ("gmul", vp), ("hyper", vp),
                ("n", i64), ("kind", i32), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32),
                ("weight_decay", f32), ("max_norm", f32), ("clip_value", f32), ("bias_c1", f32), ("bias_c2", f32),
                ("first_step", i32), ("shadow_bf16", vp), ("max_blocks", i32), ("skip", vp), ("sq_partials", vp), ("n_sq_partials", i32)]


class SpnConvArgs(C.Structure):
    _fields_ = [("X", vp), ("Wp", vp), ("bias", vp), ("mask", vp), ("Y", vp)] + [(k, i32) for k in
                ("B", "H", "W", "Cx", "KH", "KW", "stride", "pad", "groups", "Cg", "Ng", "Kp", "relu")]


class SpnPackJob(C.Structure):
    _fields_ = [("W", vp), ("out", vp), ("outT", vp)] + [(k, i32) for k in ("Cout", "Cin", "groups", "KH", "KW", "Kp", "mode", "chw")]


class PreprocArgs(C.Structure):
    _fields_ = [("src", vp), ("table", vp), ("ftable", vp), ("noise", vp), ("out", vp), ("bounds", vp), ("coeffs", vp), ("tmp", vp),
                ("B", i32), ("S", i32), ("C", i32), ("max_h", i32), ("flags_any", i32), ("noise_std", f32)]


class FcEpiArgs(C.Structure):
    _fields_ = [("accT", vp), ("src", vp), ("bias", vp), ("H", vp), ("Y", vp), ("YT", vp), ("mask", vp), ("db", vp),
                ("M", i32), ("F", i32), ("mode", i32), ("relu", i32), ("p", f32), ("scale", f32), ("seed", C.c_ulonglong),
                ("mask_given", i32)]


class ActInfo(C.Structure):
    _fields_ = [("z_off", i64), ("g_off", i64), ("H", i32), ("W", i32), ("C", i32), ("bn_index", i32)]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", i64), ("numel", i64), ("ndim", i32), ("shape", i32 * 4)]


# every symbol include/spb_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "spb_pwconv_gemm": (i32, [i32, C.POINTER(GemmArgs), vp]),
    "spb_pwconv_wgrad": (i32, [i32, C.POINTER(WgradArgs), vp]),
    "spb_partial_reduce": (i32, [C.POINTER(RedJob), i32, vp]),
    "spb_pwconv_bwd_fused": (i32, [i32, C.POINTER(PwBwdArgs), vp]),
    "spb_dwconv_fwd": (i32, [i32, C.POINTER(DwArgs), vp]),
    "spb_dwconv_dgrad": (i32, [i32, C.POINTER(DwArgs), vp]),
    "spb_dwconv_wgrad": (i32, [i32, C.POINTER(DwArgs), vp]),
    "spb_stem_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spb_stem_wgrad": (i32, [i32, vp, vp, vp, C.POINTER(BNRef), vp, i32, i32, i32, vp]),
    "spb_bn_apply": (i32, [i32, C.POINTER(BnApplyArgs), vp]),
    "spb_bn_bwd_prep": (i32, [i32, C.POINTER(BnBwdArgs), vp]),
    "spb_head_fwd": (i32, [i32, C.POINTER(HeadArgs), vp]),
    "spb_head_bwd": (i32, [i32, C.POINTER(HeadBwdArgs), vp]),
    "spb_bn_running_update": (i32, [vp, i32, vp, vp, vp, f32, vp]),
    "spb_bn_param_grads": (i32, [vp, i32, vp, vp, vp]),
    "spb_bn_param_grads_zero": (i32, [vp, i32, vp, vp, vp]),
    "spb_bn_load_running": (i32, [vp, i32, vp, vp, vp]),
    "spb_weight_prep": (i32, [i32, vp, i32, i32, vp, vp, vp]),
    "spb_grad_sqnorm": (i32, [vp, i64, vp, vp]),
    "spb_grad_sqnorm_partials": (i32, [vp, i64, vp, vp]),
    "spb_arena_zero": (i32, [vp, i64, vp]),
    "spb_arena_add": (i32, [vp, vp, i64, vp]),
    "spb_stream_create": (i32, [i32, C.POINTER(C.c_void_p)]),
    "spb_stream_destroy": (i32, [vp]),
    "spb_fork_create": (i32, [C.POINTER(C.c_void_p)]),
    "spb_fork_destroy": (None, [vp]),
    "spb_fork_streams": (i32, [vp, vp, vp]),
    "spb_optim_step": (i32, [C.POINTER(OptimArgs), vp]),
    "spb_fc_wgrad_update": (i32, [vp, vp, i32, i32, i32, C.POINTER(OptimArgs), vp]),
    "spb_krn_create": (i32, [i32, i32, C.POINTER(vp)]),
    "spb_krn_destroy": (None, [vp]),
    "spb_krn_num_params": (i32, [vp]),
    "spb_krn_param_info": (i32, [vp, i32, C.POINTER(TensorInfo)]),
    "spb_krn_num_buffers": (i32, [vp]),
    "spb_krn_buffer_info": (i32, [vp, i32, C.POINTER(TensorInfo)]),
    "spb_krn_num_bn": (i32, [vp]),
    "spb_krn_bn_name": (i32, [vp, i32, C.c_char_p]),
    "spb_krn_param_numel": (i64, [vp]),
    "spb_krn_buffer_numel": (i64, [vp]),
    "spb_krn_wcompute_bytes": (i64, [vp, i32]),
    "spb_krn_tables_bytes": (i64, [vp]),
    "spb_krn_bind": (i32, [vp, vp, vp, vp, vp, vp, vp, i32]),
    "spb_krn_ctx_bytes": (i64, [vp, i32, i32]),
    "spb_krn_ctx_create": (i32, [vp, i32, vp, C.POINTER(vp)]),
    "spb_krn_ctx_destroy": (None, [vp]),
    "spb_krn_ctx_set_side_stream": (i32, [vp, i32]),
    "spb_krn_ctx_set_loss_scale": (i32, [vp, vp]),
    "spb_krn_prepare_weights": (i32, [vp, vp]),
    "spb_krn_forward": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
    "spb_krn_update_running": (i32, [vp, vp]),
    "spb_krn_bucket_split": (i64, [vp]),
    "spb_krn_ctx_wait_bucket": (i32, [vp, vp]),
    "spb_krn_ctx_set_bucket": (i32, [vp, i32]),
    "spb_krn_backward": (i32, [vp, vp, f32, i32, vp, f32, vp]),
    "spb_bce_logits": (i32, [vp, f32, i32, vp, vp, f32, vp]),
    "spb_krn_prof_enable": (i32, [vp, i32]),
    "spb_krn_prof_num_categories": (i32, []),
    "spb_krn_prof_category_name": (C.c_char_p, [i32]),
    "spb_krn_prof_read": (i32, [vp, vp, vp, vp, vp]),
    "spb_krn_prof_launches": (i32, [vp, i32, vp, vp, vp]),
    "spb_krn_weight_prep_bytes": (i64, [vp]),
    "spb_gconv": (i32, [i32, C.POINTER(GconvArgs), vp]),
    "spb_gconv_up2": (i32, [i32, C.POINTER(GconvArgs), vp]),
    "spb_gconv_wide": (i32, [i32, C.POINTER(GconvArgs), vp]),
    "spb_gconv_wide_pack": (i32, [vp, vp, vp]),
    "spb_conv9_rgb": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "spb_in_coef": (i32, [vp, vp, vp, i32, vp, i32, i32, i64, f32, vp]),
    "spb_style_fc": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "spb_in_apply": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "spb_in_apply_f32": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "spb_final_sigmoid_f32": (i32, [vp, vp, vp, i32, i64, i32, vp]),
    "spb_final_sigmoid": (i32, [vp, vp, vp, i32, i64, i32, vp]),
    "spb_in_apply_stats": (i32, [vp, vp, vp, vp, i32, f32, vp, vp, i32, i64, i32, i32, vp]),
    "spb_final_sigmoid_stats": (i32, [vp, vp, vp, vp, i32, f32, vp, i32, i64, i32, vp]),
    "spb_im2col": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_im2col_rgb": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_col2im": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_maxpool3s2_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spb_maxpool3s2_bwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spb_maxpool3s2_relu_bwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spb_lrn2_fwd": (i32, [i32, vp, vp, i64, i32, f32, f32, f32, vp]),
    "spb_lrn2_bwd": (i32, [i32, vp, vp, vp, i64, i32, f32, f32, f32, vp]),
    "spb_relu_bwd": (i32, [i32, vp, vp, vp, vp, i64, f32, vp]),
    "spb_dropout": (i32, [i32, vp, vp, i64, f32, C.c_ulonglong, i32, vp]),
    "spb_softce": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "spb_softce_scaled": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp]),
    "spb_amp_check": (i32, [vp, i64, vp, vp]),
    "spb_amp_check16": (i32, [vp, i64, vp, vp]),
    "spb_amp_decide": (i32, [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]),
    "spb_amp_step": (i32, [vp, f32, f32, f32, f32, f32, i32, vp]),
    "spb_softce_rows": (i32, [i32, vp, vp, vp, i32, i32, vp]),
    "spb_colsum": (i32, [i32, vp, vp, i64, i32, vp]),
    "spb_preproc_max_taps": (i32, []),
    "spb_preproc_batch": (i32, [C.POINTER(PreprocArgs), vp]),
    "spb_spn_pack_conv": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_spn_unpack_conv_grad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_spn_conv": (i32, [C.POINTER(SpnConvArgs), vp]),
    "spb_spn_conv_wgrad": (i32, [C.POINTER(SpnConvArgs), vp, vp, vp]),
    "spb_spn_col_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "spb_spn_stem": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spb_spn_pack_jobs": (i32, [i32, C.POINTER(SpnPackJob), i32, vp]),
    "spb_spn_pack_conv_dgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "spb_fc_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "spb_fc_dgrad": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "spb_fc_wgrad": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "spb_fc_epilogue": (i32, [C.POINTER(FcEpiArgs), vp]),
    "spb_spn_flatten": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "spb_spn_unflatten_grad": (i32, [vp, vp, i32, i32, i32, vp]),
    "spb_debug_trread": (i32, [vp, vp, vp]),
    "spb_version": (C.c_char_p, []),
    "spb_krn_num_acts": (i32, [vp]),
    "spb_krn_ctx_act_info": (i32, [vp, i32, C.POINTER(ActInfo)]),
    "spb_fork_selftest": (i32, []),
    "spb_hip_runtime_version": (i32, []),
    "spb_krn_ctx_virtual": (i32, [vp, i32]),
    "spb_krn_ctx_materialize": (i32, [vp, vp]),
    "spb_det_available": (i32, []),
    "spb_det_register": (i32, [vp, i64, vp]),
    "spb_det_unregister": (i32, [vp]),
    "spb_det_unregister_if": (i32, [vp, vp]),
    "spb_det_flush": (i32, [vp, vp]),
    "spb_det_misses": (i64, []),
    "spb_krn_set_det": (i32, [vp, i32]),
    "spb_krn_ctx_set_det": (i32, [vp, i32]),
    "spb_krn_ctx_stats": (i32, [vp, C.POINTER(vp), C.POINTER(i64)]),
}

# the knobs of the TUNING build (include/spb_hip_tuning.h): exported by libspb_hip_tune.so only
TUNING_SYMBOLS = {
    "spb_debug_set_softce_split": (i32, [i32]),
    "spb_debug_set_optim": (i32, [i32, i32, i32]),
    "spb_debug_set_conv9_band": (i32, [i32]),
    "spb_debug_set_launch_events": (i32, [i32]),
    "spb_debug_set_dw_split": (i32, [i32]),
    "spb_debug_set_wgrad_parts": (i32, [i32]),
    "spb_debug_set_join_fused": (i32, [i32]),
    "spb_debug_set_domain_tail_rows": (i32, [i32]),
    "spb_debug_set_wgrad_min_flush": (i32, [i32]),
    "spb_debug_set_wgrad_batch": (i32, [i32]),
    "spb_debug_set_wgrad_target": (i32, [i32]),
    "spb_debug_set_wgrad_tile": (i32, [i32, i32]),
    "spb_debug_set_gemm_bk64_dgrad_min_k": (i32, [i32]),
    "spb_debug_set_replica_rows": (i32, [i64]),
    "spb_debug_set_dw_xcd": (i32, [i32]),
    "spb_debug_set_stem_grid": (i32, [i32, i32]),
    "spb_debug_set_gemm_plain_dma": (i32, [i32]),
    "spb_debug_set_gemm_dma": (i32, [i32]),
    "spb_debug_set_dw_mode": (i32, [i32]),
    "spb_debug_set_side_wgrad": (i32, [i32]),
    "spb_debug_set_gconv_slab": (i32, [i32]),
    "spb_debug_set_gemm_bk64_min_k": (i32, [i32]),
    "spb_debug_set_gemm_wide_min_n": (i32, [i32]),
    "spb_debug_set_gemm_big": (i32, [i32, i32, i32]),
    "spb_debug_set_pwb": (i32, [i32, i32, i32]),
    "spb_debug_set_dw_tile": (i32, [i32, i32]),
    "spb_debug_set_gemm_rs": (i32, [i32, i32]),
    "spb_debug_set_gemm_st": (i32, [i32, i32, i32]),
    "spb_debug_set_fuse_expand": (i32, [i32]),
    "spb_debug_set_router_side": (i32, [i32]),
    "spb_debug_set_gemm_wg_cap": (i32, [i32]),
    "spb_debug_set_bn_bwd_prep_rows": (i32, [i32]),
    "spb_debug_set_stem_mfma": (i32, [i32]),
    "spb_debug_set_stem_tile": (i32, [i32]),
    "spb_debug_set_side_priority": (i32, [i32]),
    "spb_debug_set_dw_wgrad_blocks": (i32, [i32]),
    "spb_debug_set_stem_wgrad_tile": (i32, [i32]),
    "spb_debug_set_fused_pw_bwd": (i32, [i32]),
    "spb_debug_set_dw_rows": (i32, [i32]),
    "spb_debug_set_dw_plane_min_wgs": (i32, [i32]),
    "spb_debug_set_dw_plane_max_w": (i32, [i32]),
    "spb_debug_set_gemm_sk": (i32, [i32, i32, i32]),
    "spb_debug_set_gemm_os": (i32, [i32, i32, i32, i32]),
    "spb_debug_set_gconv_wlds_pxg": (i32, [i32]),
    "spb_debug_set_gconv_halo_prefetch": (i32, [i32]),
    "spb_debug_set_im2col_rgb_band": (i32, [i32]),
    "spb_debug_set_gconv_slab_pf": (i32, [i32]),
    "spb_debug_set_conv9_wgs": (i32, [i32]),
    "spb_debug_set_gconv_up2_wreg": (i32, [i32]),
    "spb_debug_set_gconv_up2_prefetch": (i32, [i32]),
    "spb_debug_set_gconv_wide_wgs": (i32, [i32]),
    "spb_debug_set_gconv_wide_rotate": (i32, [i32]),
    "spb_debug_set_gconv_wide_delay": (i32, [i32]),
}

_lib = None
_lib_tune = None
_tuning_depth = 0
LIB_TUNE_PATH = os.path.join(_HERE, "libspb_hip_tune.so")


def _load(path, symbols, what):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing (%s). Build it with `python -m speedplusbaseline_amd.build`; this package "
            "has no CPU or PyTorch fallback for the hot path." % (what, path))
    # torch must load ITS HIP runtime first: the kernels work on torch-allocated device memory and torch's streams,
    # so the library has to bind to the same libamdhip64 instance (a second runtime sees no device: error 100)
    import torch  # noqa: F401
    l = C.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(l, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return l


def lib():
    """The HIP library (built in-tree by speedplusbaseline_amd/build.py).  No fallback exists.  The product library
    libspb_hip.so has no tuning knobs; inside a `with tuning():` block -- or for the whole process when SPB_DEBUG is set (measurement
    runs) -- this returns the tuning build libspb_hip_tune.so instead, which exports the spb_debug_set_* knobs as well."""
    global _lib
    if _tuning_depth > 0 or os.environ.get("SPB_DEBUG") or os.environ.get("SPB_TUNING_LIB"):
        return lib_tune()
    if _lib is None:
        _lib = _load(LIB_PATH, SYMBOLS, "libspb_hip.so")
    return _lib


def lib_tune():
    """libspb_hip_tune.so: the sources compiled with -DSPB_TUNING (include/spb_hip_tuning.h).  Same C-ABI + the spb_debug_set_* knobs."""
    global _lib_tune
    if _lib_tune is None:
        both = dict(SYMBOLS); both.update(TUNING_SYMBOLS)
        _lib_tune = _load(LIB_TUNE_PATH, both, "libspb_hip_tune.so")
        _apply_debug_env(_lib_tune)
    return _lib_tune


class tuning:
    """`with tuning() as l:` -- lib() returns the tuning build inside the block (kernel-variant tests, A/B measurements).  Objects that
    cached a library handle before the block (KrnEngine.lib) keep theirs."""

    def __enter__(self):
        global _tuning_depth
        _tuning_depth += 1
        return lib_tune()

    def __exit__(self, *exc):
        global _tuning_depth
        _tuning_depth -= 1
        return False


def tuned(fn):
    """decorator: run `fn` inside `with tuning():` (tests that select a kernel variant through the knobs)"""
    import functools

    @functools.wraps(fn)
    def inner(*a, **k):
        with tuning():
            return fn(*a, **k)
    return inner


def _apply_debug_env(l):
    """A/B switch for measurement runs: SPB_DEBUG="spb_debug_set_pwb:32,0,4;spb_debug_set_dw_split:56" calls the named
    spb_debug_* knobs (integer arguments) once, right after the tuning library is loaded.  Unset in production."""
    spec = os.environ.get("SPB_DEBUG", "")
    for item in filter(None, (x.strip() for x in spec.split(";"))):
        name, _, args = item.partition(":")
        if not name.startswith("spb_debug_") or name not in TUNING_SYMBOLS:
            raise RuntimeError("SPB_DEBUG: %r is not a knob of libspb_hip_tune.so" % name)
        vals = [int(v) for v in args.split(",") if v.strip()]
        getattr(l, name)(*vals)


_lib_f16 = None
LIB_F16_PATH = os.path.join(_HERE, "libspb_hip_f16.so")


def lib_f16():
    """The IEEE-half twin of the library (the SPN sources compiled with -DSPB_F16, csrc/common.h): same entry points and
    argument structs, 16-bit tensors are float16 instead of bfloat16 and the matrix cores run v_mfma_f32_16x16x32_f16.  It
    holds the SPN path only (spn*.hip, the pointwise GEMMs, the elementwise / optimizer kernels); no fallback exists."""
    global _lib_f16
    if _lib_f16 is None:
        if not os.path.exists(LIB_F16_PATH):
            raise RuntimeError("libspb_hip_f16.so is missing (%s). Build it with `python -m speedplusbaseline_amd.build`." % LIB_F16_PATH)
        import torch  # noqa: F401
        l = C.CDLL(LIB_F16_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name, None)          # the twin exports the SPN subset of the C-ABI
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib_f16 = l
    return _lib_f16


_lib_det = None
LIB_DET_PATH = os.path.join(_HERE, "libspb_hip_det.so")


def lib_det():
    """The reproducible twin (the KRN sources compiled with -DSPB_DET, csrc/common.h): same entry points; float atomics are exact
    fixed-point accumulations, so results do not depend on workgroup arrival order.  KrnEngine(..., deterministic=True) uses it."""
    global _lib_det
    if _lib_det is None:
        if not os.path.exists(LIB_DET_PATH):
            raise RuntimeError("libspb_hip_det.so is missing (%s). Build it with `python -m speedplusbaseline_amd.build`." % LIB_DET_PATH)
        import torch  # noqa: F401
        l = C.CDLL(LIB_DET_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name, None)          # the twin exports the KRN subset of the C-ABI
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        if l.spb_det_available() != 1:
            raise RuntimeError("libspb_hip_det.so was not built with -DSPB_DET")
        _lib_det = l
    return _lib_det


def lib_for(precision):
    return lib_f16() if precision == "fp16" else lib()


class SpbError(RuntimeError):
    pass


_CODES = {-1: "SPB_E_ARG", -2: "SPB_E_SHAPE", -3: "SPB_E_STATE", -4: "SPB_E_UNSUPPORTED",
          -5: "SPB_E_TIMEOUT: a stream-fork gate gave up after SPB_FORK_TIMEOUT_S seconds (the launch it waited for never ran); "
              "the context is poisoned -- rebuild the engine"}


def check(code, what):
    if code != 0:
        raise SpbError("%s failed with code %d%s" % (what, code, (" (%s)" % _CODES[code]) if code in _CODES else ""))
