"""ctypes binding of libotvm_hip.so (C ABI: include/otvm_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  The
product path never routes through PyTorch ops or the CPU oracle for its device compute.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTVM_HIP_LIB") or os.path.join(_HERE, "libotvm_hip.so")   # env: kernel-variant A/B runs

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
PREC_F32, PREC_F16X3, PREC_F16 = 0, 1, 2
ABI_VERSION = 19         # include/otvm_hip.h OTVM_ABI_VERSION


class ConvParams(C.Structure):
    _fields_ = [("inp", vp), ("H", i32), ("W", i32), ("Cin", i32), ("in_ld", i32),
                ("w", vp), ("K_pad", i32),
                ("bias", vp),
                ("residual", vp), ("res_ld", i32),
                ("out", vp), ("Ho", i32), ("Wo", i32), ("Cout", i32), ("out_ld", i32),
                ("kh", i32), ("kw", i32), ("stride", i32), ("pad", i32), ("dil", i32),
                ("in_relu", i32), ("act", i32),
                ("precision", i32), ("w_hi", vp), ("w_lo", vp), ("w_scale", vp), ("w_frag", vp),
                ("gn_stats", vp),
                ("in_scale", vp), ("in_shift", vp), ("in_act", i32),
                ("tune", i32),
                ("splitk_ws", vp), ("splitk_ws_bytes", i64),
                ("batch", i32), ("in_bs", i64), ("out_bs", i64), ("res_bs", i64), ("gn_bs", i32), ("norm_bs", i32),
                ("w_wfrag", vp),
                ("gn_gamma", vp), ("gn_beta", vp), ("gn_scale_out", vp), ("gn_shift_out", vp), ("gn_counter", vp), ("gn_tab_bs", i32),
                ("ws_bs", i32), ("res_scale", vp), ("res_scale_bs", i32),
                ("in_res", vp), ("in_res_ld", i32), ("in_res_bs", i64)]


class StmBottleneckParams(C.Structure):
    _fields_ = [("x", vp), ("H", i32), ("W", i32), ("Cin", i32), ("x_ld", i32), ("y", vp), ("y_ld", i32),
                ("w1f", vp), ("w2f", vp), ("w3f", vp), ("s1", vp), ("s2", vp), ("s3", vp), ("b1", vp), ("b2", vp), ("b3", vp),
                ("batch", i32), ("x_bs", i64), ("y_bs", i64), ("tile", i32)]


class GnApplyParams(C.Structure):
    _fields_ = [("x", vp), ("P", i64), ("C", i32), ("ld", i32), ("stats", vp), ("gamma", vp), ("beta", vp),
                ("residual", vp), ("res_ld", i32), ("res_scale", vp), ("res_shift", vp), ("res_act", i32), ("act", i32),
                ("out", vp), ("out_ld", i32),
                ("batch", i32), ("x_bs", i64), ("res_bs", i64), ("out_bs", i64), ("stats_bs", i32), ("norm_bs", i32)]


class HeadParams(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("n_out", i32), ("img", vp), ("img_ld", i32), ("P", i64), ("alpha_out", vp),
                ("alpha_stride", i32), ("tri_out", vp), ("sm", vp), ("sm_ld", i32),
                ("img_bs", i64), ("alpha_bs", i64), ("tri_bs", i64), ("sm_bs", i64), ("w16", vp)]


class GramParams(C.Structure):
    _fields_ = [("x", vp), ("P", i64), ("C", i32), ("ld", i32), ("in_scale", vp), ("in_shift", vp), ("in_act", i32),
                ("gpart", vp), ("spart", vp), ("passes", i32), ("batch", i32), ("x_bs", i64), ("norm_bs", i32), ("diag", vp)]


class GnPredictParams(C.Structure):
    _fields_ = [("gpart", vp), ("spart", vp), ("P", i64), ("C", i32), ("Cout", i32), ("Mp", vp), ("v", vp), ("ws", vp),
                ("counter", vp), ("wscale", vp), ("gamma", vp), ("beta", vp), ("res_shift", vp),
                ("scale_eff", vp), ("bias_eff", vp), ("stat_out", vp),
                ("batch", i32), ("tab_bs", i32), ("rs_bs", i32), ("diag", vp)]


class PreprocessParams(C.Structure):
    _fields_ = [("fg", vp), ("bg", vp), ("a", vp),
                ("H", i32), ("W", i32), ("Hp", i32), ("Wp", i32), ("lh", i32), ("lw", i32),
                ("mean", f32 * 3), ("std", f32 * 3), ("mean_q", f32 * 3), ("std_q", f32 * 3),
                ("mean_m", f32 * 3), ("std_m", f32 * 3),
                ("scaled_imgs", vp),
                ("x11", vp), ("x11_ld", i32), ("sq", vp), ("sq_ld", i32), ("sm", vp), ("sm_ld", i32),
                ("d80", vp), ("d80_ld", i32),
                ("fg_u8", vp), ("bg_u8", vp), ("u8_rgb", i32)]


class PpmHeadParams(C.Structure):
    _fields_ = [("pooled", vp), ("C", i32), ("K_pad", i32), ("Cout", i32),
                ("w", vp * 4), ("bias", vp * 4), ("gamma", vp * 4), ("beta", vp * 4), ("out", vp * 4),
                ("out_ld", i32), ("act", i32)]


_PROTOS = {
    "otvm_abi_version": (i32, []),
    "otvm_patch_weight_bytes_f16x3": (i64, [i32, i32]),
    "otvm_pack_patch_weight_f16x3": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "otvm_stem_weight_bytes_f16x3": (i64, [i32]),
    "otvm_wave_weight_bytes_f16x3": (i64, [i32, i32]),
    "otvm_pack_wave_weight_f16x3": (i32, [vp, vp, i32, i32, vp, vp]),
    "otvm_pack_stem_weight_f16x3": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "otvm_fold_bn": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp]),
    "otvm_pack_conv_weight": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp]),
    "otvm_conv2d": (i32, [C.POINTER(ConvParams), vp]),
    "otvm_split_conv_weight_f16x3": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "otvm_gn_stats": (i32, [vp, i64, i32, i32, vp, vp]),
    "otvm_gn_table": (i32, [vp, i64, i32, vp, vp, vp, vp, vp]),
    "otvm_conv2d_accepts_input_norm": (i32, [C.POINTER(ConvParams)]),
    "otvm_conv2d_input_norm_kind": (i32, [C.POINTER(ConvParams)]),
    "otvm_conv2d_accepts_input_residual": (i32, [C.POINTER(ConvParams)]),
    "otvm_conv2d_head": (i32, [C.POINTER(ConvParams), C.POINTER(HeadParams), vp]),
    "otvm_head16_weight_bytes_f16x3": (i64, []),
    "otvm_pack_head16_weight_f16x3": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "otvm_conv2d_candidates": (i32, [C.POINTER(ConvParams), C.POINTER(i32), i32]),
    "otvm_gn_apply": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, i32, vp]),
    "otvm_maxpool3x3s2": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
    "otvm_upsample_bilinear": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
    "otvm_ppm_pool_ws_bytes": (i64, [i32, i32]),
    "otvm_ppm_pool": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "otvm_ppm_head": (i32, [C.POINTER(PpmHeadParams), vp]),
    "otvm_stm_bottleneck_f16x3": (i32, [C.POINTER(StmBottleneckParams), vp]),
    "otvm_ppm_conv_z": (i32, [C.POINTER(vp), i32, vp, vp, vp]),
    "otvm_ppm_conv_add": (i32, [vp, i32, i32, vp, i32, vp, vp]),
    "otvm_memory_read_ws_bytes": (i64, [i32, i32]),
    "otvm_memory_read": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp), i32, i32, vp, i32, vp, vp]),
    "otvm_bank_slot_bytes_f16x3": (i64, [i32]),
    "otvm_bank_pack_f16x3": (i32, [vp, vp, i32, vp, vp]),
    "otvm_memory_read_f16x3": (i32, [vp, i32, C.POINTER(vp), i32, i32, vp, i32, vp, vp]),
    "otvm_memory_read_f16x3_partial_count": (i32, [i32, i32]),
    "otvm_memory_read_f16x3_partial": (i32, [vp, i32, C.POINTER(vp), i32, i32, vp, i32, i32, C.POINTER(i32), vp]),
    "otvm_memory_read_f16x3_combine": (i32, [vp, i32, i32, i32, vp, i32, vp]),
    "otvm_preprocess": (i32, [C.POINTER(PreprocessParams), vp]),
    "otvm_pad_trimap": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, vp]),
    "otvm_upsample4_softmax3": (i32, [vp, i32, i32, i32, vp, vp]),
    "otvm_trimap_encode_ws_bytes": (i64, [i32, i32]),
    "otvm_trimap_encode": (i32, [vp, i32, i32, vp, vp, vp, i32, vp, i32, vp, vp]),
    "otvm_fba_head": (i32, [vp, i32, vp, vp, i32, vp, i32, i64, vp, i32, vp, vp, i32, vp]),
    "otvm_crop_outputs": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "otvm_trimap_from_alpha": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "otvm_onehot_argmax3": (i32, [vp, i64, vp, vp]),
    "otvm_matting_metrics": (i32, [vp, vp, vp, vp, vp, vp, i64, vp, vp]),
    "otvm_gn_stats_b": (i32, [vp, i64, i32, i32, vp, i32, i64, i32, vp]),
    "otvm_gn_table_b": (i32, [vp, i64, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "otvm_gn_apply_b": (i32, [C.POINTER(GnApplyParams), vp]),
    "otvm_maxpool3x3s2_b": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i64, i64, vp]),
    "otvm_upsample_bilinear_b": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i64, i64, i64, i32, vp]),
    "otvm_fba_head_train": (i32, [vp, i32, vp, vp, i32, vp, i32, i64, vp, vp, vp]),
    "otvm_upsample4_logits3": (i32, [vp, i32, i32, i32, vp, vp]),
    "otvm_trimap_to_sm": (i32, [vp, i64, vp, i32, vp]),
    "otvm_scale_flip3": (i32, [vp, i64, i64, f32, vp, vp]),
    "otvm_trimask": (i32, [vp, i64, i64, vp, vp, vp, vp, vp]),
    "otvm_loss_fba_comp": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]),
    "otvm_loss_grad_l1": (i32, [vp, vp, i64, i32, i32, f32, vp, vp]),
    "otvm_loss_exclusion_level": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "otvm_avgpool2": (i32, [vp, i64, i32, i32, vp, vp]),
    "otvm_loss_lap_level": (i32, [vp, vp, i64, i32, i32, C.c_double, vp, vp, vp, vp]),
    "otvm_loss_temporal": (i32, [vp, vp, i32, i32, i64, vp, vp]),
    "otvm_loss_ce3": (i32, [vp, vp, i64, i64, vp, vp]),
    "otvm_finite_guard": (i32, [vp, i64, i32, i32, f32, i32, vp, vp]),
    "otvm_clear": (i32, [vp, i64, vp]),
    "otvm_gram_block": (i32, [i32]),
    "otvm_gram_entries": (i64, [i32]),
    "otvm_gram_chunks": (i32, [i64, i32, C.POINTER(i32)]),
    "otvm_gram_f16": (i32, [C.POINTER(GramParams), vp]),
    "otvm_gn_predict": (i32, [C.POINTER(GnPredictParams), vp]),
    "otvm_gn_predict_ws_bytes": (i64, []),
}

EXPORTED = sorted(list(_PROTOS) + ["otvm_last_error"])

_lib = None


def load():
    """Load the HIP library (after torch, so both share one libamdhip64 runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "otvm_amd: %s is missing -- build it with `python otvm_amd/csrc/build.py` "
            "(or __graft_entry__.build()).  There is no CPU/PyTorch fallback." % LIB_PATH)
    import torch  # noqa: F401  (loads torch's bundled libamdhip64.so.7 first; same SONAME)
    lib = C.CDLL(LIB_PATH)
    lib.otvm_last_error.restype = C.c_char_p
    lib.otvm_last_error.argtypes = []
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = lib.otvm_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError("otvm_amd: %s has ABI version %d, this binding expects %d -- rebuild it with "
                           "`python otvm_amd/csrc/build.py`" % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("otvm_hip %s failed (%d): %s" % (what, rc, load().otvm_last_error().decode()))
