"""ctypes binding of libammc_hip.so (the C ABI declared in include/ammc_hip.h).

The HIP library is the product: if it is missing or fails to load this module
raises - there is no CPU or ATen fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AMMC_LIB: another build of the SAME library (A/B measurements: `python -m ammcnet_aaai2021_amd.build --variant x`)
LIB_PATH = os.environ.get("AMMC_LIB") or os.path.join(HERE, "libammc_hip.so")
ABI_VERSION = 36

ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class AmmcConvDesc(C.Structure):
    """mirror of `struct AmmcConvDesc` (include/ammc_hip.h)"""
    _fields_ = [
        ("x", _p), ("w", _p), ("y", _p), ("scale", _p), ("shift", _p), ("res", _p),
        ("batch", _i32), ("height", _i32), ("width", _i32),
        ("cin", _i32), ("ntaps", _i32), ("n", _i32), ("up", _i32), ("cgroup", _i32),
        ("act", _i32), ("n_store", _i32),
        ("x_bs", _i64), ("x_rs", _i64), ("x_ps", _i64),
        ("y_bs", _i64), ("y_rs", _i64), ("y_ps", _i64),
        ("r_bs", _i64), ("r_rs", _i64), ("r_ps", _i64),
        ("y_cs", _i64), ("x_step", _i32), ("y_f32", _i32), ("s16_mf", _i32), ("outc_stream", _i32), ("overflow_flag", _p),
        ("splitk_ws", _p), ("splitk_ws_floats", _i64), ("sq_target", _p), ("sq_acc", _p),
        ("pool_y", _p), ("pool_bs", _i64), ("pool_rs", _i64), ("pool_ps", _i64), ("stats", _p),
        ("bn_c", _p), ("bn_bs", _i64), ("bn_rs", _i64), ("bn_ps", _i64),
        ("bn_mean", _p), ("bn_invstd", _p), ("bn_scale", _p), ("bn_shift", _p), ("bn_relu", _i32), ("reserved0", _i32),
    ]


class AmmcWgradDesc(C.Structure):
    """mirror of `struct AmmcWgradDesc` (include/ammc_hip.h)"""
    _fields_ = [
        ("g", _p), ("a", _p), ("dw", _p), ("zeros", _p),
        ("batch", _i32), ("height", _i32), ("width", _i32), ("n", _i32), ("cin", _i32), ("ntaps", _i32),
        ("a_step", _i32), ("reserved", _i32),
        ("g_bs", _i64), ("g_rs", _i64), ("g_ps", _i64), ("a_bs", _i64), ("a_rs", _i64), ("a_ps", _i64),
    ]


_s3 = [_i64, _i64, _i64]

# name -> (restype, argtypes); every symbol include/ammc_hip.h declares
SIGNATURES = {
    "ammc_abi_version": (C.c_int, []),
    "ammc_build_info": (C.c_char_p, []),
    "ammc_source_digests": (C.c_char_p, []),
    "ammc_error_string": (C.c_char_p, [C.c_int]),
    "ammc_set_option": (C.c_int, [C.c_char_p, _i32]),
    "ammc_conv_gemm_f32": (C.c_int, [C.POINTER(AmmcConvDesc), _p]),
    "ammc_maxpool2x2_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "ammc_nchw_to_nhwc_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _i64, _i64, _i64, _i32, _p]),
    "ammc_nhwc_to_nchw_f32": (C.c_int, [_p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_zero_halo_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p]),
    "ammc_pack_conv_weight_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_pack_convt_weight_f32": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_bn_fold_f32": (C.c_int, [_p, _p, _p, _p, _f32, _i32, _p, _p, _p]),
    "ammc_pack_codebook_f32": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "ammc_memory_topk_blocks": (C.c_int, [_i32]),
    "ammc_memory_topk_fwd_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "ammc_pack_codebook_s16": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_pack_codebook_s16_guarded": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "ammc_pack_frag_rows_s16": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_memory_block_s16": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p,
                                         _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ammc_memory_topk_fwd_s16": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "ammc_sum_partials_f32": (C.c_int, [_p, _i32, _f32, _p, _p]),
    "ammc_conv_gemm_s16": (C.c_int, [C.POINTER(AmmcConvDesc), _p]),
    "ammc_pack_up_conv_f32": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _p, _p, _p]),
    "ammc_conv_up_s16": (C.c_int, [C.POINTER(AmmcConvDesc), _p, _i64, _i64, _i64, _i32, _p, _p, _p]),
    "ammc_first_conv_image_floats": (C.c_int, []),
    "ammc_pack_first_conv_f32": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_conv_first_s16": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _i64, _i64, _p, _p]),
    "ammc_conv_first_s16_bs": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _i64, _i64, _p, _p]),
    "ammc_conv_gemm_s16_variant": (C.c_int, [C.POINTER(AmmcConvDesc), C.c_char_p, _i32]),
    "ammc_conv_gemm_s16_stats_rows": (C.c_int, [C.POINTER(AmmcConvDesc)]),
    "ammc_split_rows_f32": (C.c_int, [_p, _i64, _p, _p]),
    "ammc_split_rows_guarded_f32": (C.c_int, [_p, _i64, _p, _p, _p]),
    "ammc_absmax_bits_f32": (C.c_int, [_p, _i64, _p, _p]),
    "ammc_split_rows_scaled_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i32, _p]),
    "ammc_nchw_to_s16_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _i64, _i64, _i64, _i32, _p]),
    "ammc_s16_to_nchw_f32": (C.c_int, [_p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_maxpool2x2_s16": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "ammc_maxpool2x2_s16_idx": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i32, _i32, _i32, _i32, _p]),
    "ammc_pack_codebook_f16": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "ammc_memory_topk_f16_blocks": (C.c_int, [_i32]),
    "ammc_memory_topk_fwd_f16": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "ammc_codebook_f16_tiles_bytes": (C.c_int64, [_i32, _i32]),
    "ammc_pack_codebook_f16_tiles": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_memory_topk_f16r_blocks": (C.c_int, [_i32]),
    "ammc_memory_topk_fwd_f16r": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "ammc_conv_wgrad_f32": (C.c_int, [C.POINTER(AmmcWgradDesc), _p]),
    "ammc_conv_wgrad_s16": (C.c_int, [C.POINTER(AmmcWgradDesc), _p, _p]),
    "ammc_conv_wgrad_s16_slab_floats": (C.c_int64, [C.POINTER(AmmcWgradDesc)]),
    "ammc_conv_wgrad_s16_slabs": (C.c_int, [C.POINTER(AmmcWgradDesc), _p, _p, _i64, _p, _i32, _i32, _p]),
    "ammc_unpack_conv_wgrad_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_unpack_convt_wgrad_f32": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ammc_pack_conv_dgrad_weight_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_transpose_pad_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "ammc_pack_filters_item_bytes": (C.c_int, []),
    "ammc_pack_filters_s16": (C.c_int, [_p, _i32, _i64, _p]),
    "ammc_pack_conv4_dgrad_weight_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_flownet_prep_scratch_doubles": (C.c_int, [_i32]),
    "ammc_flownet_prep_f32": (C.c_int, [_p, _i32, _i32, _i32, _p] + _s3 + [_f32, _p, _p]),
    "ammc_lrelu_f32": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _f32, _p]),
    "ammc_lrelu_s16": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _f32, _p]),
    "ammc_upsample4_bilinear_f32": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _f32, _p, _p]),
    "ammc_lrelu_bwd_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_i32, _i32, _i32, _i32, _f32, _p]),
    "ammc_frames_u8_to_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _i32, _i32, _p]),
    "ammc_flows_to_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _i32, _i32, _p]),
    "ammc_chan_reduce_blocks": (C.c_int, [_i32]),
    "ammc_bn_stats_f32": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _p, _p]),
    "ammc_bn_finalize_f32": (C.c_int, [_p, _i32, _i32, _f32, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p]),
    "ammc_scale_shift_act_f32": (C.c_int, [_p] + _s3 + [_p, _p, _p] + _s3 + [_p] + _s3 + [_i32] * 5 + [_p]),
    "ammc_scale_shift_act_s16_f32": (C.c_int, [_p] + _s3 + [_p, _p, _p] + _s3 + [_p, _p] + _s3 + [_i32] * 5 + [_p]),
    "ammc_scale_shift_act_s16_pool_f32": (C.c_int, [_p] + _s3 + [_p, _p, _p, _p] + _s3 + [_p] + _s3 + [_p] + [_i32] * 5 + [_p]),
    "ammc_bn_bwd_reduce_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p, _p, _p, _p] + [_i32] * 5 + [_p, _p]),
    "ammc_bn_bwd_apply_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p, _p, _p, _p, _p, _i32, _p] + _s3 + [_i32] * 4 + [_p, _p]),
    "ammc_bn_bwd_reduce_bound_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p, _p, _p, _p] + [_i32] * 5 + [_p, _p]),
    "ammc_bn_bwd_finalize_f32": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p]),
    "ammc_bn_bwd_apply_s16_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p, _p, _p, _p, _p, _i32, _p, _p] + _s3 + [_i32] * 4 + [_p, _p, _i32, _p]),
    "ammc_scale_shift_act_s16_pool_supported": (C.c_int, [_i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64]),
    "ammc_bn_bwd_unpool_supported": (C.c_int, [_i32, _i64, _i64, _i64, _i32]),
    "ammc_bn_bwd_reduce_bound_unpool_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_p, _i32, _i32] + [_p, _p, _p, _p] + [_i32] * 5 + [_p, _p]),
    "ammc_bn_bwd_apply_s16_unpool_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_p, _i32, _i32] + [_p, _p, _p, _p, _p, _i32, _p, _p] + _s3 + [_i32] * 4 + [_p, _p, _i32, _p]),
    "ammc_chan_sum_f32": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _p, _p]),
    "ammc_chan_sum_absmax_f32": (C.c_int, [_p] + _s3 + [_i32, _i32, _i32, _i32, _p, _p, _p]),
    "ammc_split_scaled_strided_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_i32, _i32, _i32, _i32, _p, _p, _i32, _p]),
    "ammc_reduce_partials_f32": (C.c_int, [_p, _i32, _i32, _f32, _p, _p]),
    "ammc_reduce_partials_seg_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "ammc_maxpool2x2_bwd_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_i32] * 6 + [_p]),
    "ammc_maxpool2x2_bwd_s16x_f32": (C.c_int, [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_i32] * 6 + [_p]),
    "ammc_maxpool2x2_bwd_idx_f32": (C.c_int, [_p, _p] + _s3 + [_p] + _s3 + [_p] + _s3 + [_i32] * 6 + [_p]),
    "ammc_tanh_bwd_nhwc_f32": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p] + _s3 + [_i32, _p]),
    "ammc_commit_bwd_f32": (C.c_int, [_p, _p, _p, _i32, _p, _p, _p, _i32, _i32, _p]),
    "ammc_codebook_count_f32": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "ammc_codebook_ema_apply_f32": (C.c_int, [_p, _p, _i32, _i32, _f32, _f32, _f32, _p, _p, _p, _p]),
    "ammc_codebook_ema_f32": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _p, _p, _p, _p]),
}

_lib = None


class AmmcHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libammc_hip.so and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmmcHipError(
            f"{LIB_PATH} is missing: build it with `python -m ammcnet_aaai2021_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.ammc_abi_version()
    if got != ABI_VERSION:
        raise AmmcHipError(f"libammc_hip ABI {got} != binding ABI {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ammc_error_string(rc).decode()
        raise AmmcHipError(f"libammc_hip {what} failed: {msg} (code {rc})")
