"""ctypes binding of libmrfa_hip.so (C ABI: include/mrfa_hip.h).

The library is the product: if it cannot be loaded the import of any op raises -- there is no CPU or eager fallback.
Kernels are launched on torch's current HIP stream with raw device pointers taken from torch tensors; torch is only
the allocator / stream owner here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRFA_HIP_LIB: another build of the same library (kernel A/B experiments: tools/ab_lib.sh); the default is the in-tree build
LIB_PATH = os.environ.get("MRFA_HIP_LIB") or os.path.join(_HERE, "_lib", "libmrfa_hip.so")

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class ConvParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("ups", C.c_int),
        ("N", C.c_int), ("Cin", C.c_int),
        ("w", C.c_void_p), ("w_ld", C.c_int), ("w_tap", C.c_longlong), ("w_rows", C.c_int),
        ("y", C.c_void_p), ("ldy", C.c_int), ("Cout", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("R", C.c_int), ("S", C.c_int), ("pad", C.c_int),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_relu", C.c_int),
        ("bias", C.c_void_p), ("out_scale", C.c_void_p), ("out_shift", C.c_void_p), ("relu", C.c_int),
        ("res", C.c_void_p), ("ldr", C.c_int),
        ("stats", C.c_void_p),
        ("alpha", C.c_float), ("accumulate", C.c_int),
        ("nbatch", C.c_int), ("x_bs", C.c_longlong), ("w_bs", C.c_longlong), ("y_bs", C.c_longlong),
        ("splitk", C.c_int),
        ("ktab", C.c_void_p), ("kflat", C.c_int), ("tile", C.c_int),
        ("w_split", C.c_void_p), ("w_piece", C.c_longlong),
        ("mask", C.c_void_p), ("ldm", C.c_int),
        ("w_phase", C.c_void_p), ("w_phase_piece", C.c_longlong),
        ("stride", C.c_int),
        ("w_wino", C.c_void_p), ("w_wino_piece", C.c_longlong),
        ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p), ("fin_rmean", C.c_void_p), ("fin_rvar", C.c_void_p),
        ("fin_momentum", C.c_float), ("fin_eps", C.c_float), ("fin_count", C.c_longlong),
        ("fin_scale", C.c_void_p), ("fin_shift", C.c_void_p), ("fin_mean", C.c_void_p), ("fin_invstd", C.c_void_p),
        ("fin_counter", C.c_void_p),
        ("bst_x", C.c_void_p), ("bst_ldx", C.c_int), ("bst_scale", C.c_void_p), ("bst_shift", C.c_void_p), ("bst_mean", C.c_void_p),
        ("bst_invstd", C.c_void_p), ("bst_relu", C.c_int),
        ("groups", C.c_int),
        ("sk_ticket", C.c_void_p), ("y_zero", C.c_int),
    ]


class WgradParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("ups", C.c_int),
        ("N", C.c_int), ("Cin", C.c_int),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("in_relu", C.c_int),
        ("dy", C.c_void_p), ("ldy", C.c_int), ("Cout", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("R", C.c_int), ("S", C.c_int), ("pad", C.c_int),
        ("dw", C.c_void_p), ("dbias", C.c_void_p), ("alpha", C.c_float),
        ("nbatch", C.c_int), ("x_bs", C.c_longlong), ("dy_bs", C.c_longlong), ("dw_bs", C.c_longlong),
        ("ksplit", C.c_int), ("ktab", C.c_void_p), ("kflat", C.c_int), ("tile8_off", C.c_int),
        ("ws", C.c_void_p), ("ws_bytes", C.c_longlong),
        ("stride", C.c_int),
        ("groups", C.c_int),
    ]


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p * 3), ("mode", C.c_int * 3), ("ndst", C.c_int),
                ("Cout", C.c_int), ("Cin", C.c_int), ("R", C.c_int), ("S", C.c_int)]


class ResizeSumTerm(C.Structure):
    _fields_ = [("src", C.c_void_p), ("lds", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int), ("mul", C.c_float)]


class ResizeSumDesc(C.Structure):
    """mrfa_resize_sum_desc: dst (=|+=) sum_k mul_k resize(src_k)"""
    _fields_ = [("dst", C.c_void_p), ("ldd", C.c_int), ("N", C.c_int), ("Hd", C.c_int), ("Wd", C.c_int), ("C", C.c_int),
                ("nterm", C.c_int), ("overwrite", C.c_int), ("term", ResizeSumTerm * 4)]


class UnpackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("T", C.c_int), ("fewout", C.c_int)]


class BnActParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int), ("pool", C.c_int),
        ("blend_a", C.c_void_p), ("lda", C.c_int), ("occ", C.c_void_p), ("ldo", C.c_int),
        ("y", C.c_void_p), ("ldy", C.c_int),
        ("res", C.c_void_p), ("ldr", C.c_int),
        ("groups", C.c_int),
    ]


class BnBwdParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int), ("pool", C.c_int),
        ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p),
        ("dy", C.c_void_p), ("lddy", C.c_int),
        ("blend_a", C.c_void_p), ("lda", C.c_int), ("occ", C.c_void_p), ("ldo", C.c_int),
        ("dblend_a", C.c_void_p), ("ldda", C.c_int), ("docc", C.c_void_p), ("lddo", C.c_int),
        ("red", C.c_void_p), ("dx", C.c_void_p), ("lddx", C.c_int),
        ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("train", C.c_int), ("phase", C.c_int), ("dx_overwrite", C.c_int),
        ("res", C.c_void_p), ("ldr", C.c_int), ("dres", C.c_void_p), ("lddr", C.c_int),
        ("sync", C.c_void_p), ("red_world", C.c_int), ("red_all", C.c_int), ("groups", C.c_int),
    ]


class PriorParams(C.Structure):
    _fields_ = [
        ("kd", C.c_void_p), ("ks", C.c_void_p), ("jd", C.c_void_p), ("js", C.c_void_p), ("bg", C.c_void_p),
        ("src", C.c_void_p), ("lds", C.c_int),
        ("B", C.c_int), ("K", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
        ("inv_var", C.c_float),
        ("motions", C.c_void_p), ("ldm", C.c_int),
        ("inp", C.c_void_p), ("ldi", C.c_int),
        ("sparse", C.c_void_p),
        ("dinp", C.c_void_p), ("lddi", C.c_int),
        ("dmotions", C.c_void_p), ("dsparse", C.c_void_p),
        ("dkd", C.c_void_p), ("dks", C.c_void_p), ("djd", C.c_void_p), ("djs", C.c_void_p), ("dbg", C.c_void_p),
    ]


_V, _I, _L, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float

_SIGNATURES = {
    "mrfa_version": ([], C.c_int),
    "mrfa_last_error": ([], C.c_char_p),
    "mrfa_conv2d_nhwc": ([_V, C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_phase_dgrad_supported": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_last_config": ([], C.c_int),
    "mrfa_set_mfma_mode": ([_I], C.c_int),
    "mrfa_get_mfma_mode": ([], C.c_int),
    "mrfa_set_tuning": ([C.c_char_p, _I], C.c_int),
    "mrfa_conv2d_wgrad_nhwc": ([_V, C.POINTER(WgradParams)], C.c_int),
    "mrfa_conv2d_wgrad_multi": ([_V, _V, _I], C.c_int),
    "mrfa_conv_fewout_fwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _V, _V, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_conv_fewout_wgrad": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _I, _I, _I, _V, _V], C.c_int),
    "mrfa_conv_fewout_dgrad": ([_V, _V, _I, _I, _I, _I, _I, _V, _V, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_conv2d_mask_supported": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_stride_supported": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_wgrad_stride_supported": ([C.POINTER(WgradParams)], C.c_int),
    "mrfa_conv2d_wgrad_groups_supported": ([C.POINTER(WgradParams)], C.c_int),
    "mrfa_conv2d_wgrad_lean_supported": ([C.POINTER(WgradParams)], C.c_int),
    "mrfa_conv_fewout_dgrad_supported": ([_I, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_pack_conv_weight": ([_V, _V, _V, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_pack_conv_weights_multi": ([_V, C.POINTER(PackDesc), _I], C.c_int),
    "mrfa_unpack_wgrads_multi": ([_V, C.POINTER(UnpackDesc), _I], C.c_int),
    "mrfa_build_ktab": ([c_int_p, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_bn_stats": ([_V, _V, _I, _L, _I, _V], C.c_int),
    "mrfa_bn_finalize": ([_V, _V, _L, _V, _V, _V, _V, _F, _F, _I, _I, _V, _V, _V, _V], C.c_int),
    "mrfa_bn_finalize_groups": ([_V, _V, _L, _V, _V, _V, _V, _F, _F, _I, _I, _V, _V, _V, _V], C.c_int),
    "mrfa_bn_act_fwd": ([_V, C.POINTER(BnActParams)], C.c_int),
    "mrfa_bn_act_bwd": ([_V, C.POINTER(BnBwdParams)], C.c_int),
    "mrfa_grid_sample_fwd": ([_V, _V, _I, _L, _I, _I, _I, _I, _V, _I, _I, _I, _I, _V, _I, _I], C.c_int),
    "mrfa_grid_sample_bwd": ([_V, _V, _I, _L, _I, _I, _I, _I, _V, _I, _I, _I, _I, _V, _I, _I, _V, _I, _L, _V, _I], C.c_int),
    "mrfa_resize_bilinear_fwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _I, _I, _F, _I], C.c_int),
    "mrfa_resize_bilinear_bwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _I, _I, _F], C.c_int),
    "mrfa_resize_sum_multi": ([_V, C.POINTER(ResizeSumDesc), _I], C.c_int),
    "mrfa_resize_sum_multi_bwd": ([_V, C.POINTER(ResizeSumDesc), _I], C.c_int),
    "mrfa_corr_lookup_fwd": ([_V, _V, _V, _I, _I, _V, _I, _L, _I, _V, _I], C.c_int),
    "mrfa_corr_lookup_bwd": ([_V, _V, _V, _I, _I, _V, _I, _L, _I, _V, _I, _V, _V, _V, _I], C.c_int),
    "mrfa_nchw_to_nhwc": ([_V, _V, _V, _I, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_nhwc_to_nchw": ([_V, _V, _I, _V, _I, _I, _I, _I, _I], C.c_int),
    "mrfa_avgpool2_fwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_sumpool2_acc": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _F], C.c_int),
    "mrfa_unpool2_acc": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _F], C.c_int),
    "mrfa_bias_act": ([_V, _V, _I, _L, _I, _V, _I, _V, _I, _V], C.c_int),
    "mrfa_act_bwd": ([_V, _V, _I, _V, _I, _L, _I, _I, _V, _I, _I], C.c_int),
    "mrfa_copy_view": ([_V, _V, _I, _L, _I, _V, _I, _F, _I], C.c_int),
    "mrfa_timestamp": ([_V, _V], C.c_int),
    "mrfa_conv2d_bwdstats_supported": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_groups_supported": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_reads_fp32_weights": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_conv2d_split_k": ([C.POINTER(ConvParams)], C.c_int),
    "mrfa_bn_param_grad": ([_V, _V, _I, _V, _V], C.c_int),
    "mrfa_bn_param_grad_groups": ([_V, _V, _I, _I, _V, _V], C.c_int),
    "mrfa_warp_frame_reflect": ([_V, _V, _I, _I, _I, _I, _V, _I, _I, _V], C.c_int),
    "mrfa_blend_fwd": ([_V, _V, _I, _V, _I, _V, _I, _L, _I, _V, _I], C.c_int),
    "mrfa_blend_bwd": ([_V, _V, _I, _V, _I, _V, _I, _V, _I, _L, _I, _V, _I, _V, _I, _V, _I], C.c_int),
    "mrfa_antialias_down": ([_V, _V, _I, _I, _I, _I, _V, _I, _I, _V, _I], C.c_int),
    "mrfa_colsum": ([_V, _V, _I, _L, _I, _V], C.c_int),
    "mrfa_subsample_fwd": ([_V, _V, _I, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_subsample_bwd": ([_V, _V, _I, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_upsample_add_act_fwd": ([_V, _V, _I, _I, _I, _I, _I, _I, _V, _I, _I, _V, _I], C.c_int),
    "mrfa_upsample_add_act_bwd": ([_V, _V, _I, _V, _I, _I, _I, _I, _I, _I, _I, _V, _I, _V, _I], C.c_int),
    "mrfa_layernorm_fwd": ([_V, _V, _I, _L, _I, _V, _V, _F, _V, _I, _V, _V], C.c_int),
    "mrfa_layernorm_bwd": ([_V, _V, _I, _V, _I, _L, _I, _V, _V, _V, _V, _I, _V, _V, _V], C.c_int),
    "mrfa_gelu_fwd": ([_V, _V, _I, _L, _I, _V, _I], C.c_int),
    "mrfa_gelu_bwd": ([_V, _V, _I, _V, _I, _L, _I, _V, _I], C.c_int),
    "mrfa_attention_fwd": ([_V, _V, _I, _I, _I, _I, _I, _F, _V, _I, _V], C.c_int),
    "mrfa_attention_bwd": ([_V, _V, _I, _V, _I, _V, _I, _V, _V, _I, _I, _I, _I, _F, _V, _I], C.c_int),
    "mrfa_maxpool2_fwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_maxpool2_bwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _V, _I], C.c_int),
    "mrfa_maxpool3s2_fwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I], C.c_int),
    "mrfa_maxpool3s2_bwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _V, _I], C.c_int),
    "mrfa_l1_diff_fwd": ([_V, _V, _I, _V, _I, _L, _I, C.c_double, _V], C.c_int),
    "mrfa_l1_diff_bwd": ([_V, _V, _I, _V, _I, _L, _I, _V, _F, _V, _I], C.c_int),
    "mrfa_antialias_down_bwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _I, _I, _V], C.c_int),
    "mrfa_kp_gaussian_fwd": ([_V, _V, _V, _I, _I, _I, _I, _F, _V, _I], C.c_int),
    "mrfa_kp_gaussian_bwd": ([_V, _V, _I, _I, _I, _I, _F, _V, _I, _V, _V], C.c_int),
    "mrfa_prior_motion_fwd": ([_V, C.POINTER(PriorParams)], C.c_int),
    "mrfa_prior_motion_bwd": ([_V, C.POINTER(PriorParams)], C.c_int),
    "mrfa_softmax_combine_fwd": ([_V, _V, _I, _V, _I, _I, _I, _I, _I, _V, _V, _V], C.c_int),
    "mrfa_softmax_combine_bwd": ([_V, _V, _I, _I, _I, _I, _I, _V, _V, _V, _V, _V, _I, _V], C.c_int),
    "mrfa_kp_head_fwd": ([_V, _V, _I, _V, _I, _I, _I, _I, _I, _F, _V, _V, _V], C.c_int),
    "mrfa_kp_head_bwd": ([_V, _V, _I, _V, _I, _I, _I, _I, _I, _F, _V, _V, _V, _V, _V, _V, _I, _V, _I], C.c_int),
    "mrfa_adam_prepare": ([_V, _V, _I, C.c_double, C.c_double], C.c_int),
    "mrfa_grad_absmax": ([_V, _V, _L, _V, _I], C.c_int),
    "mrfa_adam_flat": ([_V, _V, _V, _V, _V, _L, _V, C.c_double, C.c_double, _F, _F, _I, _F], C.c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
LN_SLOTS = 16          # MRFA_LN_SLOTS
RESIZE_SUM_TERMS = 4   # MRFA_RESIZE_SUM_TERMS
ABI_VERSION = 9        # MRFA_ABI_VERSION of include/mrfa_hip.h: the struct layouts above mirror THAT header; lib() refuses any other library

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Loads libmrfa_hip.so once; raises loudly if it is absent (the HIP path is the only path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m mrfa_amd.build` (hipcc --offload-arch=gfx950). "
                "mrfa_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the ABI and the header drift apart
            fn.argtypes = argtypes
            fn.restype = restype
        if L.mrfa_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} speaks ABI version {L.mrfa_version()}, this binding was written against {ABI_VERSION} "
                               "(include/mrfa_hip.h MRFA_ABI_VERSION): rebuild the library (`python -m mrfa_amd.build --force`)")
        _lib = L
        mode = os.environ.get("MRFA_MFMA", DEFAULT_MFMA)
        if mode not in MFMA_MODES:
            raise ValueError(f"MRFA_MFMA={mode!r}: expected one of {sorted(MFMA_MODES)}")
        check(L.mrfa_set_mfma_mode(MFMA_MODES[mode]), "mrfa_set_mfma_mode")
    return _lib


# matrix-pipe selection for the 128 x 128 chunked conv tiles (include/mrfa_hip.h, mrfa_set_mfma_mode):
#   "f32"     v_mfma_f32_32x32x2_f32 on fp32 operands
#   "bf16x6"  fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulate
#             (fp32-accurate, csrc/conv_split.hip)
#   "bf16x3"  the same kernels with the three leading products only: ~1e-5 relative product error (opt-in)
#   "bf16"    operands rounded to bf16, one product (a bf16 autocast's arithmetic; fp32 accumulate / storage): BASELINE config 4
STATS_SLOTS = 32       # MRFA_STATS_SLOTS of include/mrfa_hip.h: BatchNorm statistics buffers are [STATS_SLOTS][2C] doubles
FIN_WORDS = 16         # MRFA_FIN_WORDS: zeroed 32-bit ticket words behind a statistics buffer whose finalize rides in the producing launch
MFMA_MODES = {"f32": 0, "bf16x6": 1, "bf16x3": 2, "bf16": 3}
DEFAULT_MFMA = "bf16x6"


def set_mfma_mode(mode: str):
    check(lib().mrfa_set_mfma_mode(MFMA_MODES[mode]), "mrfa_set_mfma_mode")


def mfma_mode() -> str:
    m = lib().mrfa_get_mfma_mode()
    return next(k for k, v in MFMA_MODES.items() if v == m)


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().mrfa_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t, offset_elems: int = 0):
    """Raw device pointer of a tensor (+ element offset); None -> NULL."""
    if t is None:
        return None
    return t.data_ptr() + 4 * offset_elems if offset_elems else t.data_ptr()
