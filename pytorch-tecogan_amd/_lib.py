"""ctypes binding of libtecogan_hip.so (include/tecogan_hip.h).  No fallback: if the library is missing the
product path raises."""
import ctypes as C
import os

# torch must be imported (and its bundled HIP runtime mapped) BEFORE libtecogan_hip.so is dlopen'ed: the library's
# libamdhip64 dependency then resolves to the runtime torch already loaded.  Loading in the other order maps a second
# HIP runtime and every launch fails with hipErrorNoDevice against torch-allocated memory.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TECOGAN_LIB") or os.path.join(_HERE, "csrc", "libtecogan_hip.so")  # TECOGAN_LIB: A/B builds (tools)

TG_F32, TG_BF16, TG_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_TANH24 = 0, 1, 2, 3, 4
MASK_NONE, MASK_RELU, MASK_LRELU, MASK_BNZ, MASK_RELU_BITS = 0, 1, 2, 3, 4
OUT_NHWC, OUT_NCHW_F32 = 0, 1
TILE_AUTO, TILE_64x256, TILE_64x64, TILE_128x128, TILE_32x128 = 0, 1, 2, 3, 4
TILE_32x64, TILE_64x128, TILE_64x128_8W, TILE_64x64_8W = 5, 6, 7, 8
MAX_TAPS, MAX_CLASSES = 16, 4
WGROUP_C3, WGROUP_CT, WGROUP_C4S2, WGROUP_C3_B128, WGROUP_CT_B128 = 0, 1, 2, 3, 4


class ConvClass(C.Structure):
    _fields_ = [("ooy", C.c_int32), ("oox", C.c_int32), ("ntaps", C.c_int32), ("dy", C.c_int8 * MAX_TAPS),
                ("dx", C.c_int8 * MAX_TAPS), ("widx", C.c_int16 * MAX_TAPS)]


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("N", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("Cin", C.c_int32),
                ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("S", C.c_int32), ("OS", C.c_int32),
                ("ncls", C.c_int32), ("cls", ConvClass * MAX_CLASSES), ("act", C.c_int32), ("mask_mode", C.c_int32),
                ("stats_mode", C.c_int32), ("stats_groups", C.c_int32), ("out_mode", C.c_int32),
                ("c_real", C.c_int32), ("out_n_stride", C.c_int64), ("tile_cfg", C.c_int32),
                ("stats_replicas", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("N", C.c_int32), ("XH", C.c_int32), ("XW", C.c_int32), ("Cx", C.c_int32),
                ("YH", C.c_int32), ("YW", C.c_int32), ("Cy", C.c_int32), ("S", C.c_int32), ("ntaps", C.c_int32),
                ("dy", C.c_int8 * MAX_TAPS), ("dx", C.c_int8 * MAX_TAPS), ("nsplit", C.c_int32),
                ("taps_per_wg", C.c_int32), ("y_sum", C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

_PROTOS = {
    "tg_abi_version": (C.c_int, []),
    "tg_error_string": (C.c_char_p, [_I]),
    "tg_has_experiments": (C.c_int, []),
    "tg_packed_weight_bytes": (_L, [_I, _I, _I, _I]),
    "tg_pack_conv_weights": (_I, [_I, _P, _P, _I, _I, _I, _I, _L, _L, _I, _P, _P]),
    "tg_pack_conv_weights_multi": (_I, [_I, _P, _I, _I, _P]),
    "tg_conv_pick_tile": (_I, [C.POINTER(ConvDesc)]),
    "tg_conv": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "tg_conv3x3_rw": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_conv3x3_cw": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_wgrad_slab_floats": (_L, [C.POINTER(WgradDesc)]),
    "tg_wgrad": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P]),
    "tg_wgrad_multi": (_I, [C.POINTER(WgradDesc), _P, _I, _P]),
    "tg_wgrad_group_slot_floats_v": (_L, [_I]),
    "tg_wgrad_group_v": (_I, [_I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "tg_wgrad_finalize": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _L, _L, _P, _I, _P, _L, _P]),
    "tg_wgrad_finalize_multi": (_I, [_P, _I, _I, _P]),
    "tg_wgrad_fold_items": (_I, [_P, _I, _I, _I, _P]),
    "tg_nchw_to_nhwc": (_I, [_I, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "tg_nhwc_to_nchw": (_I, [_I, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "tg_resblock_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "tg_resblock_fwd_ws": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_conv3x3_rgb": (_I, [_I, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "tg_conv3x3_rgb_bwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_conv3x3_rgb_bwd_slot_floats": (_L, []),
    "tg_conv4s2_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_conv4s2_fwd_capped": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_conv4s2_dgrad": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "tg_conv4s2_dgrad_cw": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "tg_convt_dgrad": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_conv4s2_fwd_cw": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_convt_dgrad_cw": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_convt_fwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_convt_fwd_cw": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "tg_resblock_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "tg_maxpool2": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "tg_up2_bilinear": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "tg_up2_bilinear_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tg_tanh24_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _P]),
    "tg_warp_grid_grad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "tg_up4_planes": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P]),
    "tg_copy_blocks": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "tg_warp_nchw": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tg_gen_input": (_I, [_I, _P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _P]),
    "tg_d_assemble": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_bn_apply": (_I, [_I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "tg_bn_bwd_reduce": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_bn_bwd_apply": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_fc_head_fwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tg_fc_head_bwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tg_d_tail_max_pixels": (_L, []),
    "tg_d_tail_scratch_floats": (_L, [_I, _I, _I]),
    "tg_d_tail_fwd": (_I, [_I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "tg_d_tail_bwd": (_I, [_I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "tg_absdiff_sum": (_I, [_I, _P, _P, _P, _L, _I, _I, _P]),
    "tg_absdiff_sum_multi": (_I, [_I, _P, _I, _I, _P]),
    "tg_content_loss": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _I, _F, _P, _P, _I, _P]),
    "tg_loss_finalize": (_I, [_P, _P, _P, _P, _I, _P, _P, _P]),
    "tg_dlogit_real": (_I, [_P, _P, _I, _P, _P, _P]),
    "tg_reduce_replicas": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "tg_adam": (_I, [_P, _P, _P, _P, _L, _P, _P]),
    "tg_check_finite": (_I, [_P, _L, _P, _P]),
    "tg_adam_scaled": (_I, [_P, _P, _P, _P, _L, _P, _P, _I, _P]),
    "tg_scaler_update": (_I, [_P, _F, _F, _I, _P]),
    "tg_vgg_input": (_I, [_I, _P, _P, _I, _I, _I, _F, _P, _P]),
    "tg_cosine_loss": (_I, [_I, _P, _P, _P, _L, _I, _F, _I, _P, _P, _P]),
    "tg_maxpool2_bwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_vgg_input_grad": (_I, [_I, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "tg_resample_u8": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tg_stream_create_cumask": (_I, [_I, C.POINTER(C.c_void_p)]),
    "tg_stream_destroy": (_I, [_P]),
}

# entry points of the experiments build only (include/tecogan_hip.h, `#ifdef TG_EXPERIMENTS`; csrc/build.sh --experiments)
_PROTOS_EXPERIMENTS = {
    "tg_bn_bwd_coop_max_workgroups": (_I, []),
    "tg_bn_bwd_coop": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "tg_resblock2_fwd": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "tg_resblock2_fwd_ws": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tg_resblock_bwd_pp": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_bn_bwd_fused": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "tg_bn_bwd_fused_max_pixels": (_I, []),
}

EXPORTED = tuple(_PROTOS.keys())
ABI_VERSION = 4   # TG_ABI_VERSION of include/tecogan_hip.h
_lib = None


class TecoganHipError(RuntimeError):
    pass


def load():
    """Load the C-ABI library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TecoganHipError(
            f"{LIB_PATH} is missing: build it with pytorch-tecogan_amd/csrc/build.sh (or __graft_entry__.build()). "
            "There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    # the version first: a stale library must fail with this message, not with an AttributeError on a missing symbol
    try:
        lib.tg_abi_version.restype = C.c_int
        ver = lib.tg_abi_version()
    except AttributeError:
        ver = None
    if ver != ABI_VERSION:
        raise TecoganHipError(f"{LIB_PATH}: ABI version {ver}, this package needs {ABI_VERSION} - rebuild it with "
                              "pytorch-tecogan_amd/csrc/build.sh")
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.tg_has_experiments():
        for name, (res, args) in _PROTOS_EXPERIMENTS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def has_experiments():
    """True when the loaded library is the experiments build (TECOGAN_LIB=.../libtecogan_hip_experiments.so)"""
    return bool(load().tg_has_experiments())


def check(code, what=""):
    if code != 0:
        msg = load().tg_error_string(int(code)).decode()
        raise TecoganHipError(f"{what}: {msg} (status {code})")
