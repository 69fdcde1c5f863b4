"""ctypes binding of libgvcnn_hip.so (the C ABI declared in include/gvcnn_hip.h).

The HIP library is the only compute path of this package: if it cannot be loaded the import
fails loudly — there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GVCNN_HIP_LIB: another build of the SAME library, e.g. the profiling build of tools/phase_times.py — never a fallback)
LIB_PATH = os.environ.get("GVCNN_HIP_LIB") or os.path.join(_HERE, "libgvcnn_hip.so")

GV_F32, GV_BF16, GV_F16 = 0, 1, 2
GV_CONV_RELU, GV_CONV_RELU2, GV_CONV_SPLIT, GV_CONV_X_F32 = 1, 2, 4, 8
GV_CONV_X_P3, GV_CONV_Y_P3, GV_CONV_Y2_P3, GV_CONV_MAXPOOL3S2, GV_CONV_MAXPOOL3S2_SAME = 16, 32, 64, 128, 256
GV_CONV_POOL_ACT2 = 512
GV_CHAIN_PROJ = 4096
GV_MATH_F32, GV_MATH_BF16X3, GV_MATH_BF16X2, GV_MATH_BF16X1 = 0, 1, 2, 3
GV_POOL_MAX, GV_POOL_AVG, GV_POOL_AVG_RELU = 0, 1, 2
GV_POOL_BWD_STORE = 0x100
GV_POOL_X_P3 = 0x200
GV_POOL_Y_P3 = 0x400
GV_ACCUM_ZEROED = 0x100
GV_ACCUM_RAW_Z = 0x200
GV_BN_STATS_FWD, GV_BN_STATS_BWD, GV_BN_STATS_MAX_SEG = 1, 2, 8
GV_OK, GV_E_BADARG, GV_E_UNSUPPORTED, GV_E_ALIGN, GV_E_PLAN = 0, -1, -2, -3, -4
GV_VIEWPOOL_MAX, GV_VIEWPOOL_MEAN = 0, 1
GV_ORDER_SHAPE_MAJOR, GV_ORDER_VIEW_MAJOR = 0, 1
GV_WEIGHT_COUNT, GV_WEIGHT_MEAN_SCORE = 0, 1
GV_ABI_VERSION = 1


class GvError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed with code %d: %s" % (what, code, error_string(code)))


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb", "ih", "iw", "cin", "x_ld", "kh", "kw", "stride", "pad_t", "pad_l",
        "oh", "ow", "cout", "y_ld", "res_ld", "y2_ld", "flags", "dtype", "split_col", "tile_cfg",
        "math_mode", "in_dilation", "relu_cols", "y_step", "y_py", "y_px", "y_ih", "y_iw")]


class ChainDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("m", "d", "x_ld", "res_ld", "y_ld", "z_ld", "dtype", "flags", "tile_cfg")]


class UnitDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nb", "ih", "iw", "d", "x_ld", "res_ld", "y_ld", "z_ld", "dtype", "flags", "tile_cfg")]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("kh", C.c_int32), ("kw", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("flipped", C.c_int32), ("first_block", C.c_int32), ("k_off", C.c_int32),
                ("k_total", C.c_int32), ("w_ld", C.c_int32), ("sub_step", C.c_int32), ("sub_r0", C.c_int32),
                ("sub_s0", C.c_int32), ("src_kw", C.c_int32), ("reserved2", C.c_int32)]


class BnMovingJob(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("var", C.c_void_p), ("counts", C.c_void_p), ("moving_mean", C.c_void_p),
                ("moving_var", C.c_void_p), ("c", C.c_int32), ("first_block", C.c_int32), ("ld", C.c_int32),
                ("reserved", C.c_int32)]


class BnStatsSeg(C.Structure):
    _fields_ = [("c0", C.c_int32), ("c1", C.c_int32), ("z_ld", C.c_int32), ("reserved", C.c_int32), ("z", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("acc", C.c_void_p)]


class BnStats(C.Structure):
    _fields_ = [("mode", C.c_int32), ("groups", C.c_int32), ("nseg", C.c_int32), ("reserved", C.c_int32),
                ("seg", BnStatsSeg * 8)]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb", "ih", "iw", "c", "x_ld", "kh", "kw", "stride", "pad_t", "pad_l",
        "oh", "ow", "y_ld", "mode", "dtype")]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); every symbol include/gvcnn_hip.h declares
SIGNATURES = {
    "gv_abi_version": (C.c_int, []),
    "gv_error_string": (C.c_char_p, [C.c_int]),
    "gv_packed_filter_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "gv_pack_filter_hwio": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "gv_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gv_conv2d_fwd_xpre": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gv_conv2d_fwd_bnstats": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.POINTER(BnStats), _P]),
    "gv_pool2d_fwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P]),
    "gv_scale_shift_act": (C.c_int, [_P, _L, _I, _I, _P, _P, _I, _P, _I, _I, _P]),
    "gv_global_avg_pool": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "gv_view_score_partial": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P]),
    "gv_view_score_finalize": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "gv_group_assign": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gv_group_weight": (C.c_int, [_P, _I, _I, _P, _P]),
    "gv_view_pool_fuse_fwd": (C.c_int, [_P, _I, _I, _L, _L, _L, _P, _I, _P, _I, _F, _P, _P, _I, _P]),
    "gv_view_score_per_shape": (C.c_int, [_P, _I, _P, _P]),
    "gv_group_assign_per_shape": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gv_view_pool_fuse_fwd_per_shape": (C.c_int, [_P, _I, _I, _L, _L, _L, _P, _I, _P, _I, _F, _P, _P, _I, _P]),
    "gv_preprocess_views": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "gv_png_unfilter": (C.c_int, [_P, _I, _I, _I, _P]),
    "gv_eval_metrics": (C.c_int, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "gv_dense_fwd": (C.c_int, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "gv_bn_stats_grouped": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gv_bn_sums_grouped": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "gv_bn_finalize_grouped": (C.c_int, [_P, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P]),
    "gv_bn_relu_bwd_sums_grouped": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "gv_bn_relu_bwd_apply_grouped": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P,
                                               _P, _P]),
    "gv_scale": (C.c_int, [_P, _L, _F, _P]),
    "gv_bn_update_moving": (C.c_int, [_P, _P, _P, _I, _I, _F, _P, _P, _P]),
    "gv_scale_shift_act_grouped": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P]),
    "gv_bn_relu_bwd_grouped": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P,
                                         _P, _P]),
    "gv_accumulate": (C.c_int, [_P, _I, _P, _I, _L, _I, _P]),
    "gv_bias_grad": (C.c_int, [_P, _I, _L, _I, _P, _P, _P]),
    "gv_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P]),
    "gv_conv2d_wgrad_ws": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _I, _P, _P, C.c_int64, _P]),
    "gv_pool2d_bwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _I, _P, _I, _P]),
    "gv_pool2d_fwd_argmax": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P, _P]),
    "gv_pool2d_bwd_argmax": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _I, _P, _I, _P]),
    "gv_bn_bwd_coeffs_t": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "gv_pool2d_bwd_argmax_bn": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "gv_view_pool_fuse_bwd": (C.c_int, [_P, _P, _I, _I, _L, _L, _L, _P, _I, _P, _I, _P, _P]),
    "gv_view_pool_fuse_bwd_per_shape": (C.c_int, [_P, _P, _I, _I, _L, _L, _L, _P, _I, _P, _I, _P, _P]),
    "gv_global_avg_pool_bwd": (C.c_int, [_P, _I, _I, _I, _P, _I, _P]),
    "gv_softmax_ce": (C.c_int, [_P, _P, _I, _I, _P, _P, _P]),
    "gv_dense_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "gv_sgd_momentum": (C.c_int, [_P, _P, _P, _L, _F, _F, _F, _P]),
    "gv_bn_sums_grouped_t": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "gv_scale_shift_act_grouped_t": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P]),
    "gv_bn_relu_bwd_sums_grouped_t": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "gv_bn_relu_bwd_apply_grouped_t": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P,
                                                 _P, _P, _P, _I, _I, _P]),
    "gv_accumulate_t": (C.c_int, [_P, _I, _P, _I, _L, _I, _I, _P]),
    "gv_bias_grad_t": (C.c_int, [_P, _I, _L, _I, _P, _P, _I, _P]),
    "gv_view_pool_fuse_bwd_t": (C.c_int, [_P, _P, _I, _I, _L, _L, _L, _P, _I, _P, _I, _P, _I, _I, _P]),
    "gv_pack_filters_batched": (C.c_int, [_P, _I, _P, _I, _I, _P]),
    "gv_bn_update_moving_batched": (C.c_int, [_P, _I, _P, _I, _I, _F, _P]),
    "gv_bn_finalize_apply_grouped_t": (C.c_int, [_P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P,
                                                 _I, _P]),
    "gv_plan_create": (C.c_int, [C.POINTER(_P)]),
    "gv_plan_destroy": (None, [_P]),
    "gv_plan_num_ops": (C.c_int, [_P]),
    "gv_plan_add_conv": (C.c_int, [_P, C.POINTER(ConvDesc), _I, _L, _I, _L, _I, _L, _L, _I, _L, _I, _L,
                                   _I, _L, _L, _L]),
    "gv_plan_add_chain": (C.c_int, [_P, C.POINTER(ChainDesc), _I, _L, _I, _L, _L, _I, _L, _L, _L, _L, _L, _L, _I, _L, _I, _L,
                                    _I, _L]),
    "gv_bottleneck_chain_fwd": (C.c_int, [C.POINTER(ChainDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gv_plan_add_unit": (C.c_int, [_P, C.POINTER(UnitDesc), _I, _L, _I, _L, _L, _L, _I, _L, _L, _L, _L, _L, _L, _L, _L, _I, _L,
                                   _I, _L, _I, _L]),
    "gv_bottleneck_unit_fwd": (C.c_int, [C.POINTER(UnitDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gv_plan_set_conv_tile": (C.c_int, [_P, _I, _I]),
    "gv_plan_set_conv_xpre": (C.c_int, [_P, _I, _L, _L]),
    "gv_plan_set_schedule": (C.c_int, [_P, _I, _I, C.POINTER(_I), _I]),
    "gv_plan_add_pool": (C.c_int, [_P, C.POINTER(PoolDesc), _I, _L, _I, _L]),
    "gv_plan_add_scale_shift_act": (C.c_int, [_P, _L, _I, _I, _I, _I, _I, _I, _L, _I, _L, _L, _I, _L]),
    "gv_plan_run": (C.c_int, [_P, C.POINTER(_P), _I, _P]),
    "gv_plan_run_range": (C.c_int, [_P, _I, _I, C.POINTER(_P), _I, _P]),
    "gv_capture_begin": (C.c_int, [_P]),
    "gv_capture_end": (C.c_int, [_P, C.POINTER(_P)]),
    "gv_graph_launch": (C.c_int, [_P, _P]),
    "gv_graph_destroy": (None, [_P]),
    "gv_conv2d_time": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, C.POINTER(_F), _P]),
    "gv_plan_time": (C.c_int, [_P, _I, _I, C.POINTER(_P), _I, _I, C.POINTER(_F), _P]),
    "gv_plan_time_each": (C.c_int, [_P, C.POINTER(_P), _I, _I, C.POINTER(_F), _P]),
}
# tuning hooks (exported, not part of the drop-in surface)
TUNING = {
    "gv_conv2d_set_tile_override": (None, [C.c_int]),
    "gv_conv2d_set_debug": (None, [C.c_int]),
    "gv_bottleneck_chain_set_debug": (None, [C.c_int]),
    "gv_conv2d_num_tile_cfgs": (C.c_int, [C.c_int]),
    "gv_conv2d_special_tile_cfg": (C.c_int, [C.c_int]),
    "gv_conv2d_wgrad_set_v1": (None, [C.c_int]),
    "gv_conv2d_wgrad_set_lp_f32": (None, [C.c_int]),
    "gv_conv2d_wgrad_num_cfgs": (C.c_int, [C.c_int]),
    "gv_conv2d_wgrad_set_strip_taps": (None, [C.c_int]),
    "gv_pool2d_bwd_set_scatter": (None, [C.c_int]),
    "gv_pool2d_set_rows": (None, [C.c_int]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libgvcnn_hip.so is not built (%s). Run `python gvcnn-tf_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for table in (SIGNATURES, TUNING):
        for name, (res, args) in table.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise ImportError("libgvcnn_hip.so does not export %s — rebuild it" % name)
            fn.restype = res
            fn.argtypes = args
    if lib.gv_abi_version() != GV_ABI_VERSION:
        raise ImportError("libgvcnn_hip.so ABI version %d != %d — rebuild it"
                          % (lib.gv_abi_version(), GV_ABI_VERSION))
    _lib = lib
    return lib


def error_string(code):
    try:
        return load().gv_error_string(int(code)).decode()
    except Exception:
        return "?"


def check(code, what):
    if code != 0:
        raise GvError(code, what)
