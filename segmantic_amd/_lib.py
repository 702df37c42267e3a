"""ctypes binding of ``libsegmi.so`` (the C-ABI declared in ``include/segmi.h``).

The library is the product: there is no CPU or eager-PyTorch fallback.  Importing this module
raises ``ImportError`` when the shared object is missing or does not export the full ABI, and
every call raises ``RuntimeError`` with ``segmi_last_error()`` when the native side fails.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# PyTorch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  libsegmi.so is linked
# against libamdhip64.so.7 by soname, so when torch's copy is already in the process the loader
# binds libsegmi to THAT instance and both share one runtime (streams, device memory).  Loaded the
# other way round the process ends up with two HIP runtimes and libsegmi's sees no device.
import torch  # noqa: F401,E402

SEGMI_F32 = 0
SEGMI_BF16 = 1

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libsegmi.so"


class Act(C.Structure):
    """``segmi_act``: NDHWC activation view."""

    _fields_ = [("data", C.c_void_p), ("n", C.c_int32), ("d", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("ld", C.c_int32)]


class WpackDesc(C.Structure):
    """``segmi_wpack_desc``: one entry of the batched weight-pack table."""

    _fields_ = [("w_src", C.c_void_p), ("scale", C.c_void_p), ("packed", C.c_void_p),
                ("kind", C.c_int32), ("cin_k", C.c_int32), ("cout_k", C.c_int32),
                ("ksize", C.c_int32), ("w_src2", C.c_void_p), ("cout_split", C.c_int32),
                ("reserved", C.c_int32)]


class InAffine(C.Structure):
    """``segmi_in_affine``: BatchNorm-apply + PReLU of the producer, applied by the consumer's staging."""

    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("prelu_alpha", C.c_void_p)]


class Windows(C.Structure):
    """``segmi_windows``: the samples of an input are window views into a larger single-channel volume."""

    _fields_ = [("count", C.c_int32), ("row_stride", C.c_int32), ("plane_stride", C.c_int64),
                ("offset", C.c_int64 * 32)]


class BnFin(C.Structure):
    """``segmi_bn_fin``: BatchNorm statistics finalised by the launch that produces the partial rows."""

    _fields_ = [("count", C.c_double), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("momentum", C.c_float),
                ("eps", C.c_float), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("scale", C.c_void_p),
                ("shift", C.c_void_p)]


class BnBwdFin(C.Structure):
    """``segmi_bn_bwd_fin``: BatchNorm-backward sums finalised by the launch that produces them."""

    _fields_ = [("count", C.c_double), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dalpha", C.c_void_p),
                ("coef", C.c_void_p)]


class BnBwdSums(C.Structure):
    """``segmi_bn_bwd_sums``: BatchNorm-backward reduction fused into an input-gradient conv's epilogue."""

    _fields_ = [("x", C.POINTER(Act)), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("prelu_alpha", C.c_void_p), ("partials", C.c_void_p),
                ("fin", C.POINTER(BnBwdFin))]


_P = C.c_void_p
_AP = C.POINTER(Act)
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes); mirrors include/segmi.h one to one
SIGNATURES = {
    "segmi_version": (_i, []),
    "segmi_last_error": (C.c_char_p, []),
    "segmi_stream_create_cumask": (_i, [_i, C.POINTER(C.c_void_p)]),
    "segmi_stream_destroy": (_i, [_P]),
    "segmi_wpack_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "segmi_wpack": (_i, [_i, _i, _P, _P, _i, _i, _i, _P, _P]),
    "segmi_wpack_batch": (_i, [_i, C.POINTER(WpackDesc), _i, _P, _i, _P]),
    "segmi_conv3d_stats_rows": (_i, [_i, _AP, _AP, _i, _i]),
    "segmi_conv3d_in_affine_ok": (_i, [_i, _AP, _AP, _i, _i]),
    "segmi_conv3d_fwd_kernel_name": (C.c_char_p, [_i, _AP, _AP, _i, _i]),
    "segmi_conv3d_split_act_ok": (_i, [_i, _AP, _AP, _i, _i]),
    "segmi_conv3d_fwd_split_act": (_i, [_i, _AP, _AP, _P, _P, _P, _i, _i, _i, _P, _P, C.POINTER(BnFin), _P]),
    "segmi_conv3d_fwd": (_i, [_i, _AP, _AP, _P, _P, _i, _P, _P, _AP, _P, _i, _i, C.POINTER(InAffine),
                              C.POINTER(BnBwdSums), C.POINTER(BnFin), _P]),
    "segmi_conv3d_bn_bwd_sums_ok": (_i, [_i, _AP, _AP, _i, _i]),
    "segmi_dectop_ok": (_i, [_i, _AP, _AP]),
    "segmi_dectop_fwd": (_i, [_i, _AP, _AP, _P, _P, _P, _i, _P, _P, _P]),
    "segmi_conv3d_pair_ok": (_i, [_i, _AP, _AP, _AP]),
    "segmi_conv3d_fwd_pair": (_i, [_i, _AP, _AP, _P, _P, _P, _P, _AP, _P, _P, _i, C.POINTER(BnFin),
                                   C.POINTER(Windows), _P]),
    "segmi_convT3d_stats_rows": (_i, [_i, _AP, _AP]),
    "segmi_convT3d_fwd": (_i, [_i, _AP, _AP, _P, _P, _P, _P, _AP, _P, C.POINTER(BnFin), _P]),
    "segmi_conv3d_wgrad_workspace": (_i64, [_i, _AP, _AP, _i, _i, _i]),
    "segmi_conv3d_wgrad": (_i, [_i, _AP, _AP, _P, _P, _i, _i, _P, C.POINTER(InAffine), _i, _P]),
    "segmi_bias_grad": (_i, [_i, _AP, _P, _P, _P]),
    "segmi_bn_stats_rows": (_i, [_AP]),
    "segmi_bn_stats": (_i, [_i, _AP, _P, _P]),
    "segmi_bn_finalize": (_i, [_P, _i, _i, _d, _P, _P, _P, _P, _f, _f, _P, _P, _P, _P, _P]),
    "segmi_bn_eval_affine": (_i, [_i, _P, _P, _P, _P, _f, _P, _P, _P]),
    "segmi_bn_act_fwd": (_i, [_i, _AP, _AP, _P, _P, _P, _AP, _f, C.c_uint32, _P]),
    "segmi_bn_act_bwd_rows": (_i, [_AP]),
    "segmi_bn_act_bwd_reduce": (_i, [_i, _AP, _AP, _P, _P, _P, _P, _P, _P, _f, C.c_uint32,
                                     C.POINTER(BnBwdFin), _P]),
    "segmi_bn_act_bwd_finalize": (_i, [_P, _i, _i, _d, _P, _P, _P, _P, _P, _P, _P]),
    "segmi_bn_act_bwd_fused_ok": (_i, [_i, _AP, _AP, _AP]),
    "segmi_bn_act_bwd_fused_rows": (_i, [_AP]),
    "segmi_bn_act_bwd_fused": (_i, [_i, _AP, _AP, _AP, _P, _P, _P, _P, _P, _P, C.POINTER(BnBwdFin), _i, _P]),
    "segmi_bn_act_bwd_fused_wgs": (_i, [_i, _AP, _i]),
    "segmi_fused_timeouts": (C.c_uint, [_i]),
    "segmi_fused_test_hook": (_i, [C.c_uint, _i]),
    "segmi_bn_act_bwd_apply": (_i, [_i, _AP, _AP, _AP, _P, _P, _P, _P, _P, _P, _f, C.c_uint32, _P]),
    "segmi_bn_act_bwd_apply_conv_ok": (_i, [_i, _AP, _AP, _AP, _AP]),
    "segmi_bn_act_bwd_apply_conv": (_i, [_i, _AP, _AP, _AP, _P, _P, _P, _P, _P, _P, _AP, _P, _P]),
    "segmi_wgrad_cus": (_i, [_i]),
    "segmi_add": (_i, [_i, _AP, _AP, _AP, _P]),
    "segmi_cast_copy": (_i, [_i, _AP, _i, _AP, _P]),
    "segmi_nchw_to_ndhwc": (_i, [_P, _i, _AP, _P]),
    "segmi_ndhwc_to_nchw": (_i, [_i, _AP, _P, _P]),
    "segmi_dice_chunks": (_i, [_AP]),
    "segmi_softmax_dice_fwd": (_i, [_i, _AP, _P, _P, _P, _P, _f, _f, _P]),
    "segmi_softmax_dice_bwd": (_i, [_i, _AP, _P, _P, _f, _AP, _P, _P, _P]),
    "segmi_adam_step": (_i, [_P, _P, _P, _P, _P, _i64, _d, _d, _d, _d, _d, _i64, _f, _P]),
    "segmi_sgd_step": (_i, [_P, _P, _P, _i64, _d, _d, _d, _i, _f, _P]),
    "segmi_adabelief_step": (_i, [_P, _P, _P, _P, _i64, _d, _d, _d, _d, _d, _i, _i64, _f, _P]),
    "segmi_sw_gather": (_i, [_i, _AP, _i, _P, _i, _i, _AP, _P]),
    "segmi_sw_scatter_add": (_i, [_i, _AP, _P, _i, _P, _AP, _P, _P]),
    "segmi_sw_finalize": (_i, [_AP, _P, _i, _P, _i, _P]),
    "segmi_sw_blend": (_i, [_i, _P, _i, _i, _P, _i, _P, _i, _P, _i, _i, _i, _i, _i, _i, _P, _i, _i, _i,
                            _P, _i, _P, _P, _i, _i, _P]),
    "segmi_argmax": (_i, [_i, _AP, _P, _i, _P]),
    "segmi_label_counts": (_i, [_P, _P, _i64, _i, _P, _P]),
    "segmi_resample3d": (_i, [_i, _P, _i, _i, _i, _P, _i, _i, _i, _P, _i, _d, _P]),
    "segmi_normalize_workspace": (_i64, [_i, _i64]),
    "segmi_normalize_intensity": (_i, [_P, _i, _i64, _P, _P]),
    "segmi_crop_patches": (_i, [_AP, _P, _P, _P, _i, _i, _AP, _P, _P]),
    "segmi_warp_crop_patches": (_i, [_AP, _P, _P, _P, _i, _P, _i, _AP, _P, _P]),
    "segmi_intensity_workspace": (_i64, [_i]),
    "segmi_intensity_augment": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _i, _P, _P, _P, _P]),
    "segmi_ensemble_mean": (_i, [_P, _P, _i, _i64, _P, _P]),
    "segmi_ensemble_vote": (_i, [_P, _i, _i64, _P, _P]),
    "segmi_ensemble_select": (_i, [_P, _i, _P, _P, _i, _i64, _P, _P]),
    "segmi_kspace_workspace": (_i64, [_i, _i, _i, _i]),
    "segmi_kspace_augment": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P]),
}


def _load():
    path = Path(os.environ.get("SEGMI_LIB", str(_LIB_PATH)))
    if not path.exists():
        raise ImportError(
            f"segmantic_amd: native library {path} not found. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C segmantic_amd/csrc`. "
            f"There is no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise ImportError(f"segmantic_amd: cannot load {path}: {e}") from e
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError(f"segmantic_amd: {path} lacks symbols {missing}; rebuild it")
    return lib, path


lib, LIB_PATH = _load()


def last_error() -> str:
    return lib.segmi_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libsegmi {what} failed (code {rc}): {last_error()}")
