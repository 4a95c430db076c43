"""ctypes binding of include/mi355_nnunet.h.

There is no Python/CPU fallback: if the shared library is missing the import of any compute
entry point raises, and on a machine without a gfx950 device the library itself returns
MI355_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _build

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)

NORM_NONE, NORM_BATCH, NORM_INSTANCE, NORM_GROUP = 0, 1, 2, 3
F32, F16 = 0, 1
NONLIN_IDENTITY, NONLIN_SIGMOID, NONLIN_SOFTMAX = 0, 1, 2

#: every symbol include/mi355_nnunet.h declares (tests check that the .so exports all of them)
EXPORTS = [
    "mi355_last_error", "mi355_version", "mi355_device_count", "mi355_unet_create", "mi355_unet_destroy",
    "mi355_unet_flops", "mi355_unet_forward", "mi355_sw_predict", "mi355_compute_steps", "mi355_sw_partial",
    "mi355_sw_finish", "mi355_regions_to_labels", "mi355_label_ensemble", "mi355_prob_mean",
    "mi355_zscore_masked", "mi355_conv3d_ndhwc", "mi355_tconv3d_ndhwc", "mi355_profile_enable",
    "mi355_profile_read", "mi355_conv3d_ndhwc_f16", "mi355_tconv3d_ndhwc_f16",
    "mi355_label_remap", "mi355_label_confusion", "mi355_cosine_topk", "mi355_crop_mask", "mi355_label_stats",
    "mi355_last_conv_kernel",
    "mi355_conv3d_sums_ndhwc",
    "mi355_sw_partial_folds", "mi355_sw_finish_folds",
    "mi355_resize_axis", "mi355_clip_to_range_of", "mi355_threshold_ge", "mi355_mask_to_float",
]


class ConvDesc(C.Structure):
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("stride", C.c_int32),
                ("weight", c_float_p), ("bias", c_float_p), ("gamma", c_float_p), ("beta", c_float_p),
                ("running_mean", c_float_p), ("running_var", c_float_p)]


class TConvDesc(C.Structure):
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("weight", c_float_p)]


class HeadDesc(C.Structure):
    _fields_ = [("cin", C.c_int32), ("num_classes", C.c_int32), ("weight", c_float_p), ("bias", c_float_p)]


class UNetDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("num_classes", C.c_int32), ("num_pool", C.c_int32),
                ("norm", C.c_int32), ("num_groups", C.c_int32), ("eps", C.c_float), ("lrelu_slope", C.c_float),
                ("nonlin_first", C.c_int32), ("dtype", C.c_int32),
                ("enc_convs", c_int32_p), ("dec_convs", c_int32_p),
                ("convs", C.POINTER(ConvDesc)), ("n_convs", C.c_int32),
                ("tconvs", C.POINTER(TConvDesc)), ("head", HeadDesc)]


class SwOpts(C.Structure):
    _fields_ = [("patch", C.c_int32 * 3), ("step_size", C.c_float), ("use_gaussian", C.c_int32),
                ("mirror_axes", C.c_int32), ("nonlin", C.c_int32), ("batch_tiles", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class Mi355Error(RuntimeError):
    pass


_lib = None


def lib_path() -> Path:
    return _build.LIB_PATH


def load():
    """Load (once) the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise Mi355Error(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if _build.needs_build():
        # the digest of csrc/ differs from the one the library was built from: rebuild (under the build lock) rather than run stale code
        try:
            _build.build()
        except Exception as e:  # no hipcc on this machine: say so instead of silently running the old library
            import warnings
            warnings.warn(f"{path} was built from a different csrc/ and could not be rebuilt ({e}); running the stale library")
    # torch first: its wheel bundles a HIP runtime of its own (torch/lib/libamdhip64.so), and the library must share THAT runtime -
    # it is handed torch's device pointers and streams.  Loaded before torch, this library pulls in /opt/rocm's libamdhip64 under
    # the same SONAME and the process ends up with one runtime torch was not built against (seen in round 5: build() + smoke() in
    # one process - mi355_device_count() = 0 on a box with a GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(path))
    vp = C.c_void_p
    lib.mi355_last_error.restype = C.c_char_p
    lib.mi355_last_conv_kernel.restype = C.c_char_p
    lib.mi355_unet_create.argtypes = [C.POINTER(UNetDesc), C.POINTER(vp)]
    lib.mi355_unet_destroy.argtypes = [vp]
    lib.mi355_unet_flops.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.mi355_unet_flops.restype = C.c_int64
    lib.mi355_unet_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.mi355_sw_predict.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(SwOpts), vp, vp]
    lib.mi355_compute_steps.argtypes = [C.c_int, C.c_int, C.c_float, c_int32_p, C.c_int]
    lib.mi355_sw_partial.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(SwOpts), C.c_int, C.c_int, vp, vp, vp]
    lib.mi355_sw_finish.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, c_int32_p, vp, vp]
    lib.mi355_sw_partial_folds.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(SwOpts), C.c_int, C.c_int, vp, vp, vp]
    lib.mi355_sw_finish_folds.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int, vp, vp]
    lib.mi355_regions_to_labels.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_int32_p, c_int32_p, c_int32_p, vp, vp]
    lib.mi355_label_ensemble.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.mi355_prob_mean.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.mi355_zscore_masked.argtypes = [vp, vp, C.c_int, C.c_int64, vp]
    lib.mi355_conv3d_ndhwc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int,
                                       C.c_int, C.c_int, C.c_float, C.c_int, vp, vp]
    lib.mi355_tconv3d_ndhwc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, vp, vp]
    lib.mi355_conv3d_ndhwc_f16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int,
                                           C.c_int, C.c_int, C.c_float, vp, vp]
    lib.mi355_tconv3d_ndhwc_f16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, vp, vp]
    lib.mi355_conv3d_sums_ndhwc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int,
                                            C.c_int, C.c_int, C.c_float, vp, vp, vp]
    lib.mi355_label_remap.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_uint8), vp]
    lib.mi355_label_confusion.argtypes = [vp, vp, C.c_int64, C.c_int, C.POINTER(C.c_uint64), vp]
    lib.mi355_cosine_topk.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, c_int32_p, c_float_p, vp]
    lib.mi355_crop_mask.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, c_int32_p, vp]
    lib.mi355_label_stats.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), vp]
    lib.mi355_resize_axis.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, vp]
    lib.mi355_clip_to_range_of.argtypes = [vp, C.c_int64, C.c_int64, vp, C.c_int64, vp]
    lib.mi355_threshold_ge.argtypes = [vp, C.c_float, vp, C.c_int64, vp]
    lib.mi355_mask_to_float.argtypes = [vp, vp, C.c_int64, vp]
    lib.mi355_profile_enable.argtypes = [vp, C.c_int]
    lib.mi355_profile_read.argtypes = [vp, C.POINTER(ProfEntry), C.c_int]
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = load().mi355_last_error()
        raise Mi355Error(f"{what or 'mi355 call'} failed ({rc}): {msg.decode() if msg else '?'}")


def fptr(a):
    """float32 C-contiguous numpy array -> float* (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(c_float_p)
