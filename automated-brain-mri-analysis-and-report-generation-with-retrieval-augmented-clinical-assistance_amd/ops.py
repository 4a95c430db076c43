"""Thin torch-CUDA wrappers over the single-purpose C-ABI entry points.

torch is used for device memory and streams only; every computation happens in the HIP
library.  All functions raise ``Mi355Error`` if the library is missing or no GPU is present.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_cuda(t, dtype, name):
    if not t.is_cuda or t.dtype != dtype:
        raise ValueError(f"{name}: expected a CUDA tensor of dtype {dtype}")
    return t.contiguous()


def regions_to_labels(probs, order=(1, 2, 3), bbox_lo=(0, 0, 0), full_shape=None):
    """seg = 0; seg[probs[i] > 0.5] = order[i] in order; pasted at bbox_lo of a zero uint8 volume of
    full_shape (reference driver :144-156, region_class_order=(1,2,3)).  ``order=None``: seg = argmax over channels
    (the same export for trainers without regions)."""
    import torch
    probs = _require_cuda(probs, torch.float32, "probs")
    c, z, y, x = probs.shape
    full = tuple(full_shape) if full_shape is not None else (z, y, x)
    out = torch.empty(full, dtype=torch.uint8, device=probs.device)
    order_a = None if order is None else (C.c_int32 * len(order))(*[int(v) for v in order])
    lo = (C.c_int32 * 3)(*[int(v) for v in bbox_lo])
    fu = (C.c_int32 * 3)(*[int(v) for v in full])
    _lib.check(_lib.load().mi355_regions_to_labels(probs.data_ptr(), c, z, y, x, order_a, lo, fu, out.data_ptr(),
                                                   _stream(probs)), "mi355_regions_to_labels")
    return out


def label_ensemble(a, b):
    """uint8(np.round((a + b) / 2.0)) - the reference's 2-model label ensemble (driver :305)."""
    import torch
    a = _require_cuda(a, torch.uint8, "a")
    b = _require_cuda(b, torch.uint8, "b")
    if a.shape != b.shape:
        raise ValueError("label_ensemble: shape mismatch")
    out = torch.empty_like(a)
    _lib.check(_lib.load().mi355_label_ensemble(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a)),
               "mi355_label_ensemble")
    return out


def prob_mean(a, b):
    import torch
    a = _require_cuda(a, torch.float32, "a")
    b = _require_cuda(b, torch.float32, "b")
    if a.shape != b.shape:
        raise ValueError("prob_mean: shape mismatch")
    out = torch.empty_like(a)
    _lib.check(_lib.load().mi355_prob_mean(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a)),
               "mi355_prob_mean")
    return out


def crop_mask(vol):
    """Nonzero mask with holes filled and its bounding box, on the device (nnU-Net v1 ``crop_to_nonzero`` as the
    reference's ``trainer.preprocess_patient`` runs it, driver :89).  vol: CUDA fp32 [C, Z, Y, X].
    Returns (mask uint8 [Z, Y, X], bbox [[z_lo, z_hi], [y_lo, y_hi], [x_lo, x_hi]])."""
    import torch
    _require_cuda(vol, torch.float32, "vol")
    if vol.dim() != 4:
        raise ValueError("crop_mask expects [C, Z, Y, X]")
    vol = vol.contiguous()
    c, z, y, x = vol.shape
    mask = torch.empty((z, y, x), dtype=torch.uint8, device=vol.device)
    box = (C.c_int32 * 6)()
    _lib.check(_lib.load().mi355_crop_mask(vol.data_ptr(), c, z, y, x, mask.data_ptr(), box, _stream(vol)), "mi355_crop_mask")
    return mask, [[int(box[0]), int(box[1])], [int(box[2]), int(box[3])], [int(box[4]), int(box[5])]]


def zscore_masked_(vol, mask):
    """In place: per channel x[m] = (x[m]-mean)/(std+1e-8), x[~m] = 0 (nonCT + use_mask_for_norm)."""
    import torch
    if not (vol.is_cuda and vol.dtype == torch.float32 and vol.is_contiguous()):
        raise ValueError("zscore_masked_: vol must be a contiguous CUDA fp32 tensor")
    mask = _require_cuda(mask, torch.uint8, "mask")
    c = vol.shape[0]
    v = vol[0].numel()
    if mask.numel() != v:
        raise ValueError("zscore_masked_: mask shape mismatch")
    _lib.check(_lib.load().mi355_zscore_masked(vol.data_ptr(), mask.data_ptr(), c, v, _stream(vol)),
               "mi355_zscore_masked")
    return vol


def resize_axis(x, axis: int, n_out: int, order: int):
    """One 1-D resampling pass along ``axis`` of a contiguous CUDA fp32 tensor (``mi355_resize_axis``): half-pixel-centred grid,
    edge replication, interpolation order 0 / 1 / 3 - skimage.transform.resize(order, mode='edge', anti_aliasing=False) along one
    axis, before its clipping."""
    import torch
    x = _require_cuda(x, torch.float32, "x")
    axis = axis % x.dim()
    n_in = int(x.shape[axis])
    if int(n_out) == n_in:
        return x
    outer = int(np.prod(x.shape[:axis], dtype=np.int64)) if axis > 0 else 1
    inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64)) if axis + 1 < x.dim() else 1
    out = torch.empty(tuple(x.shape[:axis]) + (int(n_out),) + tuple(x.shape[axis + 1:]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi355_resize_axis(x.data_ptr(), out.data_ptr(), outer, n_in, int(n_out), inner, int(order), _stream(x)),
               "mi355_resize_axis")
    return out


def clip_to_range_of_(x, ref, group_dims: int):
    """In place: every group of ``x`` (its first ``group_dims`` axes index the groups) clipped to [min, max] of the same group of
    ``ref`` (skimage's resize clips its output to the range of its input)."""
    import torch
    x = _require_cuda(x, torch.float32, "x")
    ref = _require_cuda(ref, torch.float32, "ref")
    if tuple(x.shape[:group_dims]) != tuple(ref.shape[:group_dims]):
        raise ValueError("clip_to_range_of_: group shapes differ")
    groups = int(np.prod(x.shape[:group_dims], dtype=np.int64)) if group_dims else 1
    _lib.check(_lib.load().mi355_clip_to_range_of(x.data_ptr(), groups, x.numel() // groups, ref.data_ptr(), ref.numel() // groups,
                                                  _stream(x)), "mi355_clip_to_range_of")
    return x


def mask_to_float(mask):
    """fp32 indicator (1.0 / 0.0) of a uint8 mask (``mi355_mask_to_float``)."""
    import torch
    mask = _require_cuda(mask, torch.uint8, "mask")
    out = torch.empty(mask.shape, dtype=torch.float32, device=mask.device)
    _lib.check(_lib.load().mi355_mask_to_float(mask.data_ptr(), out.data_ptr(), mask.numel(), _stream(mask)), "mi355_mask_to_float")
    return out


def threshold_ge(x, thr: float):
    """uint8 mask ``x >= thr`` (``mi355_threshold_ge``)."""
    import torch
    x = _require_cuda(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().mi355_threshold_ge(x.data_ptr(), float(thr), out.data_ptr(), x.numel(), _stream(x)), "mi355_threshold_ge")
    return out


def conv3d_ndhwc(x, weight, bias=None, stride=1, act=0, slope=0.01, impl="mfma"):
    """Single conv (test entry point). x: CUDA fp32 or fp16 [N,D,H,W,Cin]; weight: numpy [Cout,Cin,3,3,3]."""
    import torch
    if x.dtype == torch.float16:
        x = _require_cuda(x, torch.float16, "x")
        n, d, h, w, cin = x.shape
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        cout = weight.shape[0]
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
        y = torch.empty((n, do, ho, wo, cout), dtype=torch.float16, device=x.device)
        _lib.check(_lib.load().mi355_conv3d_ndhwc_f16(x.data_ptr(), n, d, h, w, cin, _lib.fptr(weight), _lib.fptr(b), cout,
                                                      stride, act, slope, y.data_ptr(), _stream(x)), "mi355_conv3d_ndhwc_f16")
        return y
    x = _require_cuda(x, torch.float32, "x")
    n, d, h, w, cin = x.shape
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    cout = weight.shape[0]
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, do, ho, wo, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi355_conv3d_ndhwc(x.data_ptr(), n, d, h, w, cin, _lib.fptr(weight), _lib.fptr(b), cout,
                                              stride, act, slope, {"mfma": 0, "direct": 1}[impl], y.data_ptr(),
                                              _stream(x)), "mi355_conv3d_ndhwc")
    return y


def conv3d_sums_ndhwc(x, weight, bias=None, stride=1, act=0, slope=0.01):
    """Single conv with the run-time-norm statistics epilogue (test entry point): returns (y, sums) with
    sums[n, cout] = (sum of y, sum of y^2) over the voxels, fp64 - what InstanceNorm / GroupNorm reduce y to
    (generic_UNet.py:62-72).  x: CUDA fp32 or fp16 [N,D,H,W,Cin]; weight: numpy [Cout,Cin,3,3,3]."""
    import torch
    f16 = x.dtype == torch.float16
    x = _require_cuda(x, torch.float16 if f16 else torch.float32, "x")
    n, d, h, w, cin = x.shape
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    cout = weight.shape[0]
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, do, ho, wo, cout), dtype=x.dtype, device=x.device)
    sums = torch.empty((n, cout, 2), dtype=torch.float64, device=x.device)
    _lib.check(_lib.load().mi355_conv3d_sums_ndhwc(x.data_ptr(), 1 if f16 else 0, n, d, h, w, cin, _lib.fptr(weight), _lib.fptr(b),
                                                   cout, stride, act, slope, y.data_ptr(), sums.data_ptr(), _stream(x)),
               "mi355_conv3d_sums_ndhwc")
    return y, sums


def last_conv_kernel() -> str:
    """Kernel instantiation the last ``conv3d_ndhwc`` call of this thread ran on (test aid)."""
    return (_lib.load().mi355_last_conv_kernel() or b"").decode()


def tconv3d_ndhwc(x, weight):
    """Single ConvTranspose3d k=2 s=2 (test entry point). weight: numpy [Cin,Cout,2,2,2]."""
    import torch
    if x.dtype == torch.float16:
        x = _require_cuda(x, torch.float16, "x")
        n, d, h, w, cin = x.shape
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        cout = weight.shape[1]
        y = torch.empty((n, 2 * d, 2 * h, 2 * w, cout), dtype=torch.float16, device=x.device)
        _lib.check(_lib.load().mi355_tconv3d_ndhwc_f16(x.data_ptr(), n, d, h, w, cin, _lib.fptr(weight), cout, y.data_ptr(),
                                                       _stream(x)), "mi355_tconv3d_ndhwc_f16")
        return y
    x = _require_cuda(x, torch.float32, "x")
    n, d, h, w, cin = x.shape
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    cout = weight.shape[1]
    y = torch.empty((n, 2 * d, 2 * h, 2 * w, cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi355_tconv3d_ndhwc(x.data_ptr(), n, d, h, w, cin, _lib.fptr(weight), cout, y.data_ptr(),
                                               _stream(x)), "mi355_tconv3d_ndhwc")
    return y


def compute_steps(patch, image, step_size):
    buf = (C.c_int32 * 256)()
    n = _lib.load().mi355_compute_steps(int(patch), int(image), float(step_size), buf, 256)
    _lib.check(n, "mi355_compute_steps")
    return [int(buf[i]) for i in range(n)]
