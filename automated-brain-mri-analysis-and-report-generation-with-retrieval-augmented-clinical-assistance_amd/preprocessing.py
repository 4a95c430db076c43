"""Test-case preprocessing of the BraTS nnU-Net plans.

Mirrors ``trainer.preprocess_patient`` as the reference driver calls it
(run_brats2021_inference_singlethread.py:89) for the plans in ``data/temp_inference_output1``:
crop to the nonzero bounding box (mask = OR over modalities, holes filled), identity transpose,
resampling to the plans' spacing (a no-op for BraTS: 1 mm -> 1 mm; otherwise data with cubic splines, the inside mask
linearly, a low-resolution axis separately - round 4, ``resample_to_spacing``), per-modality ``nonCT`` z-score with
``use_mask_for_norm``.
Everything after the file read runs on the GPU: nonzero mask, hole filling (border flood fill, bit-exact with
scipy.ndimage.binary_fill_holes), bounding box (``mi355_crop_mask``), masked statistics and normalisation
(``mi355_zscore_masked``).

What the plans ask for is CHECKED, not assumed: a model folder whose plans need something this path does not
implement (a transpose, CT normalisation) is refused with a clear error instead of being segmented wrongly
(``check_plans``).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np


class UnsupportedPlansError(ValueError):
    """The model's plans (or the input geometry under those plans) need a preprocessing step that is not built."""


def _stage_plans(plans: Dict) -> Dict:
    st = plans["plans_per_stage"]
    return st[max(st.keys())]  # the trainer predicts with the last (full-resolution) stage


def check_plans(plans: Optional[Dict], num_channels: int) -> Sequence[bool]:
    """Validates the plan fields ``GenericPreprocessor.preprocess_test_case`` acts on and returns the per-channel
    ``use_mask_for_norm`` flags.  ``plans=None`` means the BraTS plans of ``data/temp_inference_output1``."""
    if plans is None:
        return [True] * num_channels
    tf = [int(v) for v in plans.get("transpose_forward", [0, 1, 2])]
    if tf != [0, 1, 2]:
        raise UnsupportedPlansError(f"plans ask for transpose_forward={tf}; only the identity [0, 1, 2] (BraTS plans) is implemented")
    schemes = plans.get("normalization_schemes") or {}
    flags = plans.get("use_mask_for_norm") or {}
    out = []
    for c in range(num_channels):
        scheme = schemes.get(c, "nonCT")
        if scheme != "nonCT":
            raise UnsupportedPlansError(f"plans ask for normalization scheme {scheme!r} on channel {c}; only 'nonCT' "
                                        "(per-case z-score) is implemented")
        out.append(bool(flags.get(c, True)))
    nmod = plans.get("num_modalities")
    if nmod is not None and int(nmod) != num_channels:
        raise UnsupportedPlansError(f"plans expect {nmod} modalities, the case has {num_channels}")
    return out


RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3.0  # nnunet.configuration


def get_do_separate_z(spacing, anisotropy_threshold: float = RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD) -> bool:
    return bool((np.max(spacing) / np.min(spacing)) > anisotropy_threshold)


def get_lowres_axis(spacing):
    return np.where(max(spacing) / np.asarray(spacing, dtype=np.float64) == 1)[0]


def resample_plan(shape_zyx: Sequence[int], original_spacing: Sequence[float], target_spacing: Sequence[float],
                  force_separate_z=None) -> Tuple[Tuple[int, int, int], bool, Optional[int]]:
    """The decisions of nnU-Net v1 ``resample_patient`` (called by ``GenericPreprocessor.resample_and_normalize`` behind driver
    :89): the new shape ``round(original / target * shape)``, and whether the low-resolution axis is resampled separately
    (anisotropy beyond 3 in the original or, failing that, the target spacing; not when two or three axes share the largest
    spacing).  Returns ``(new_shape, do_separate_z, axis)``."""
    orig = np.asarray(original_spacing, dtype=np.float64)
    target = np.asarray(target_spacing, dtype=np.float64)
    new_shape = np.round((orig / target).astype(float) * np.asarray(shape_zyx)).astype(int)
    if force_separate_z is not None:
        sep, axis = bool(force_separate_z), (get_lowres_axis(orig) if force_separate_z else None)
    elif get_do_separate_z(orig):
        sep, axis = True, get_lowres_axis(orig)
    elif get_do_separate_z(target):
        sep, axis = True, get_lowres_axis(target)
    else:
        sep, axis = False, None
    if axis is not None and len(axis) != 1:
        sep = False
    return tuple(int(v) for v in new_shape), sep, (int(axis[0]) if (sep and axis is not None) else None)


def resample_data_or_seg(x, new_shape: Sequence[int], order: int = 3, do_separate_z: bool = False, axis: Optional[int] = None,
                         order_z: int = 0, is_mask: bool = False):
    """nnU-Net v1 ``resample_data_or_seg`` on the device.  ``x``: CUDA fp32 ``[C, Z, Y, X]``.  A tensor-product spline resize is one
    1-D pass per axis (``mi355_resize_axis``: half-pixel-centred grid, edge replication = skimage's ``resize(order, mode='edge',
    anti_aliasing=False)``), clipped to the range of its input (``mi355_clip_to_range_of``): per channel for the 3-D resize, per
    slice in the separate-z mode, where the low-resolution axis is then resampled with ``order_z`` (0 = nearest).  ``is_mask``: the
    0/1 inside mask goes through batchgenerators' ``resize_segmentation`` rule - linear resize of the indicator, kept where >= 0.5 -
    and comes back as uint8."""
    import torch
    from . import ops
    x = x.contiguous()
    c = x.shape[0]
    shape = tuple(int(v) for v in x.shape[1:])
    new_shape = tuple(int(v) for v in new_shape)
    if shape == new_shape:
        return ops.threshold_ge(x, 0.5) if is_mask else x
    if do_separate_z:
        if axis is None:
            raise ValueError("separate-z resampling needs the low-resolution axis")
        if axis != 0:  # bring the low-resolution axis to the front of the spatial axes (memory movement only)
            perm = [0, 1 + axis] + [1 + a for a in range(3) if a != axis]
            inv = [perm.index(i) for i in range(4)]
            y = resample_data_or_seg(x.permute(perm).contiguous(), (new_shape[axis],) + tuple(v for i, v in enumerate(new_shape) if i != axis),
                                     order, True, 0, order_z, is_mask)
            return y.permute(inv).contiguous()
        y = ops.resize_axis(x, 3, new_shape[2], order)       # in-plane, slice by slice (the passes never mix slices)
        y = ops.resize_axis(y, 2, new_shape[1], order)
        if order > 1:
            ops.clip_to_range_of_(y, x, 2)                   # skimage clips each 2-D slice to ITS input's range
        # (is_mask: the reference thresholds each slice before the z step; a nearest-neighbour pick of slices commutes with that)
        if shape[0] != new_shape[0]:
            y = ops.resize_axis(y, 1, new_shape[0], order_z)
        return ops.threshold_ge(y, 0.5) if is_mask else y
    y = ops.resize_axis(x, 3, new_shape[2], order)
    y = ops.resize_axis(y, 2, new_shape[1], order)
    y = ops.resize_axis(y, 1, new_shape[0], order)
    if order > 1:
        ops.clip_to_range_of_(y, x, 1)                       # per channel: the reference resizes data[c] one 3-D image at a time
    return ops.threshold_ge(y, 0.5) if is_mask else y


def check_spacing(plans: Optional[Dict], spacing_zyx: Optional[Sequence[float]], shape_zyx: Sequence[int]):
    """The resampling the plans ask for at this input geometry: ``None`` when the grid already is the target grid (every BraTS
    case: 1 mm isotropic against ``current_spacing`` 1 mm), else ``(new_shape, do_separate_z, axis, target_spacing)``."""
    if plans is None or spacing_zyx is None:
        return None
    target = np.asarray(_stage_plans(plans).get("current_spacing", [1.0, 1.0, 1.0]), dtype=np.float64)
    new_shape, sep, axis = resample_plan(shape_zyx, spacing_zyx, target)
    if tuple(new_shape) == tuple(int(v) for v in shape_zyx):
        return None
    return new_shape, sep, axis, tuple(float(v) for v in target)


def nonzero_crop_shape(raw: np.ndarray) -> Tuple[int, int, int]:
    """Shape of the nonzero crop of ``raw`` [C, Z, Y, X], on the host, without the device pass: the bounding box of
    ``crop_to_nonzero`` is that of the plain nonzero mask (filled holes are interior, they never move a face).  Used to
    cost a case (``parallel.tiles_of_shape``) before deciding which rank preprocesses it."""
    nz = np.any(np.asarray(raw) != 0, axis=0)
    if not nz.any():
        return tuple(int(v) for v in nz.shape)
    out = []
    for ax in range(3):
        proj = np.any(nz, axis=tuple(a for a in range(3) if a != ax))
        idx = np.flatnonzero(proj)
        out.append(int(idx[-1] - idx[0] + 1))
    return tuple(out)


def preprocess_case(raw: np.ndarray, device="cuda", plans: Optional[Dict] = None,
                    spacing_zyx: Optional[Sequence[float]] = None) -> Tuple["object", Dict]:
    """raw [C,Z,Y,X] (any real dtype) -> (CUDA fp32 [C,Zc,Yc,Xc] normalised, properties dict with
    crop_bbox / original_size_of_raw_data / size_after_cropping, the fields export needs)."""
    import torch
    from . import ops
    raw = np.asarray(raw, dtype=np.float32)
    use_mask = check_plans(plans, raw.shape[0])
    vol = torch.from_numpy(np.ascontiguousarray(raw)).to(device)
    full_mask, bbox = ops.crop_mask(vol)
    sl = tuple(slice(lo, hi) for lo, hi in bbox)
    data = vol[(slice(None),) + sl].contiguous()
    mask = full_mask[sl].contiguous()
    del vol, full_mask
    size_after_cropping = tuple(int(v) for v in data.shape[1:])
    rs = check_spacing(plans, spacing_zyx, data.shape[1:])
    if rs is not None:  # step 4 of preprocess_patient: data with cubic splines, the inside mask linearly (nnU-Net: order 3 / order 1)
        new_shape, sep, axis, target = rs
        data = resample_data_or_seg(data, new_shape, 3, sep, axis, 0)
        mask = resample_data_or_seg(ops.mask_to_float(mask)[None], new_shape, 1, sep, axis, 0, is_mask=True)[0].contiguous()
    if all(use_mask):
        ops.zscore_masked_(data, mask)
    else:
        # use_mask_for_norm False: statistics over the whole cropped image (mask of ones), per channel
        ones = torch.ones_like(mask)
        for c, flag in enumerate(use_mask):
            ops.zscore_masked_(data[c:c + 1], mask if flag else ones)
    props = dict(crop_bbox=bbox, original_size_of_raw_data=tuple(int(v) for v in raw.shape[1:]),
                 size_after_cropping=size_after_cropping)
    if rs is not None:
        props["size_after_resampling"] = tuple(int(v) for v in data.shape[1:])
        props["spacing_after_resampling"] = rs[3]
    if spacing_zyx is not None:
        props["original_spacing"] = tuple(float(v) for v in spacing_zyx)
    return data, props


def resample_probabilities_for_export(probs, props: Dict, order: int = 1, order_z: int = 0):
    """The resampling inside ``save_segmentation_nifti_from_softmax(..., order=1, force_separate_z=None, interpolation_order_z=0)``
    (driver :131-138, :144-156): class probabilities on the resampled grid back to the shape after cropping - linear, the
    low-resolution axis separately (nearest) when the original or the resampled spacing is anisotropic beyond 3.  A no-op when
    preprocessing did not resample (every BraTS case)."""
    target = tuple(int(v) for v in props["size_after_cropping"])
    if tuple(int(v) for v in probs.shape[1:]) == target:
        return probs
    orig, after = props.get("original_spacing"), props.get("spacing_after_resampling")
    if orig is not None and get_do_separate_z(orig):
        sep, axis = True, get_lowres_axis(orig)
    elif after is not None and get_do_separate_z(after):
        sep, axis = True, get_lowres_axis(after)
    else:
        sep, axis = False, None
    if axis is not None and len(axis) != 1:
        sep = False
    return resample_data_or_seg(probs, target, order, sep, int(axis[0]) if sep else None, order_z)
