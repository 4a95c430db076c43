"""Test-case preprocessing of the BraTS nnU-Net plans.

Mirrors ``trainer.preprocess_patient`` as the reference driver calls it
(run_brats2021_inference_singlethread.py:89) for the plans in ``data/temp_inference_output1``:
crop to the nonzero bounding box (mask = OR over modalities, holes filled), identity transpose,
no resampling (1 mm -> 1 mm), per-modality ``nonCT`` z-score with ``use_mask_for_norm``.
Everything after the file read runs on the GPU: nonzero mask, hole filling (border flood fill, bit-exact with
scipy.ndimage.binary_fill_holes), bounding box (``mi355_crop_mask``), masked statistics and normalisation
(``mi355_zscore_masked``).

What the plans ask for is CHECKED, not assumed: a model folder whose plans need something this path does not
implement (a transpose, CT normalisation, resampling to another grid) is refused with a clear error instead of being
segmented on the wrong grid (``check_plans``).
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


def check_spacing(plans: Optional[Dict], spacing_zyx: Optional[Sequence[float]], shape_zyx: Sequence[int]) -> None:
    """nnU-Net v1 resamples whenever ``round(original_spacing / target_spacing * shape) != shape``
    (preprocessing.resample_patient); resampling is not built here, so such an input is refused."""
    if plans is None or spacing_zyx is None:
        return
    target = np.asarray(_stage_plans(plans).get("current_spacing", [1.0, 1.0, 1.0]), dtype=np.float64)
    orig = np.asarray(spacing_zyx, dtype=np.float64)
    shape = np.asarray(shape_zyx, dtype=np.int64)
    new_shape = np.round(orig / target * shape).astype(np.int64)
    if np.any(new_shape != shape):
        raise UnsupportedPlansError(
            f"input spacing (z, y, x) = {tuple(float(v) for v in orig)} differs from the plans' current_spacing "
            f"{tuple(float(v) for v in target)}: nnU-Net would resample {tuple(int(v) for v in shape)} -> "
            f"{tuple(int(v) for v in new_shape)}; resampling is not implemented (BraTS inputs are 1 mm isotropic)")


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
    check_spacing(plans, spacing_zyx, data.shape[1:])
    if all(use_mask):
        ops.zscore_masked_(data, mask)
    else:
        # use_mask_for_norm False: statistics over the whole cropped image (mask of ones), per channel
        ones = torch.ones_like(mask)
        for c, flag in enumerate(use_mask):
            ops.zscore_masked_(data[c:c + 1], mask if flag else ones)
    props = dict(crop_bbox=bbox, original_size_of_raw_data=tuple(int(v) for v in raw.shape[1:]),
                 size_after_cropping=tuple(int(v) for v in data.shape[1:]))
    if spacing_zyx is not None:
        props["original_spacing"] = tuple(float(v) for v in spacing_zyx)
    return data, props
