"""Test-case preprocessing of the BraTS nnU-Net plans.

Mirrors ``trainer.preprocess_patient`` as the reference driver calls it
(run_brats2021_inference_singlethread.py:89) for the plans in ``data/temp_inference_output1``:
crop to the nonzero bounding box (mask = OR over modalities, holes filled), identity transpose,
no resampling (1 mm -> 1 mm), per-modality ``nonCT`` z-score with ``use_mask_for_norm=True``.
Everything after the file read runs on the GPU: nonzero mask, hole filling (border flood fill, bit-exact with
scipy.ndimage.binary_fill_holes), bounding box (``mi355_crop_mask``), masked statistics and normalisation
(``mi355_zscore_masked``).  ``crop_to_nonzero`` below is the host version of the same crop (kept for callers without
a device tensor).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def crop_to_nonzero(raw: np.ndarray):
    """-> (cropped [C,..], inside-mask (bool, crop shape), bbox [[lo, hi], ...])."""
    from scipy.ndimage import binary_fill_holes
    nonzero = np.zeros(raw.shape[1:], dtype=bool)
    for c in range(raw.shape[0]):
        nonzero |= raw[c] != 0
    if not nonzero.any():
        raise ValueError("volume is all zeros: nothing to segment")
    nonzero = binary_fill_holes(nonzero)
    bbox = []
    for ax in range(nonzero.ndim):
        proj = np.flatnonzero(nonzero.any(axis=tuple(a for a in range(nonzero.ndim) if a != ax)))
        bbox.append([int(proj[0]), int(proj[-1]) + 1])
    sl = tuple(slice(lo, hi) for lo, hi in bbox)
    return np.ascontiguousarray(raw[(slice(None),) + sl]), np.ascontiguousarray(nonzero[sl]), bbox


def preprocess_case(raw: np.ndarray, device="cuda") -> Tuple["object", Dict]:
    """raw [4,Z,Y,X] (any real dtype) -> (CUDA fp32 [4,Zc,Yc,Xc] normalised, properties dict with
    crop_bbox / original_size_of_raw_data / size_after_cropping, the fields export needs)."""
    import torch
    from . import ops
    raw = np.asarray(raw, dtype=np.float32)
    vol = torch.from_numpy(np.ascontiguousarray(raw)).to(device)
    full_mask, bbox = ops.crop_mask(vol)
    sl = tuple(slice(lo, hi) for lo, hi in bbox)
    data = vol[(slice(None),) + sl].contiguous()
    mask = full_mask[sl].contiguous()
    del vol, full_mask
    ops.zscore_masked_(data, mask)
    props = dict(crop_bbox=bbox, original_size_of_raw_data=tuple(int(v) for v in raw.shape[1:]),
                 size_after_cropping=tuple(int(v) for v in data.shape[1:]))
    return data, props
