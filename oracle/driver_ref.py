"""Oracle: the pure-numpy parts of the reference driver, restated with the reference's own
expressions (run_brats2021_inference_singlethread.py = "driver").  Test infrastructure only."""
from __future__ import annotations

import numpy as np

from . import tiler_ref, unet_ref


def fold_mean(all_softmax):
    """driver :128  softmax_mean = np.mean(all_softmax, axis=0)"""
    return np.mean(all_softmax, axis=0)


def label_ensemble(seg1, seg2):
    """driver :299-305 on get_fdata() float64 label maps."""
    return np.round((seg1.astype(np.float64) + seg2.astype(np.float64)) / 2.0).astype(np.uint8)


def calculate_volumes(seg, zooms):
    """driver :217-243 (labels 1, 2, 4)."""
    vv = float(np.prod(zooms)) / 1000.0
    ncr, ed, et = np.sum(seg == 1), np.sum(seg == 2), np.sum(seg == 4)
    return {"NCR": ncr * vv, "ED": ed * vv, "ET": et * vv, "TC": (ncr + et) * vv, "WT": (ncr + ed + et) * vv}


def predict_case(raw_zyx, fold_state_dicts, cfg, patch, do_tta=True, step_size=0.5, nonlin="sigmoid"):
    """driver :81-158 for one model: preprocess, per-fold tiled prediction, fold mean, region export
    pasted into the raw-size volume.  Returns (labels [Z,Y,X] uint8, mean probabilities, props)."""
    data, props = tiler_ref.preprocess_case(raw_zyx)
    per_fold = [tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, cfg), data, patch, 3, step_size, do_tta,
                                           (0, 1, 2), True, nonlin) for sd in fold_state_dicts]
    probs = fold_mean(per_fold)
    seg = tiler_ref.regions_to_labels(probs, (1, 2, 3))
    full = tiler_ref.paste_into_original(seg, props["crop_bbox"], props["original_size_of_raw_data"])
    return full, probs, props


def apply_brats_threshold(seg, threshold=200, replace_with=2):
    """archived/kaist_original_inference.py:33 -> nnunet.dataset_conversion.Task500_BraTS_2021.apply_threshold_to_folder
    (nnU-Net v1, KAIST BraTS21 fork; un-vendored, see oracle/__init__.py: parity unpinned).  Published algorithm,
    per file: ``s = np.sum(img == 3); if s < threshold: img[img == 3] = replace_with`` - a case with fewer than
    `threshold` enhancing-tumour voxels has them relabelled (to label 2 at the KAIST call site)."""
    out = np.array(seg, copy=True)
    if int(np.sum(out == 3)) < threshold:
        out[out == 3] = replace_with
    return out


def convert_labels_back_to_brats(seg):
    """archived/kaist_original_inference.py:34 -> Task032_BraTS_2018.convert_labels_back_to_BraTS_2018_2019_convention
    (un-vendored): ``new[seg == 1] = 2; new[seg == 3] = 4; new[seg == 2] = 1`` - identical to the in-repo
    convert_labels_to_brats.py:46-55 (BraTS 2021 convention), which pins it."""
    seg = np.round(seg).astype(np.uint8)
    new = np.zeros_like(seg)
    new[seg == 1] = 2
    new[seg == 2] = 1
    new[seg == 3] = 4
    return new
