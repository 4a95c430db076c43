"""GPU mirror of the reference evaluator and label converter.

``evaluate_segmentation.py`` builds float masks and sums them four times per label; here one
pass over the two uint8 label maps on the device yields the K x K confusion counts, and every
metric of ``calculate_metrics`` (:12-49) / ``calculate_metrics_binary`` (:181-195) and the compound
WT / TC / ET regions (:129-162) is derived from those integers with the reference's formulas
(same 1e-8 epsilons).  ``convert_labels`` is convert_labels_to_brats.py:34-55 as a 256-entry map.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

LABEL_MAPS = {
    "nnunet": {},
    "brats2025": {1: 2, 2: 1, 3: 3},   # convert_labels_to_brats.py:34-43
    "brats2021": {1: 2, 2: 1, 3: 4},   # :46-55
}


def convert_labels(seg, fmt="brats2025"):
    """seg: CUDA uint8 tensor of nnU-Net labels -> same shape in the requested convention
    (labels the reference's converter does not mention become 0, as ``np.zeros_like`` leaves them)."""
    import torch
    if fmt not in LABEL_MAPS:
        raise ValueError(f"unknown label format {fmt}")
    if fmt == "nnunet":
        return seg.clone()
    table = np.zeros(256, dtype=np.uint8)
    for k, v in LABEL_MAPS[fmt].items():
        table[k] = v
    seg = seg.contiguous()
    out = torch.empty_like(seg)
    stream = torch.cuda.current_stream(seg.device).cuda_stream
    _lib.check(_lib.load().mi355_label_remap(seg.data_ptr(), out.data_ptr(), seg.numel(),
                                             table.ctypes.data_as(C.POINTER(C.c_uint8)), stream), "mi355_label_remap")
    return out


def apply_brats_threshold(seg, threshold=200, replace_with=2):
    """KAIST post-processing (archived/kaist_original_inference.py:33, ``apply_threshold_to_folder(..., 200, 2)``):
    a case with fewer than `threshold` voxels of label 3 (enhancing tumour) gets them relabelled to `replace_with`.
    seg: CUDA uint8 label map.  Returns (new label map, number of label-3 voxels found)."""
    n3 = int(confusion(seg, seg, num_labels=5)[3, 3])  # one device pass; bin 4 = "other" keeps labels >= 4 out of bin 3
    if n3 >= threshold:
        return seg.clone(), n3
    import torch
    table = np.arange(256, dtype=np.uint8)
    table[3] = replace_with
    seg = seg.contiguous()
    out = torch.empty_like(seg)
    stream = torch.cuda.current_stream(seg.device).cuda_stream
    _lib.check(_lib.load().mi355_label_remap(seg.data_ptr(), out.data_ptr(), seg.numel(),
                                             table.ctypes.data_as(C.POINTER(C.c_uint8)), stream), "mi355_label_remap")
    return out, n3


def confusion(pred, gt, num_labels=6):
    """K x K integer matrix, rows = predicted label, columns = ground truth.  The last bin (K-1) is "other": it
    collects every label >= K-1 (the reference compares ``== label`` per label, evaluate_segmentation.py:20-21, so a
    stray label must never be counted as a real one); the matrix always sums to the voxel count."""
    import torch
    if pred.shape != gt.shape or pred.dtype != torch.uint8 or gt.dtype != torch.uint8:
        raise ValueError("confusion: two uint8 tensors of equal shape expected (evaluate_segmentation.py:78-81)")
    pred, gt = pred.contiguous(), gt.contiguous()
    counts = np.zeros(num_labels * num_labels, dtype=np.uint64)
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    _lib.check(_lib.load().mi355_label_confusion(pred.data_ptr(), gt.data_ptr(), pred.numel(), num_labels,
                                                 counts.ctypes.data_as(C.POINTER(C.c_uint64)), stream),
               "mi355_label_confusion")
    return counts.reshape(num_labels, num_labels).astype(np.int64)


def _metrics(tp, fp, fn, tn=None):
    out = {"dice": (2 * tp) / (2 * tp + fp + fn + 1e-8), "iou": tp / (tp + fp + fn + 1e-8),
           "sensitivity": tp / (tp + fn + 1e-8), "tp": float(tp), "fp": float(fp), "fn": float(fn)}
    if tn is not None:
        out["specificity"] = tn / (tn + fp + 1e-8)
        out["tn"] = float(tn)
    return out


def metrics_from_confusion(cm, labels=(1, 2, 3)):
    """Per-label metrics (calculate_metrics) + WT {1,2,3}, TC {1,3}, ET {3} in the BraTS convention
    (evaluate_segmentation.py:129-162) + the mean Dice the pipeline reports."""
    cm = np.asarray(cm, dtype=np.float64)
    total = cm.sum()
    res = {}
    for lab in labels:
        tp = cm[lab, lab]
        fp = cm[lab, :].sum() - tp
        fn = cm[:, lab].sum() - tp
        res[lab] = _metrics(tp, fp, fn, total - tp - fp - fn)

    def region(members):
        m = list(members)
        tp = cm[np.ix_(m, m)].sum()
        fp = cm[m, :].sum() - tp
        fn = cm[:, m].sum() - tp
        return _metrics(tp, fp, fn)

    res["WT"], res["TC"], res["ET"] = region((1, 2, 3)), region((1, 3)), region((3,))
    res["mean_dice"] = float(np.mean([res["WT"]["dice"], res["TC"]["dice"], res["ET"]["dice"]]))
    return res


def evaluate(pred, gt):
    """pred, gt: CUDA uint8 label maps in the BraTS convention -> metrics dict."""
    return metrics_from_confusion(confusion(pred, gt, 6))  # labels 0..4 + "other"


# ---- feature_extraction/utils.py:167-216 on the device --------------------------------------------------------
REGION_LABELS = {"ncr": (1,), "ed": (2,), "et": (3, 4), "tc": (1, 3, 4), "wt": (1, 2, 3, 4)}  # utils.py:171-178


def label_stats(seg, num_labels=5):
    """seg: CUDA uint8 label map [d0, d1, d2] -> int64 array [num_labels, 10]: count, coordinate sums (3), minima (3),
    maxima (3) per label value (one pass on the device)."""
    import torch
    if seg.dtype != torch.uint8 or not seg.is_cuda or seg.dim() != 3:
        raise ValueError("label_stats: CUDA uint8 [d0, d1, d2] tensor expected")
    seg = seg.contiguous()
    out = np.zeros(num_labels * 10, dtype=np.int64)
    stream = torch.cuda.current_stream(seg.device).cuda_stream
    _lib.check(_lib.load().mi355_label_stats(seg.data_ptr(), seg.shape[0], seg.shape[1], seg.shape[2], num_labels,
                                             out.ctypes.data_as(C.POINTER(C.c_int64)), stream), "mi355_label_stats")
    return out.reshape(num_labels, 10)


def tumor_region_features(seg, voxel_volume_cm3):
    """Volumes, centroids and bounding boxes of the reference's tumour regions (utils.py:167-216: get_tumor_masks,
    calculate_volume, get_centroid, get_bounding_box) from one device pass; axes named x, y, z in array order as there."""
    st = label_stats(seg, 5)
    out = {}
    for name, labels in REGION_LABELS.items():
        rows = st[list(labels)]
        n = int(rows[:, 0].sum())
        feat = {"volume_cm3": float(n * voxel_volume_cm3), "centroid": None, "bounding_box": None}
        if n > 0:
            present = rows[rows[:, 0] > 0]
            sums = rows[:, 1:4].sum(axis=0)
            lo, hi = present[:, 4:7].min(axis=0), present[:, 7:10].max(axis=0)
            feat["centroid"] = {k: float(sums[i] / n) for i, k in enumerate("xyz")}
            feat["bounding_box"] = {**{f"min_{k}": int(lo[i]) for i, k in enumerate("xyz")},
                                    **{f"max_{k}": int(hi[i]) for i, k in enumerate("xyz")},
                                    **{f"size_{k}": int(hi[i] - lo[i] + 1) for i, k in enumerate("xyz")}}
        out[name] = feat
    return out
