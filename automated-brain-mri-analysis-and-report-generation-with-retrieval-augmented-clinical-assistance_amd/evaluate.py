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
    n3 = int(confusion(seg, seg, num_labels=4)[3, 3])  # one device pass over the label map
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


def confusion(pred, gt, num_labels=5):
    """K x K integer matrix, rows = predicted label, columns = ground truth."""
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
    return metrics_from_confusion(confusion(pred, gt, 5))
