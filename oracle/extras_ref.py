"""Oracle for the section-8f rows: the reference's own numpy expressions, restated verbatim.
Test infrastructure only."""
from __future__ import annotations

import re

import numpy as np


def convert_labels(seg, fmt="brats2025"):
    """convert_labels_to_brats.py:34-55"""
    seg = np.round(seg).astype(np.uint8)
    new_seg = np.zeros_like(seg)
    new_seg[seg == 1] = 2
    new_seg[seg == 2] = 1
    new_seg[seg == 3] = 3 if fmt == "brats2025" else 4
    return new_seg


def calculate_metrics(pred, gt, label):
    """evaluate_segmentation.py:12-49 (float64 sums instead of float32: exact for any volume size)"""
    pred_mask = (pred == label).astype(np.float64)
    gt_mask = (gt == label).astype(np.float64)
    tp = np.sum(pred_mask * gt_mask)
    fp = np.sum(pred_mask * (1 - gt_mask))
    fn = np.sum((1 - pred_mask) * gt_mask)
    tn = np.sum((1 - pred_mask) * (1 - gt_mask))
    return {"dice": (2 * tp) / (2 * tp + fp + fn + 1e-8), "iou": tp / (tp + fp + fn + 1e-8),
            "sensitivity": tp / (tp + fn + 1e-8), "specificity": tn / (tn + fp + 1e-8), "tp": tp, "fp": fp, "fn": fn, "tn": tn}


def calculate_metrics_binary(pred_mask, gt_mask):
    """evaluate_segmentation.py:181-195"""
    tp = np.sum(pred_mask * gt_mask)
    fp = np.sum(pred_mask * (1 - gt_mask))
    fn = np.sum((1 - pred_mask) * gt_mask)
    return {"dice": (2 * tp) / (2 * tp + fp + fn + 1e-8), "iou": tp / (tp + fp + fn + 1e-8), "sensitivity": tp / (tp + fn + 1e-8)}


def compound(pred, gt):
    """evaluate_segmentation.py:129-162"""
    out = {}
    for name, members in (("WT", [1, 2, 3]), ("TC", [1, 3]), ("ET", [3])):
        out[name] = calculate_metrics_binary(np.isin(pred, members).astype(np.float64), np.isin(gt, members).astype(np.float64))
    out["mean_dice"] = float(np.mean([out["WT"]["dice"], out["TC"]["dice"], out["ET"]["dice"]]))
    return out


class DummyVectorStoreRef:
    """RAG_Assistant/rag_assistant.py:131-211, numpy only."""

    def __init__(self, documents):
        self.documents = documents
        toks = [re.findall(r"[a-z]+", d["text"].lower()) for d in documents]
        self.vocab = sorted(set(t for tt in toks for t in tt))
        w2i = {w: i for i, w in enumerate(self.vocab)}
        m = np.zeros((len(documents), len(self.vocab)))
        for r, tt in enumerate(toks):
            for t in tt:
                m[r, w2i[t]] += 1
        n = np.linalg.norm(m, axis=1, keepdims=True)
        n[n == 0] = 1
        self.vectors = m / n
        self.w2i = w2i

    def query_vector(self, query):
        vec = np.zeros(len(self.vocab))
        for t in re.findall(r"[a-z]+", query.lower()):
            if t in self.w2i:
                vec[self.w2i[t]] += 1
        n = np.linalg.norm(vec)
        return vec / n if n > 0 else vec

    def retrieve(self, query, top_k=2):
        scores = self.vectors @ self.query_vector(query)
        top = np.argsort(scores)[::-1][:top_k]
        return [(self.documents[i], float(scores[i])) for i in top]


def tumor_region_features(seg_data, voxel_volume_cm3):
    """feature_extraction/utils.py:167-216 verbatim in behaviour: get_tumor_masks, calculate_volume, get_centroid,
    get_bounding_box for the regions ncr, ed, et, tc, wt."""
    seg = np.round(seg_data).astype(np.int32)
    masks = {"ncr": seg == 1, "ed": seg == 2, "et": (seg == 3) | (seg == 4), "tc": (seg == 1) | (seg == 3) | (seg == 4),
             "wt": seg > 0}
    out = {}
    for name, mask in masks.items():
        feat = {"volume_cm3": float(mask.sum() * voxel_volume_cm3), "centroid": None, "bounding_box": None}
        if mask.sum() > 0:
            coords = np.array(np.where(mask)).T
            c = coords.mean(axis=0)
            feat["centroid"] = {"x": float(c[0]), "y": float(c[1]), "z": float(c[2])}
            w = np.where(mask)
            feat["bounding_box"] = {"min_x": int(w[0].min()), "min_y": int(w[1].min()), "min_z": int(w[2].min()),
                                    "max_x": int(w[0].max()), "max_y": int(w[1].max()), "max_z": int(w[2].max()),
                                    "size_x": int(w[0].max() - w[0].min() + 1), "size_y": int(w[1].max() - w[1].min() + 1),
                                    "size_z": int(w[2].max() - w[2].min() + 1)}
        out[name] = feat
    return out
