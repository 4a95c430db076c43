"""Sliding-window predictor: the host mirror of what the reference driver asks of nnU-Net v1.

``predict_preprocessed_data_return_seg_and_softmax`` keeps the name and argument meaning of the
trainer method the reference calls (run_brats2021_inference_singlethread.py:97-106); underneath
it is one call into the HIP library (``mi355_sw_predict``): tile gather with mirror flips, the
network, sigmoid / softmax, flip-back, Gaussian-weighted aggregation and normalisation all stay
on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .network import UNet

NONLIN = {"identity": _lib.NONLIN_IDENTITY, "sigmoid": _lib.NONLIN_SIGMOID, "softmax": _lib.NONLIN_SOFTMAX}


def _opts(patch_size, step_size, use_gaussian, do_mirroring, mirror_axes, nonlin, batch_tiles):
    o = _lib.SwOpts()
    o.patch[0], o.patch[1], o.patch[2] = [int(p) for p in patch_size]
    o.step_size = float(step_size)
    o.use_gaussian = int(bool(use_gaussian))
    mask = 0
    if do_mirroring:
        for a in mirror_axes:
            if a not in (0, 1, 2):
                raise ValueError(f"mirror axis {a} out of range")
            mask |= 1 << a
    o.mirror_axes = mask
    o.nonlin = NONLIN[nonlin]
    o.batch_tiles = int(batch_tiles)
    return o


def _to_device(data, device):
    import torch
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
    if not data.is_cuda:
        data = data.to(device, non_blocking=True)
    if data.dtype != torch.float32:
        data = data.float()
    return data.contiguous()


_LANE_STREAMS = {}   # device index -> the side streams of predict_folds(lanes > 1): created once, the library keeps a lane of scratch per stream


def default_lanes() -> int:
    """MI355_LANES (default 2): how many HIP streams one ``predict_folds`` call spreads its (fold, tile) work list over."""
    import os
    try:
        return max(1, min(4, int(os.environ.get("MI355_LANES", "2"))))
    except ValueError:
        return 2


def _lane_streams(device, n):
    import torch
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    have = _LANE_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=key))
    return have[:n]


def _predict_folds_lanes(nets, data, lanes, patch_size, step_size, do_mirroring, mirror_axes, use_gaussian, nonlin, batch_tiles):
    """``predict_folds`` over ``lanes`` HIP streams of ONE GPU: the (fold, tile) work list is dealt round-robin over the lanes exactly
    as ``mi355_sw_partial_folds`` deals it over the ranks of SURVEY.md 8e partitioning B; every lane accumulates its items into an
    aggregate of its own on a stream of its own (the library keeps an activation arena per stream), the aggregates are added in
    lane order and normalised once.  Why: a sixth of a TTA step is HBM-bound kernels (norm passes, transposed convs, first layer,
    aggregation) that leave the matrix cores idle, and the deep levels' launches do not fill the chip; with two lanes in flight
    the hardware runs one lane's memory-bound kernels beside the other's matrix-bound ones."""
    import torch
    cur = torch.cuda.current_stream(data.device)
    ready = cur.record_event()
    parts = []
    for r, st in enumerate(_lane_streams(data.device, lanes)):
        st.wait_event(ready)
        with torch.cuda.stream(st):
            agg, cnt = predict_tile_sharded(list(nets), data, r, lanes, patch_size, step_size, do_mirroring, mirror_axes, use_gaussian,
                                            nonlin, batch_tiles, data.device, want_cnt=(r == 0))
        for t in (agg, cnt):
            if t is not None:
                t.record_stream(cur)   # (allocated on the lane's stream, consumed on the caller's)
        parts.append((agg, cnt))
        cur.wait_stream(st)
    total = parts[0][0]
    for agg, _ in parts[1:]:
        total += agg   # lane order: deterministic
    return finish_sharded(total, parts[0][1], tuple(data.shape[1:]), patch_size, len(nets))


def _n_items(nets, shape_zyx, patch_size, step_size):
    from . import ops
    return len(nets) * int(np.prod([len(ops.compute_steps(int(p), max(int(p), int(d)), float(step_size)))
                                    for p, d in zip(patch_size, shape_zyx)]))


def predict_members(members: Sequence[Sequence[UNet]], data, patch_size=(128, 128, 128), step_size=0.5, do_mirroring=True,
                    mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid", batch_tiles=0, device="cuda", lanes=None):
    """``[predict_folds(m, data, ...) for m in members]`` - the ensemble members of the reference's driver (:263-264: model 1, then
    model 2) on one preprocessed volume - with ALL members' work enqueued on the lanes before the caller's stream joins them (one
    join per case instead of one per member): every lane takes its share of every member's (fold, tile) list.  Optionally
    staggered (lane r starts with member r; measured neutral, see below).  Results are those of ``predict_folds(lanes = ...)`` bit
    for bit (same per-lane item lists, same lane-ordered sum)."""
    import torch
    members = [list(m) for m in members]
    if not members or any(len(m) == 0 for m in members):
        raise ValueError("no networks given")
    data = _to_device(data, device)
    lanes = default_lanes() if lanes is None else int(lanes)
    _, z, y, x = data.shape
    n_mirrors = 2 ** len(set(mirror_axes)) if do_mirroring else 1
    ok = lanes > 1 and all(_n_items(m, (z, y, x), patch_size, step_size) * n_mirrors >= 32 * lanes for m in members)
    if not ok or len(members) == 1:
        return [predict_folds(m, data, patch_size, step_size, do_mirroring, mirror_axes, use_gaussian, nonlin, batch_tiles, device, lanes)
                for m in members]
    import os
    # (MI355_LANE_STAGGER=1: lane r starts with member r.  Measured on config 3 fp16, alternating runs on one box: 267.3 / 268.0 ms
    #  without, 268.9 / 268.4 with, 271.3 on one lane (profiles/r05_lanes_stagger_ab.txt) - the lanes drift apart by themselves;
    #  off by default)
    stagger = 1 if os.environ.get("MI355_LANE_STAGGER", "0") == "1" else 0
    cur = torch.cuda.current_stream(data.device)
    ready = cur.record_event()
    parts = [[None] * lanes for _ in members]
    for r, st in enumerate(_lane_streams(data.device, lanes)):
        st.wait_event(ready)
        with torch.cuda.stream(st):
            for k in range(len(members)):
                mi = (r * stagger + k) % len(members)
                parts[mi][r] = predict_tile_sharded(members[mi], data, r, lanes, patch_size, step_size, do_mirroring, mirror_axes,
                                                    use_gaussian, nonlin, batch_tiles, data.device, want_cnt=(r == 0))
        for mi in range(len(members)):
            for t in parts[mi][r]:
                if t is not None:
                    t.record_stream(cur)
        cur.wait_stream(st)
    out = []
    for mi, m in enumerate(members):
        total = parts[mi][0][0]
        for agg, _ in parts[mi][1:]:
            total += agg
        out.append(finish_sharded(total, parts[mi][0][1], (z, y, x), patch_size, len(m)))
    return out


def predict_folds(nets: Sequence[UNet], data, patch_size=(128, 128, 128), step_size=0.5, do_mirroring=True,
                  mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid", batch_tiles=0, device="cuda", lanes=None):
    """Class probabilities ``[K, Z, Y, X]`` (CUDA fp32) of one preprocessed ``[C, Z, Y, X]`` volume,
    averaged over ``nets`` (the fold ensemble of driver :95-128).  ``lanes`` (default ``MI355_LANES``, 2): number of HIP streams
    the work list is spread over when every lane still gets 32 (tile, mirror) samples; otherwise, and with 1, one ``mi355_sw_predict`` call."""
    import torch
    if len(nets) == 0:
        raise ValueError("no networks given")
    data = _to_device(data, device)
    if data.dim() != 4 or data.shape[0] != nets[0].topology.in_channels:
        raise ValueError(f"data must be [C={nets[0].topology.in_channels}, Z, Y, X]")
    _, z, y, x = data.shape
    lanes = default_lanes() if lanes is None else int(lanes)
    if lanes > 1:
        n_items = _n_items(nets, (z, y, x), patch_size, step_size)
        # (the split pays when every lane still runs full forward batches - 32 samples = (tile, mirror) pairs per lane; measured:
        #  config 3 fp16, 8 tiles x 8 mirrors per member: 273 -> 266 ms with two lanes, 272 with three; config 2, 8 tiles without
        #  mirrors: 28.3 ms either way, so it stays on one lane and bit-identical with mi355_sw_predict)
        n_mirrors = 2 ** len(set(mirror_axes)) if do_mirroring else 1
        if n_items * n_mirrors >= 32 * lanes:
            return _predict_folds_lanes(nets, data, lanes, patch_size, step_size, do_mirroring, mirror_axes, use_gaussian, nonlin, batch_tiles)
    k = nets[0].topology.num_classes
    probs = torch.empty((k, z, y, x), dtype=torch.float32, device=data.device)
    opts = _opts(patch_size, step_size, use_gaussian, do_mirroring, mirror_axes, nonlin, batch_tiles)
    handles = (C.c_void_p * len(nets))(*[n.handle for n in nets])
    stream = torch.cuda.current_stream(data.device).cuda_stream
    _lib.check(_lib.load().mi355_sw_predict(handles, len(nets), data.data_ptr(), z, y, x, C.byref(opts),
                                            probs.data_ptr(), stream), "mi355_sw_predict")
    return probs


def predict_preprocessed_data_return_seg_and_softmax(net: UNet, data, do_mirroring=True, mirror_axes=(0, 1, 2),
                                                     use_sliding_window=True, step_size=0.5, use_gaussian=True,
                                                     patch_size=(128, 128, 128), regions_class_order=(1, 2, 3),
                                                     nonlin="sigmoid", all_in_gpu=True, mixed_precision=False,
                                                     batch_tiles=0):
    """Same contract as nnUNetTrainer.predict_preprocessed_data_return_seg_and_softmax as the reference
    calls it: returns ``(seg, class_probabilities)``; the driver uses ``[1]``."""
    from . import ops
    if not use_sliding_window:
        raise NotImplementedError("the reference path always uses the sliding window (driver :101)")
    probs = predict_folds([net], data, patch_size, step_size, do_mirroring, mirror_axes, use_gaussian, nonlin,
                          batch_tiles)
    # regions_class_order=None: argmax variant documented at PROJECT_DOCUMENTATION.md:325-344
    seg = ops.regions_to_labels(probs, regions_class_order)
    return seg, probs


def predict_tile_sharded(net, data, rank: int, world: int, patch_size=(128, 128, 128), step_size=0.5,
                         do_mirroring=True, mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid",
                         batch_tiles=0, device="cuda", want_cnt=True):
    """This rank's share of one volume: returns (agg [K,Zp,Yp,Xp], cnt [Zp,Yp,Xp]); ``finish_sharded`` turns the
    rank-ordered sum into probabilities.  ``net`` is one network (tiles with index % world == rank) or the FOLD LIST of
    one ensemble member (driver :161, :112-128): then the work list is (fold, tile), item ``f * tiles + t`` goes to rank
    ``item % world``, and ``agg`` is the sum of this rank's items over all folds (``mi355_sw_partial_folds``)."""
    import torch
    nets = list(net) if isinstance(net, (list, tuple)) else [net]
    if len(nets) == 0:
        raise ValueError("no networks given")
    data = _to_device(data, device)
    _, z, y, x = data.shape
    zp, yp, xp = (max(z, patch_size[0]), max(y, patch_size[1]), max(x, patch_size[2]))
    k = nets[0].topology.num_classes
    agg = torch.empty((k, zp, yp, xp), dtype=torch.float32, device=data.device)
    cnt = torch.empty((zp, yp, xp), dtype=torch.float32, device=data.device) if want_cnt else None   # (the normaliser is the same on every rank)
    opts = _opts(patch_size, step_size, use_gaussian, do_mirroring, mirror_axes, nonlin, batch_tiles)
    handles = (C.c_void_p * len(nets))(*[n.handle for n in nets])
    stream = torch.cuda.current_stream(data.device).cuda_stream
    _lib.check(_lib.load().mi355_sw_partial_folds(handles, len(nets), data.data_ptr(), z, y, x, C.byref(opts), rank, world,
                                                  agg.data_ptr(), cnt.data_ptr() if want_cnt else None, stream), "mi355_sw_partial_folds")
    return agg, cnt


def finish_sharded(agg, cnt, vol_shape, patch_size, n_folds=1):
    """probs = agg / cnt / n_folds, cropped to the unpadded volume (``agg`` = the sum of every rank's partial)."""
    import torch
    k = agg.shape[0]
    z, y, x = vol_shape
    probs = torch.empty((k, z, y, x), dtype=torch.float32, device=agg.device)
    p = (C.c_int32 * 3)(*[int(v) for v in patch_size])
    stream = torch.cuda.current_stream(agg.device).cuda_stream
    _lib.check(_lib.load().mi355_sw_finish_folds(agg.data_ptr(), cnt.data_ptr(), k, z, y, x, p, int(n_folds), probs.data_ptr(),
                                                 stream), "mi355_sw_finish_folds")
    return probs
