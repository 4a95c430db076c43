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


def predict_folds(nets: Sequence[UNet], data, patch_size=(128, 128, 128), step_size=0.5, do_mirroring=True,
                  mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid", batch_tiles=0, device="cuda"):
    """Class probabilities ``[K, Z, Y, X]`` (CUDA fp32) of one preprocessed ``[C, Z, Y, X]`` volume,
    averaged over ``nets`` (the fold ensemble of driver :95-128)."""
    import torch
    if len(nets) == 0:
        raise ValueError("no networks given")
    data = _to_device(data, device)
    if data.dim() != 4 or data.shape[0] != nets[0].topology.in_channels:
        raise ValueError(f"data must be [C={nets[0].topology.in_channels}, Z, Y, X]")
    _, z, y, x = data.shape
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
                         batch_tiles=0, device="cuda"):
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
    cnt = torch.empty((zp, yp, xp), dtype=torch.float32, device=data.device)
    opts = _opts(patch_size, step_size, use_gaussian, do_mirroring, mirror_axes, nonlin, batch_tiles)
    handles = (C.c_void_p * len(nets))(*[n.handle for n in nets])
    stream = torch.cuda.current_stream(data.device).cuda_stream
    _lib.check(_lib.load().mi355_sw_partial_folds(handles, len(nets), data.data_ptr(), z, y, x, C.byref(opts), rank, world,
                                                  agg.data_ptr(), cnt.data_ptr(), stream), "mi355_sw_partial_folds")
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
