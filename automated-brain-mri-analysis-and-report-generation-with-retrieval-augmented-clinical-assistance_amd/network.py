"""Host mirror of the reference network object: ``Generic_UNet`` on MI355X.

``UNet(state_dict, norm=...)`` takes the reference's own ``state_dict`` (key names and tensor
layouts of /root/reference/model_architecture/generic_UNet.py) and
builds the device network through the C ABI (``mi355_unet_create``).  Every channel width is
read from the tensor shapes, never from a formula: the large KAIST model's decoder is irregular
(generic_UNet.py:344-375 with ``encoder_scale=2``).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _lib

NORM_KINDS = {"none": _lib.NORM_NONE, "batch": _lib.NORM_BATCH, "instance": _lib.NORM_INSTANCE,
              "group": _lib.NORM_GROUP}


def _np32(a) -> np.ndarray:
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class ConvSpec:
    prefix: str
    cin: int
    cout: int
    stride: int


@dataclass
class UNetTopology:
    """What Generic_UNet.__init__ (generic_UNet.py:283-391) would have built for this state_dict."""
    in_channels: int
    num_classes: int
    num_pool: int
    enc: List[List[ConvSpec]] = field(default_factory=list)  # num_pool + 1 stages (last = bottleneck)
    dec: List[List[ConvSpec]] = field(default_factory=list)
    tu: List[tuple] = field(default_factory=list)            # (cin, cout) per decoder stage
    head_cin: int = 0
    has_batchnorm_stats: bool = False

    def conv_flops(self, patch) -> int:
        """2*MAC of every conv evaluated for one patch (the last seg head only)."""
        vox = [int(np.prod(patch)) // (8 ** l) for l in range(self.num_pool + 1)]
        total = 0
        for l, st in enumerate(self.enc):
            total += sum(2 * vox[l] * c.cin * c.cout * 27 for c in st)
        for u, st in enumerate(self.dec):
            l = self.num_pool - 1 - u
            total += 2 * vox[l + 1] * self.tu[u][0] * self.tu[u][1] * 8
            total += sum(2 * vox[l] * c.cin * c.cout * 27 for c in st)
        total += 2 * vox[0] * self.head_cin * self.num_classes
        return total


def strip_module_prefix(sd: Dict) -> Dict:
    """nnU-Net strips DataParallel's ``module.`` prefix on load."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def topology_from_state_dict(sd: Dict) -> UNetTopology:
    sd = strip_module_prefix(sd)
    if any(k.startswith("axial_") for k in sd):
        raise ValueError("checkpoint contains axial attention modules (generic_UNet.py:377-387): not on the "
                         "BraTS ensemble path, unsupported")
    num_pool = 0
    while f"tu.{num_pool}.weight" in sd:
        num_pool += 1
    if num_pool == 0:
        raise ValueError("state_dict has no tu.* transposed convolutions: not a convolutional-upsampling Generic_UNet")

    def stage(prefixes_of_blocks, first_stride):
        convs = []
        for p in prefixes_of_blocks:
            i = 0
            while f"{p}.blocks.{i}.conv.weight" in sd:
                w = sd[f"{p}.blocks.{i}.conv.weight"]
                if tuple(w.shape[2:]) != (3, 3, 3):
                    raise ValueError(f"{p}.blocks.{i}: kernel {tuple(w.shape[2:])} unsupported (3x3x3 only)")
                convs.append(ConvSpec(f"{p}.blocks.{i}", int(w.shape[1]), int(w.shape[0]),
                                      first_stride if not convs else 1))
                i += 1
        return convs

    def seq_prefixes(base):
        out, j = [], 0
        while f"{base}.{j}.blocks.0.conv.weight" in sd:
            out.append(f"{base}.{j}")
            j += 1
        return out

    topo = UNetTopology(in_channels=int(sd["conv_blocks_context.0.blocks.0.conv.weight"].shape[1]),
                        num_classes=int(sd[f"seg_outputs.{num_pool - 1}.weight"].shape[0]), num_pool=num_pool)
    for d in range(num_pool):
        topo.enc.append(stage([f"conv_blocks_context.{d}"], 1 if d == 0 else 2))
    topo.enc.append(stage(seq_prefixes(f"conv_blocks_context.{num_pool}"), 2))
    for u in range(num_pool):
        w = sd[f"tu.{u}.weight"]
        if tuple(w.shape[2:]) != (2, 2, 2):
            raise ValueError(f"tu.{u}: kernel {tuple(w.shape[2:])} unsupported (2x2x2 stride 2 only)")
        topo.tu.append((int(w.shape[0]), int(w.shape[1])))
        topo.dec.append(stage(seq_prefixes(f"conv_blocks_localization.{u}"), 1))
    topo.head_cin = int(sd[f"seg_outputs.{num_pool - 1}.weight"].shape[1])
    topo.has_batchnorm_stats = "conv_blocks_context.0.blocks.0.instnorm.running_mean" in sd
    for st in topo.enc + topo.dec:
        if not st:
            raise ValueError("empty conv stage in state_dict")
    return topo


class UNet:
    """Device-resident Generic_UNet.  ``forward`` mirrors generic_UNet.py:423-446 (NCDHW in/out)."""

    def __init__(self, state_dict: Dict, norm: str = "auto", num_groups: int = 16, eps: float = 1e-5,
                 lrelu_slope: float = 1e-2, nonlin_first: bool = False, dtype: str = "f32"):
        sd = strip_module_prefix(state_dict)
        self.topology = topology_from_state_dict(sd)
        if norm == "auto":
            if not self.topology.has_batchnorm_stats:
                raise ValueError("norm='auto' needs BatchNorm running stats in the state_dict; pass "
                                 "norm='instance' or norm='group' (+num_groups) explicitly")
            norm = "batch"
        if norm == "batch" and not self.topology.has_batchnorm_stats:
            raise ValueError("norm='batch' but the state_dict has no running_mean/running_var")
        self.norm = norm
        self.num_groups = num_groups
        self.dtype = dtype
        self._handle = C.c_void_p()
        lib = _lib.load()

        keep = []  # host arrays must outlive mi355_unet_create

        def arr(key, optional=False):
            if key not in sd:
                if optional:
                    return None
                raise KeyError(key)
            a = _np32(sd[key])
            keep.append(a)
            return a

        topo = self.topology
        all_convs = [c for st in topo.enc for c in st] + [c for st in topo.dec for c in st]
        convs = (_lib.ConvDesc * len(all_convs))()
        for i, c in enumerate(all_convs):
            d = convs[i]
            d.cin, d.cout, d.stride = c.cin, c.cout, c.stride
            d.weight = _lib.fptr(arr(c.prefix + ".conv.weight"))
            d.bias = _lib.fptr(arr(c.prefix + ".conv.bias", optional=True))
            if norm != "none":
                d.gamma = _lib.fptr(arr(c.prefix + ".instnorm.weight", optional=True))
                d.beta = _lib.fptr(arr(c.prefix + ".instnorm.bias", optional=True))
            if norm == "batch":
                d.running_mean = _lib.fptr(arr(c.prefix + ".instnorm.running_mean"))
                d.running_var = _lib.fptr(arr(c.prefix + ".instnorm.running_var"))
        tconvs = (_lib.TConvDesc * topo.num_pool)()
        for u in range(topo.num_pool):
            tconvs[u].cin, tconvs[u].cout = topo.tu[u]
            tconvs[u].weight = _lib.fptr(arr(f"tu.{u}.weight"))
        enc_counts = (C.c_int32 * (topo.num_pool + 1))(*[len(st) for st in topo.enc])
        dec_counts = (C.c_int32 * topo.num_pool)(*[len(st) for st in topo.dec])
        desc = _lib.UNetDesc()
        desc.in_channels, desc.num_classes, desc.num_pool = topo.in_channels, topo.num_classes, topo.num_pool
        desc.norm, desc.num_groups = NORM_KINDS[norm], int(num_groups)
        desc.eps, desc.lrelu_slope = float(eps), float(lrelu_slope)
        desc.nonlin_first = int(bool(nonlin_first))
        desc.dtype = {"f32": _lib.F32, "f16": _lib.F16}[dtype]
        desc.enc_convs, desc.dec_convs = enc_counts, dec_counts
        desc.convs, desc.n_convs, desc.tconvs = convs, len(all_convs), tconvs
        hw = arr(f"seg_outputs.{topo.num_pool - 1}.weight").reshape(topo.num_classes, topo.head_cin)
        keep.append(hw)
        desc.head.cin, desc.head.num_classes = topo.head_cin, topo.num_classes
        desc.head.weight = _lib.fptr(hw)
        desc.head.bias = _lib.fptr(arr(f"seg_outputs.{topo.num_pool - 1}.bias", optional=True))
        _lib.check(lib.mi355_unet_create(C.byref(desc), C.byref(self._handle)), "mi355_unet_create")
        del keep

    # -- lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            _lib.load().mi355_unet_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if not self._handle:
            raise _lib.Mi355Error("UNet handle is closed")
        return self._handle

    def flops(self, patch) -> int:
        return int(_lib.load().mi355_unet_flops(self.handle, int(patch[0]), int(patch[1]), int(patch[2])))

    def profile(self, on: bool = True):
        """Per-kernel HIP-event timing of every launch made through this handle (bench.py)."""
        _lib.check(_lib.load().mi355_profile_enable(self.handle, int(on)), "mi355_profile_enable")

    def read_profile(self):
        buf = (_lib.ProfEntry * 32)()
        n = _lib.load().mi355_profile_read(self.handle, buf, 32)
        _lib.check(n, "mi355_profile_read")
        return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), ms=float(buf[i].ms),
                     flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(n)]

    def forward(self, x):
        """x: torch.cuda fp32 [N, C, D, H, W] -> logits [N, num_classes, D, H, W] (final_nonlin = identity)."""
        import torch
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5):
            raise ValueError("UNet.forward expects a CUDA fp32 [N,C,D,H,W] tensor")
        if x.shape[1] != self.topology.in_channels:
            raise ValueError(f"expected {self.topology.in_channels} input channels, got {x.shape[1]}")
        x = x.contiguous()
        n, _, d, h, w = x.shape
        out = torch.empty((n, self.topology.num_classes, d, h, w), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.load().mi355_unet_forward(self.handle, x.data_ptr(), n, d, h, w, out.data_ptr(), stream),
                   "mi355_unet_forward")
        return out

    __call__ = forward
