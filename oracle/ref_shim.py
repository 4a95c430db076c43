"""Imports the reference's own network definition (read-only, from /root/reference) so the
oracle restatement can be checked against it and golden vectors can be generated.

``model_architecture/generic_UNet.py`` imports four modules of the un-vendored ``nnunet`` /
``axial_attention`` packages (generic_UNet.py:17,21,22,24).  They are third-party code absent
from this container; minimal in-memory stand-ins with their published behaviour are registered
in ``sys.modules`` for the duration of the import.  The reference file itself is
executed unmodified from where it lies.  Test infrastructure only; never runs on the GPU box
(/root/reference does not exist there).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_ROOT = os.environ.get("REFERENCE_ROOT", "/root/reference")
_REF_FILE = os.path.join(REFERENCE_ROOT, "model_architecture", "generic_UNet.py")


def reference_available() -> bool:
    return os.path.isfile(_REF_FILE)


class _InitWeights_He:
    """nnunet.network_architecture.initialization.InitWeights_He (nnU-Net v1, Apache-2.0)."""

    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module):
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)


class _SegmentationNetwork(nn.Module):
    """Base class of Generic_UNet upstream; only nn.Module behaviour is needed by forward()."""


_module = None


def load_reference_module():
    global _module
    if _module is not None:
        return _module
    if not reference_available():
        raise FileNotFoundError(_REF_FILE)
    stand_ins = {
        "nnunet": {},
        "nnunet.utilities": {},
        "nnunet.utilities.nd_softmax": {"softmax_helper": lambda x: F.softmax(x, 1)},
        "nnunet.network_architecture": {},
        "nnunet.network_architecture.initialization": {"InitWeights_He": _InitWeights_He},
        "nnunet.network_architecture.neural_network": {"SegmentationNetwork": _SegmentationNetwork},
        "axial_attention": {"AxialAttention": object, "AxialPositionalEmbedding": object},
    }
    saved = {k: sys.modules.get(k) for k in stand_ins}
    try:
        for name, attrs in stand_ins.items():
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        spec = importlib.util.spec_from_file_location("_reference_generic_UNet", _REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _module = mod
    return mod


def build_reference_net(norm: str, num_groups: int = 16, base: int = 32, num_pool: int = 5, in_ch: int = 4,
                        num_classes: int = 3, max_feat=None, encoder_scale: int = 1, nonlin_first: bool = False):
    """Generic_UNet exactly as the BraTS V2 trainers construct it (ctor args)."""
    mod = load_reference_module()
    norm_op = {"batch": nn.BatchNorm3d, "instance": nn.InstanceNorm3d, "group": nn.GroupNorm}[norm]
    kw = {"eps": 1e-5, "affine": True}
    if norm == "group":
        kw["num_groups"] = num_groups
    extra = {}
    if max_feat is not None:
        extra["max_num_features"] = max_feat
    if encoder_scale != 1:
        extra["encoder_scale"] = encoder_scale
    if nonlin_first:
        extra["basic_block"] = mod.ConvDropoutNonlinNorm
    net = mod.Generic_UNet(in_ch, base, num_classes, num_pool, 2, 2, nn.Conv3d, norm_op, kw, nn.Dropout3d,
                           {"p": 0, "inplace": True}, nn.LeakyReLU, {"negative_slope": 1e-2, "inplace": True},
                           True, False, lambda x: x, _InitWeights_He(1e-2), [[2, 2, 2]] * num_pool,
                           [[3, 3, 3]] * (num_pool + 1), False, True, True, **extra)
    net.eval()
    net.do_ds = False
    return net


def load_numpy_state_dict(net, sd):
    """Loads a {key: np.ndarray} state_dict; heads the synthetic dict omits keep their init."""
    tsd = {k: torch.from_numpy(v.copy()) if hasattr(v, "dtype") and not isinstance(v, torch.Tensor) else v
           for k, v in sd.items()}
    missing, unexpected = net.load_state_dict(tsd, strict=False)
    bad = [k for k in missing if not k.startswith("seg_outputs.")]
    if bad or unexpected:
        raise RuntimeError(f"state_dict mismatch: missing {bad}, unexpected {list(unexpected)}")
    return net
