"""Oracle: functional torch-CPU fp32 restatement of the reference network.

Follows /root/reference/model_architecture/generic_UNet.py:
  * ConvDropoutNormNonlin.forward      :68-72   conv -> (dropout p=0: absent) -> norm -> LeakyReLU(0.01)
  * ConvDropoutNonlinNorm.forward      :76-80   conv -> LeakyReLU -> norm        (``nonlin_first``)
  * StackedConvLayers                  :128-146 first block carries the stride
  * Generic_UNet.__init__              :283-391 which modules exist, their strides / channel widths
  * Generic_UNet.forward               :423-446 encoder, bottleneck, (tconv, concat(up, skip), 2 blocks) x num_pool

Everything is derived from the *state_dict* (tensor shapes and key names), never
from a width formula (SURVEY.md section 7 "hard parts": the large model's decoder
widths are irregular).  Test infrastructure only - see oracle/__init__.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    if isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _has(sd, key):
    return key in sd


def _norm(x, sd, prefix, kind, num_groups, eps):
    """The attribute is called ``instnorm`` whatever the type (generic_UNet.py:62-65)."""
    w = _t(sd[prefix + ".instnorm.weight"])
    b = _t(sd[prefix + ".instnorm.bias"])
    if kind == "batch":  # eval mode -> running statistics
        rm = _t(sd[prefix + ".instnorm.running_mean"])
        rv = _t(sd[prefix + ".instnorm.running_var"])
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, eps)
    if kind == "instance":
        return F.instance_norm(x, None, None, w, b, True, 0.0, eps)
    if kind == "group":
        return F.group_norm(x, num_groups, w, b, eps)
    raise ValueError(kind)


def _block(x, sd, prefix, stride, cfg):
    """One ConvDropoutNormNonlin (generic_UNet.py:27-72); dropout is absent at p=0 (:57-61)."""
    x = F.conv3d(x, _t(sd[prefix + ".conv.weight"]), _t(sd[prefix + ".conv.bias"]),
                 stride=stride, padding=1)
    if cfg.get("nonlin_first", False):  # ConvDropoutNonlinNorm :75-80
        x = F.leaky_relu(x, cfg["slope"])
        return _norm(x, sd, prefix, cfg["norm"], cfg["num_groups"], cfg["eps"])
    x = _norm(x, sd, prefix, cfg["norm"], cfg["num_groups"], cfg["eps"])
    return F.leaky_relu(x, cfg["slope"])


def default_cfg(norm="batch", num_groups=16, eps=1e-5, slope=1e-2, nonlin_first=False):
    return dict(norm=norm, num_groups=num_groups, eps=eps, slope=slope, nonlin_first=nonlin_first)


def num_pool_of(sd):
    n = 0
    while _has(sd, f"tu.{n}.weight"):
        n += 1
    return n


@torch.no_grad()
def unet_forward(sd, x, cfg, return_stages=False):
    """Logits of the full-resolution head, ``[N, num_classes, D, H, W]`` fp32.

    ``final_nonlin`` is the identity for the BraTS V2 trainers; only
    ``seg_outputs[-1]`` is returned at inference (generic_UNet.py:442-446, do_ds False),
    so the coarser heads the reference also evaluates (:440) are skipped - same result.
    """
    x = _t(x)
    num_pool = num_pool_of(sd)
    skips = []
    stages = {}
    # encoder (generic_UNet.py:426-430): conv_blocks_context[d] = StackedConvLayers
    for d in range(num_pool):
        i = 0
        while _has(sd, f"conv_blocks_context.{d}.blocks.{i}.conv.weight"):
            stride = 2 if (d != 0 and i == 0) else 1  # convolutional_pooling :285-288
            x = _block(x, sd, f"conv_blocks_context.{d}.blocks.{i}", stride, cfg)
            i += 1
        skips.append(x)
        stages[f"ctx{d}"] = x
    # bottleneck (:329-335, :432): Sequential of two StackedConvLayers, first conv stride 2
    j = 0
    first = True
    while _has(sd, f"conv_blocks_context.{num_pool}.{j}.blocks.0.conv.weight"):
        i = 0
        while _has(sd, f"conv_blocks_context.{num_pool}.{j}.blocks.{i}.conv.weight"):
            x = _block(x, sd, f"conv_blocks_context.{num_pool}.{j}.blocks.{i}", 2 if first else 1, cfg)
            first = False
            i += 1
        j += 1
    stages["bottleneck"] = x
    # decoder (:434-440)
    for u in range(num_pool):
        x = F.conv_transpose3d(x, _t(sd[f"tu.{u}.weight"]), None, stride=2)  # :363-364 bias=False
        stages[f"tu{u}"] = x
        x = torch.cat((x, skips[-(u + 1)]), dim=1)  # upsampled first, skip second :438
        j = 0
        while _has(sd, f"conv_blocks_localization.{u}.{j}.blocks.0.conv.weight"):
            i = 0
            while _has(sd, f"conv_blocks_localization.{u}.{j}.blocks.{i}.conv.weight"):
                x = _block(x, sd, f"conv_blocks_localization.{u}.{j}.blocks.{i}", 1, cfg)
                i += 1
            j += 1
        stages[f"loc{u}"] = x
    head = _t(sd[f"seg_outputs.{num_pool - 1}.weight"])
    hb = sd.get(f"seg_outputs.{num_pool - 1}.bias")
    logits = F.conv3d(x, head, None if hb is None else _t(hb))  # :389-391, 1x1x1, bias=False
    if return_stages:
        return logits, stages
    return logits


def conv_flops(sd, patch):
    """2*MAC of every Conv3d / ConvTranspose3d incl. all seg heads (SURVEY 8d rule)."""
    num_pool = num_pool_of(sd)
    total = 0
    vox = {0: int(np.prod(patch))}
    for d in range(1, num_pool + 1):
        vox[d] = vox[d - 1] // 8
    for k, v in sd.items():
        shp = tuple(v.shape)
        if len(shp) != 5:
            continue
        if k.startswith("conv_blocks_context."):
            lvl = int(k.split(".")[1])
            total += 2 * vox[lvl] * int(np.prod(shp))
        elif k.startswith("conv_blocks_localization.") or k.startswith("seg_outputs."):
            u = int(k.split(".")[1])
            total += 2 * vox[num_pool - 1 - u] * int(np.prod(shp))
        elif k.startswith("tu."):
            u = int(k.split(".")[1])
            total += 2 * vox[num_pool - u] * int(np.prod(shp))  # per *input* voxel: Cin*Cout*8
    return total
