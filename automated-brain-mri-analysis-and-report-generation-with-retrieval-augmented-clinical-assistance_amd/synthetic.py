"""Seeded synthetic checkpoints and BraTS-shaped volumes.

Neither the KAIST checkpoints nor BraTS data exist in the build or GPU containers
(reference .gitignore:12,20,28-29), so tests, smoke and bench all use these generators
(SURVEY.md 8d).  ``np.random.RandomState`` streams are frozen across numpy versions, so the same
seed gives the same weights here, on the GPU box and in the committed golden fixtures.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np


def unet_widths(in_ch, base, num_classes, num_pool, max_feat=320, encoder_scale=1, convs_per_stage=2):
    """Channel plan of Generic_UNet.__init__ (reference generic_UNet.py:280-375) for
    convolutional_pooling=True, convolutional_upsampling=True.  Returns
    (enc[(cin,cout)...] per stage incl. bottleneck, tu[(cin,cout)], dec[[(cin,cout)...]], head_cin)."""
    enc = []
    out_f = base * encoder_scale
    in_f = in_ch
    for _ in range(num_pool):
        enc.append([(in_f, out_f)] + [(out_f, out_f)] * (convs_per_stage - 1))
        in_f = out_f
        out_f = min(int(np.round(out_f * 2)), max_feat)
    final = out_f
    enc.append([(in_f, out_f)] + [(out_f, out_f)] * (convs_per_stage - 2) + [(out_f, final)])
    tu, dec = [], []
    head_cin = None
    for u in range(num_pool):
        from_down = final if u == 0 else int(final / encoder_scale)
        from_skip = enc[-(2 + u)][-1][1]
        final = from_skip
        tu.append((from_down, from_skip))
        last = int(final / encoder_scale)
        dec.append([(2 * from_skip, from_skip)] + [(from_skip, from_skip)] * (convs_per_stage - 2) + [(from_skip, last)])
        head_cin = last
    return enc, tu, dec, head_cin


def make_state_dict(norm: str = "batch", seed: int = 7, in_ch: int = 4, base: int = 32, num_classes: int = 3,
                    num_pool: int = 5, max_feat: int = 320, encoder_scale: int = 1, head_scale: float = 12.0,
                    all_heads: bool = False) -> "OrderedDict[str, np.ndarray]":
    """He-normal (a=0.01) conv weights, zero conv bias, gamma~U(.5,1.5), beta~N(0,.1); BatchNorm
    running mean~N(0,.1), var~U(.5,1.5).  The segmentation head is scaled by ``head_scale`` so the
    logits are confidently bimodal like a trained model's (threshold sensitivity)."""
    rs = np.random.RandomState(seed)
    enc, tu, dec, head_cin = unet_widths(in_ch, base, num_classes, num_pool, max_feat, encoder_scale)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    a = 0.01

    def conv_block(prefix, cin, cout):
        std = np.sqrt(2.0 / ((1 + a * a) * cin * 27))
        sd[prefix + ".conv.weight"] = (rs.standard_normal((cout, cin, 3, 3, 3)) * std).astype(np.float32)
        sd[prefix + ".conv.bias"] = (rs.standard_normal(cout) * 0.05).astype(np.float32)
        if norm != "none":
            sd[prefix + ".instnorm.weight"] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[prefix + ".instnorm.bias"] = (rs.standard_normal(cout) * 0.1).astype(np.float32)
        if norm == "batch":
            sd[prefix + ".instnorm.running_mean"] = (rs.standard_normal(cout) * 0.1).astype(np.float32)
            sd[prefix + ".instnorm.running_var"] = rs.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[prefix + ".instnorm.num_batches_tracked"] = np.asarray(1000, dtype=np.int64)

    for d in range(num_pool):
        for i, (ci, co) in enumerate(enc[d]):
            conv_block(f"conv_blocks_context.{d}.blocks.{i}", ci, co)
    # bottleneck = Sequential(StackedConvLayers(n-1 convs), StackedConvLayers(1 conv))  (:329-335)
    bott = enc[num_pool]
    for i, (ci, co) in enumerate(bott[:-1]):
        conv_block(f"conv_blocks_context.{num_pool}.0.blocks.{i}", ci, co)
    conv_block(f"conv_blocks_context.{num_pool}.1.blocks.0", *bott[-1])
    for u in range(num_pool):
        ci, co = tu[u]
        std = np.sqrt(2.0 / ((1 + a * a) * co * 8))  # torch fan_in of a ConvTranspose weight [cin,cout,k]
        sd[f"tu.{u}.weight"] = (rs.standard_normal((ci, co, 2, 2, 2)) * std).astype(np.float32)
        blocks = dec[u]
        for i, (bi, bo) in enumerate(blocks[:-1]):
            conv_block(f"conv_blocks_localization.{u}.0.blocks.{i}", bi, bo)
        conv_block(f"conv_blocks_localization.{u}.1.blocks.0", *blocks[-1])
        if all_heads or u == num_pool - 1:
            hc = blocks[-1][1]
            std = np.sqrt(2.0 / ((1 + a * a) * hc))
            sd[f"seg_outputs.{u}.weight"] = (rs.standard_normal((num_classes, hc, 1, 1, 1)) * std * head_scale).astype(np.float32)
    return sd


#: the named synthetic models of SURVEY 8d ("model A" = base BN net, "model B" = large GroupNorm-16 net)
MODEL_PRESETS = {
    "A": dict(norm="batch", base=32, max_feat=320, encoder_scale=1),
    "A_in": dict(norm="instance", base=32, max_feat=320, encoder_scale=1),
    "B": dict(norm="group", base=32, max_feat=512, encoder_scale=2),
}


def make_model(name: str, seed: int = 7, **overrides) -> Tuple["OrderedDict[str, np.ndarray]", Dict]:
    cfg = dict(MODEL_PRESETS[name])
    cfg.update(overrides)
    sd = make_state_dict(seed=seed, **cfg)
    meta = dict(norm=cfg["norm"], num_groups=16)
    return sd, meta


def make_volume(seed: int = 1000, shape=(155, 240, 240), dense: bool = False, channels: int = 4,
                semi_axes: Optional[Tuple[float, float, float]] = None) -> np.ndarray:
    """Synthetic ``[4, Z, Y, X]`` float32 'MRI': smooth field x ellipsoidal brain mask (exact zeros
    outside, a few interior zero holes), MR-like intensities 0..~4000.  The default
    mask gives a nonzero crop of about (140,172,138) -> 8 tiles of 128^3; ``dense`` -> 18 tiles."""
    from scipy.ndimage import gaussian_filter

    rs = np.random.RandomState(seed)
    Z, Y, X = shape
    vol = np.empty((channels, Z, Y, X), dtype=np.float32)
    for c in range(channels):
        noise = rs.standard_normal((Z, Y, X)).astype(np.float32)
        smooth = gaussian_filter(noise, sigma=6.0, mode="nearest")
        smooth = (smooth - smooth.min()) / (smooth.max() - smooth.min() + 1e-12)
        vol[c] = (200.0 + 400.0 * c) + smooth * (1500.0 + 500.0 * c)
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    cz, cy, cx = (Z - 1) / 2.0, (Y - 1) / 2.0, (X - 1) / 2.0
    # bright "tumour" blobs
    for _ in range(rs.randint(1, 4)):
        bz, by, bx = cz + rs.uniform(-25, 25), cy + rs.uniform(-35, 35), cx + rs.uniform(-30, 30)
        r = rs.uniform(8, 18)
        blob = np.exp(-(((zz - bz) ** 2 + (yy - by) ** 2 + (xx - bx) ** 2) / (2 * r * r))).astype(np.float32)
        vol += blob[None] * rs.uniform(500, 1500, size=(channels, 1, 1, 1)).astype(np.float32)
    if not dense:
        if semi_axes is None:
            semi_axes = (70.0 * Z / 155.0, 86.0 * Y / 240.0, 69.0 * X / 240.0)
        mask = (((zz - cz) / semi_axes[0]) ** 2 + ((yy - cy) / semi_axes[1]) ** 2 + ((xx - cx) / semi_axes[2]) ** 2) <= 1.0
        vol *= mask[None]
        # interior zero holes (exercise binary_fill_holes in the preprocessing)
        for _ in range(3):
            hz, hy, hx = int(cz + rs.uniform(-15, 15)), int(cy + rs.uniform(-20, 20)), int(cx + rs.uniform(-20, 20))
            vol[:, max(hz - 2, 0):hz + 2, max(hy - 2, 0):hy + 2, max(hx - 2, 0):hx + 2] = 0.0
    return vol
