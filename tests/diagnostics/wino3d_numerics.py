#!/usr/bin/env python3
"""CPU emulation of the f32 Winograd convolutions, whole network (numerics of F(2x2x2, 3x3x3) BEFORE a kernel).

Model A (BatchNorm folded into the convs in fp64 as mi355_unet_create does), one forward:
  ref     fp64 direct convolutions everywhere
  direct  fp32 direct convolutions everywhere (F.conv3d)
  wino2   F(2x2, 3x3) over (z, y), direct in x, on the layers the library sends to conv3_f32_wino2_kernel (stride 1, Cin % 16 == 0,
          levels 0-2), fp32 direct elsewhere - the shipped fp32 path
  wino3   F(2x2x2, 3x3x3) on the same layers
Winograd arithmetic as the kernels do it: U = G w G^T (per transformed axis) in fp64, rounded to fp32 once; input transform
B^T d B one fp32 add per output and axis; products accumulated in fp32; output transform A^T m A as fp32 adds.
Prints the max logit error / spread against `ref` (the gate of tests/test_gpu_network.py is 6e-5) and the rms.

    python tests/diagnostics/wino3d_numerics.py [--size 64] [--input noise|brain] [--seed 7]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import brats_amd as amd  # noqa: E402

CHAIN = False
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def in_transform(p, dim):
    """B^T along `dim` (size 4): d0-d2, d1+d2, d2-d1, d1-d3 - one fp32 operation per output."""
    d = p.unbind(dim)
    return torch.stack((d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]), dim)


def out_transform(m, dim):
    """A^T along `dim` (size 4 -> 2): (m0+m1)+m2, (m1-m2)-m3."""
    d = m.unbind(dim)
    return torch.stack(((d[0] + d[1]) + d[2], (d[1] - d[2]) - d[3]), dim)


def wino_conv(x, w, b, axes):
    """x [1,C,D,H,W] fp32, w [O,C,3,3,3] fp32 -> [1,O,D,H,W]; Winograd F(2,3) along the spatial `axes` (subset of (0,1,2)),
    direct along the others."""
    C, O = w.shape[1], w.shape[0]
    xp = F.pad(x[0], (1, 1, 1, 1, 1, 1))
    U = w.double()
    for a in axes:  # G along kernel axis a: 3 -> 4
        U = torch.tensordot(U, G, dims=([2 + a], [1]))          # contracted axis goes last
        U = U.movedim(-1, 2 + a)
    U = U.float()
    # patches: transformed axes -> (blocks, 4) stride 2; direct axes -> (positions, 3) stride 1
    p = xp
    for a in range(3):
        p = p.unfold(1 + a, 4 if a in axes else 3, 2 if a in axes else 1)
    # p: [C, n0, n1, n2, k0, k1, k2]
    for a in axes:
        p = in_transform(p, 4 + a)
    n = p.shape[1:4]
    k = p.shape[4:7]
    V = p.reshape(C, n[0] * n[1] * n[2], k[0], k[1], k[2])
    # transformed axes: elementwise in the transform domain; direct axes: summed (they are taps)
    t_sub = "".join("uvw"[a] for a in axes)
    sub_u = "oc" + "".join("uvw"[a] if a in axes else "xyz"[a] for a in range(3))
    sub_v = "ct" + "".join("uvw"[a] if a in axes else "xyz"[a] for a in range(3))
    if CHAIN:
        # the matrix cores accumulate one long fp32 chain per output (K = Cin x direct taps, two products per MFMA step): emulate
        # it as a sequential sum over (channel, direct tap), one rounding per product and per addition
        direct = [a for a in range(3) if a not in axes]
        M = torch.zeros((O, V.shape[1]) + tuple(4 for _ in axes), dtype=torch.float32)
        idx_t = [slice(None)] * 3
        for c in range(C):
            for taps in np.ndindex(*[3 for _ in direct]):
                iu, iv = list(idx_t), list(idx_t)
                for a, tp in zip(direct, taps):
                    iu[a] = tp; iv[a] = tp
                M += U[(slice(None), c) + tuple(iu)][:, None] * V[(c, slice(None)) + tuple(iv)][None]
    else:
        M = torch.einsum(f"{sub_u},{sub_v}->ot{t_sub}", U, V)     # fp32 accumulation (blocked: optimistic)
    for i, a in enumerate(axes):
        M = out_transform(M, 2 + i)
    # M: [O, T, 2 (per transformed axis)...] -> full-resolution volume
    M = M.reshape((O,) + tuple(n) + (2,) * len(axes))
    dims = [0]
    j = 0
    for a in range(3):
        dims.append(1 + a)
        if a in axes:
            dims.append(4 + j)
            j += 1
    M = M.permute(dims).reshape(O, *[n[a] * (2 if a in axes else 1) for a in range(3)])
    return (M + b.view(-1, 1, 1, 1))[None]


def fold_bn(sd, prefix, eps=1e-5):
    w = torch.from_numpy(np.asarray(sd[prefix + ".conv.weight"])).double()
    b = torch.from_numpy(np.asarray(sd[prefix + ".conv.bias"])).double()
    g = torch.from_numpy(np.asarray(sd[prefix + ".instnorm.weight"])).double()
    be = torch.from_numpy(np.asarray(sd[prefix + ".instnorm.bias"])).double()
    rm = torch.from_numpy(np.asarray(sd[prefix + ".instnorm.running_mean"])).double()
    rv = torch.from_numpy(np.asarray(sd[prefix + ".instnorm.running_var"])).double()
    s = g / torch.sqrt(rv.float().double() + eps)
    return (w * s.view(-1, 1, 1, 1, 1)).float(), (b * s + (be - rm * s)).float()


def forward(sd, x, mode, wino_levels=(0, 1, 2)):
    """mode: 'ref' (fp64 direct) | 'direct' | 'wino2' | 'wino3'."""
    dt = torch.float64 if mode == "ref" else torch.float32
    x = x.to(dt)

    def block(x, prefix, stride, level):
        w, b = fold_bn(sd, prefix)
        if mode in ("wino2", "wino3") and stride == 1 and w.shape[1] % 16 == 0 and level in wino_levels:
            y = wino_conv(x, w, b, (0, 1) if mode == "wino2" else (0, 1, 2))
        elif CHAIN and mode == "direct" and stride == 1 and w.shape[1] % 16 == 0 and level in wino_levels:
            y = wino_conv(x, w, b, ())   # no transformed axis: the plain 27 x Cin chain
        else:
            y = F.conv3d(x, w.to(dt), b.to(dt), stride=stride, padding=1)
        return F.leaky_relu(y, 0.01)

    num_pool = 5
    skips = []
    for d in range(num_pool):
        for i in range(2):
            x = block(x, f"conv_blocks_context.{d}.blocks.{i}", 2 if (d and i == 0) else 1, d)
        skips.append(x)
    x = block(x, f"conv_blocks_context.{num_pool}.0.blocks.0", 2, num_pool)
    x = block(x, f"conv_blocks_context.{num_pool}.1.blocks.0", 1, num_pool)
    for u in range(num_pool):
        x = F.conv_transpose3d(x, torch.from_numpy(np.asarray(sd[f"tu.{u}.weight"])).to(dt), None, stride=2)
        x = torch.cat((x, skips[-(u + 1)]), 1)
        lvl = num_pool - 1 - u
        x = block(x, f"conv_blocks_localization.{u}.0.blocks.0", 1, lvl)
        x = block(x, f"conv_blocks_localization.{u}.1.blocks.0", 1, lvl)
    return F.conv3d(x, torch.from_numpy(np.asarray(sd[f"seg_outputs.{num_pool - 1}.weight"])).to(dt)).double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--input", default="noise")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--xseed", type=int, default=1)
    ap.add_argument("--chain", action="store_true", help="sequential fp32 accumulation chains (as the MFMA K loop), not blocked einsum")
    args = ap.parse_args()
    global CHAIN
    CHAIN = args.chain
    torch.set_num_threads(8)
    sd, _ = amd.synthetic.make_model("A", seed=args.seed)
    n = args.size
    if args.input == "noise":
        x = torch.from_numpy(np.random.RandomState(args.xseed).standard_normal((1, 4, n, n, n)).astype(np.float32))
    else:
        vol = amd.synthetic.make_volume(seed=1000)
        m = (vol != 0).any(0)
        v = vol.astype(np.float32).copy()
        for c in range(4):
            q = v[c][m]; v[c][m] = (q - q.mean()) / (q.std() + 1e-8); v[c][~m] = 0
        zz, yy, xx = [s // 2 for s in v.shape[1:]]
        x = torch.from_numpy(v[None, :, zz - 60:zz - 60 + n, yy - 70:yy - 70 + n, xx - 32:xx - 32 + n].copy())
    with torch.no_grad():
        ref = forward(sd, x, "ref")
        spread = float(ref.std())
        for mode in ("direct", "wino2", "wino3"):
            y = forward(sd, x, mode)
            e = (y - ref).abs()
            print(f"WINO-NUMERICS model A seed {args.seed} {args.input} {n}^3 {'chain' if CHAIN else 'blocked'} {mode:6s}: logit err max {float(e.max()):.3e} = {float(e.max()) / max(spread, 1):.2e} x spread "
                  f"({spread:.2f}), rms {float((e ** 2).mean().sqrt()) / max(spread, 1):.2e} x spread", flush=True)


if __name__ == "__main__":
    main()
