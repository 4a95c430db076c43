#!/usr/bin/env python3
"""CPU emulation (torch fp32 + explicit fp16 roundings in the oracle network, no GPU): which rounding makes the fp16 path's logit error -
the fp16 WEIGHTS (MFMA A operand), the fp16 ACTIVATIONS stored after every block, or both?  Model and seed from argv, a 64^3
brain-like tile of synthetic.make_volume(1000).    python tests/diagnostics/f16_error_sources.py B 8"""
import sys, numpy as np, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import brats_amd as amd
from oracle import unet_ref, tiler_ref
torch.set_num_threads(8)
name, seed = sys.argv[1], int(sys.argv[2])
sd, meta = amd.synthetic.make_model(name, seed=seed)
cfg = unet_ref.default_cfg(norm=meta["norm"], num_groups=meta["num_groups"])
vol = amd.synthetic.make_volume(seed=1000)
# crude CPU preprocessing: z-score inside nonzero mask, take a central 64^3 block inside the brain
m = (vol != 0).any(0)
x = vol.astype(np.float32).copy()
for c in range(4):
    v = x[c][m]; x[c][m] = (v - v.mean()) / (v.std() + 1e-8); x[c][~m] = 0
zz, yy, xx = [s // 2 for s in x.shape[1:]]
tile = x[None, :, zz-60:zz+4, yy-70:yy-6, xx-32:xx+32].copy()
ref = unet_ref.unet_forward(sd, tile, cfg).numpy()
h = lambda t: t.to(torch.float16).to(torch.float32)
def run(wq, aq, inq=False):
    sd2 = {k: (h(torch.from_numpy(np.asarray(v))).numpy() if (wq and (k.endswith("conv.weight") or k.startswith("tu."))) else v) for k, v in sd.items()}
    orig_block = unet_ref._block
    orig_tconv = F.conv_transpose3d
    def blk(x_, sd_, prefix, stride, cfg_):
        y = orig_block(x_, sd_, prefix, stride, cfg_)
        return h(y) if aq else y
    def tc(x_, w_, b_, stride=2):
        y = orig_tconv(x_, w_, b_, stride=stride)
        return h(y) if aq else y
    unet_ref._block = blk; F.conv_transpose3d = tc
    try:
        out = unet_ref.unet_forward(sd2, h(torch.from_numpy(tile)) if inq else tile, cfg).numpy()
    finally:
        unet_ref._block = orig_block; F.conv_transpose3d = orig_tconv
    return out
for tag, wq, aq in (("weights fp16 only", True, False), ("activations fp16 only", False, True), ("both", True, True)):
    got = run(wq, aq)
    e = got.astype(np.float64) - ref
    pr, pg = 1/(1+np.exp(-ref.astype(np.float64))), 1/(1+np.exp(-got.astype(np.float64)))
    lr, lg = tiler_ref.regions_to_labels(pr[0].astype(np.float32)), tiler_ref.regions_to_labels(pg[0].astype(np.float32))
    d = tiler_ref.brats_region_dice(lg, lr)
    print(f"{name} {tag:24s}: logit err max {np.abs(e).max():.4f} rms {np.sqrt((e**2).mean()):.5f} (spread {ref.std():.2f}) mismatches {(lr!=lg).sum()} Dice {d['mean']:.6f}", flush=True)
