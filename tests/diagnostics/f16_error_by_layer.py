#!/usr/bin/env python3
"""CPU emulation as f16_error_sources.py, split by WHERE the rounding happens: fp16 weights / fp16 stored activations of one
group of layers at a time (encoder level l, decoder stage u, transposed convs), everything else exact.  Answers "which layers
make the fp16 path's logit error" (VERDICT r2 item 3).      python tests/diagnostics/f16_error_by_layer.py B 8"""
import sys, os, re, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import brats_amd as amd
from oracle import unet_ref
torch.set_num_threads(8)
name, seed = sys.argv[1], int(sys.argv[2])
sd, meta = amd.synthetic.make_model(name, seed=seed)
cfg = unet_ref.default_cfg(norm=meta["norm"], num_groups=meta["num_groups"])
vol = amd.synthetic.make_volume(seed=1000)
m = (vol != 0).any(0)
x = vol.astype(np.float32).copy()
for c in range(4):
    v = x[c][m]; x[c][m] = (v - v.mean()) / (v.std() + 1e-8); x[c][~m] = 0
zz, yy, xx = [s // 2 for s in x.shape[1:]]
tile = x[None, :, zz-60:zz+4, yy-70:yy-6, xx-32:xx+32].copy()
ref = unet_ref.unet_forward(sd, tile, cfg).numpy().astype(np.float64)
h = lambda t: t.to(torch.float16).to(torch.float32)
def group_of(key):
    mo = re.match(r"conv_blocks_context\.(\d+)\.", key)
    if mo: return f"enc{mo.group(1)}"
    mo = re.match(r"conv_blocks_localization\.(\d+)\.", key)
    if mo: return f"dec{mo.group(1)}"
    mo = re.match(r"tu\.(\d+)\.", key)
    if mo: return f"tu{mo.group(1)}"
    return "head"
groups = sorted({group_of(k) for k in sd if k.endswith("conv.weight") or k.startswith("tu.")})
def run(wgroup=None, agroup=None):
    sd2 = {k: (h(torch.from_numpy(np.asarray(v))).numpy() if ((k.endswith("conv.weight") or k.startswith("tu.")) and (wgroup == "all" or group_of(k) == wgroup)) else v) for k, v in sd.items()}
    orig_block, orig_tconv = unet_ref._block, F.conv_transpose3d
    def blk(x_, sd_, prefix, stride, cfg_):
        y = orig_block(x_, sd_, prefix, stride, cfg_)
        return h(y) if (agroup == "all" or (agroup and group_of(prefix + ".") == agroup)) else y
    def tc(x_, w_, b_, stride=2):
        y = orig_tconv(x_, w_, b_, stride=stride)
        return h(y) if agroup in ("all", "tu") else y
    unet_ref._block = blk; F.conv_transpose3d = tc
    try:
        return unet_ref.unet_forward(sd2, tile, cfg).numpy().astype(np.float64)
    finally:
        unet_ref._block = orig_block; F.conv_transpose3d = orig_tconv
def rms(a): return float(np.sqrt(((a - ref) ** 2).mean()))
print(f"model {name} seed {seed}: logit spread {ref.std():.2f}; rms logit error when ONLY this group is rounded to fp16")
print(f"  all weights {rms(run('all', None)):.5f}   all stored activations {rms(run(None, 'all')):.5f}")
tot_w = tot_a = 0.0
for g in groups:
    ew = rms(run(g, None)); ea = rms(run(None, g if not g.startswith('tu') else 'tu')) if not g.startswith("tu") or g == "tu0" else float("nan")
    tot_w += ew ** 2; tot_a += 0 if ea != ea else ea ** 2
    print(f"  {g:6s} weights {ew:.5f}   activations {ea:.5f}", flush=True)
print(f"  root of the sum of squares: weights {tot_w ** 0.5:.5f}   activations {tot_a ** 0.5:.5f}  (activations of 'tu0' = all transposed-conv outputs)")
