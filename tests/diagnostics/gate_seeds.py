#!/usr/bin/env python3
"""Measurements behind two gates of tests/test_gpu_network.py that round 3 widened after ONE red run each:

  1. test_norm_statistics_of_small_activations: logit error / spread of the fp32 path against the CPU oracle when every conv
     weight is scaled by 2e-3 (pre-norm rms ~1e-3), models A_in and B - here over 3 model seeds x 3 input seeds each;
  2. test_f16_fused_input_norm_agrees_with_the_separate_pass: max logit difference / spread between the fp16 build that applies
     the producer's norm in the consumer's staging (MI355_FUSE_NORM=1) and the one with the separate pass (=0), model B at
     64^3 - here over 4 input seeds (the switch is read once per process: one child process per setting).

    python tests/diagnostics/gate_seeds.py            (GPU box; prints one line per sample and the maxima)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import brats_amd as amd  # noqa: E402


def part1():
    from oracle import unet_ref
    worst = {}
    for name in ("A_in", "B"):
        for mseed in (7, 8, 9):
            sd, meta = amd.synthetic.make_model(name, seed=mseed)
            sd = {k: (v * 2e-3 if k.endswith(".conv.weight") else v) for k, v in sd.items()}
            net = amd.UNet(sd, norm=meta["norm"], num_groups=meta["num_groups"])
            cfg = unet_ref.default_cfg(norm=meta["norm"], num_groups=meta["num_groups"])
            for xseed in (12, 13, 14):
                x = np.random.RandomState(xseed).standard_normal((1, 4, 64, 64, 64)).astype(np.float32)
                ref = unet_ref.unet_forward(sd, x, cfg).numpy()
                got = net(torch.from_numpy(x).cuda()).cpu().numpy()
                spread = float(ref.std())
                rel = float(np.abs(got - ref).max()) / max(spread, 1.0)
                worst[name] = max(worst.get(name, 0.0), rel)
                print(f"GATE1 {name} model seed {mseed} input seed {xseed}: logit err {rel:.3e} x spread ({spread:.2f})", flush=True)
            net.close()
    print("GATE1 maxima: " + ", ".join(f"{k} {v:.3e}" for k, v in worst.items()), flush=True)


CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
import brats_amd
sd, meta = brats_amd.synthetic.make_model("B", seed=7)
net = brats_amd.UNet(sd, norm="group", num_groups=16, dtype="f16")
out = {}
for s in (1, 2, 3, 4):
    x = np.random.RandomState(s).standard_normal((1, 4, 64, 64, 64)).astype(np.float32)
    out[f"y{s}"] = net(torch.from_numpy(x).cuda()).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def part2():
    from oracle import unet_ref
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            path = os.path.join(td, f"y{flag}.npz")
            res = subprocess.run([sys.executable, "-c", CHILD % ROOT, path], env=dict(os.environ, MI355_FUSE_NORM=flag),
                                 capture_output=True, text=True, timeout=900)
            assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
            outs[flag] = dict(np.load(path))
    sd, meta = amd.synthetic.make_model("B", seed=7)
    cfg = unet_ref.default_cfg(norm="group", num_groups=16)
    worst = 0.0
    for s in (1, 2, 3, 4):
        fused, plain = outs["1"][f"y{s}"], outs["0"][f"y{s}"]
        spread = float(plain.std())
        rel = float(np.abs(fused - plain).max()) / spread
        x = np.random.RandomState(s).standard_normal((1, 4, 64, 64, 64)).astype(np.float32)
        ref = unet_ref.unet_forward(sd, x, cfg).numpy()
        ef, ep = float(np.abs(fused - ref).max()) / spread, float(np.abs(plain - ref).max()) / spread
        worst = max(worst, rel)
        print(f"GATE2 input seed {s}: fused vs separate {rel:.3e} x spread ({spread:.2f}); vs oracle: fused {ef:.3e}, separate {ep:.3e}", flush=True)
    print(f"GATE2 maximum: {worst:.3e}", flush=True)


if __name__ == "__main__":
    part1()
    part2()
