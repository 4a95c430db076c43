#!/usr/bin/env python3
"""fp16 parity margin on the REALISTIC tile: the first 128^3 tile of bench.py's timed synthetic brain
volume (synthetic.make_volume(1000)), models A (seed 7) and B (seed 8), fp16 GPU forward against the CPU oracle forward.
Prints max / rms logit error, max probability error, label mismatches and the WT/TC/ET Dice.

    python tests/diagnostics/f16_margin.py                  # oracle forwards (cached under gpurun_out/) + the default fp16 path
    MI355_FUSE_NORM=0 python tests/diagnostics/f16_margin.py --tag separate_norm      # a variant (the switches are read once per process)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import brats_amd as amd  # noqa: E402
from brats_amd import synthetic, preprocessing, ops  # noqa: E402

PATCH = (128, 128, 128)


def bench_tile(device):
    raw = synthetic.make_volume(seed=1000)
    data, _ = preprocessing.preprocess_case(raw, device)
    steps = [ops.compute_steps(PATCH[a], max(PATCH[a], data.shape[1 + a]), 0.5) for a in range(3)]
    z0, y0, x0 = steps[0][0], steps[1][0], steps[2][0]
    tile = data[:, z0:z0 + 128, y0:y0 + 128, x0:x0 + 128]
    pad = [128 - tile.shape[1 + i] for i in range(3)]
    return torch.nn.functional.pad(tile, (0, pad[2], 0, pad[1], 0, pad[0]))[None].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--cache", default=os.path.join(ROOT, "gpurun_out", "f16_margin_cache"))
    args = ap.parse_args()
    from oracle import unet_ref, tiler_ref
    dev = torch.device("cuda", 0)
    tile = bench_tile(dev)
    os.makedirs(args.cache, exist_ok=True)
    for name, seed in (("A", 7), ("B", 8)):
        sd, meta = synthetic.make_model(name, seed=seed)
        cache = os.path.join(args.cache, f"oracle_{name}.npy")
        if os.path.exists(cache):
            ref = np.load(cache)
        else:
            t0 = time.perf_counter()
            ref = unet_ref.unet_forward(sd, tile.cpu(), unet_ref.default_cfg(norm=meta["norm"], num_groups=meta["num_groups"])).numpy()
            np.save(cache, ref)
            print(f"oracle {name}: {time.perf_counter() - t0:.1f} s", flush=True)
        net = amd.UNet(sd, norm=meta["norm"], num_groups=meta["num_groups"], dtype=args.dtype)
        got = net(tile).cpu().numpy()
        net.close()
        err = got.astype(np.float64) - ref.astype(np.float64)
        pr, pg = 1 / (1 + np.exp(-ref.astype(np.float64))), 1 / (1 + np.exp(-got.astype(np.float64)))
        lr, lg = tiler_ref.regions_to_labels(pr[0].astype(np.float32)), tiler_ref.regions_to_labels(pg[0].astype(np.float32))
        d = tiler_ref.brats_region_dice(lg, lr)
        near = int((np.abs(ref) < 0.05).sum())
        print(f"MARGIN {args.tag:16s} model {name} {args.dtype}: logit err max {np.abs(err).max():.4f} rms {np.sqrt((err ** 2).mean()):.5f} "
              f"(spread {ref.std():.2f})  prob err max {np.abs(pg - pr).max():.4f}  label mismatches {int((lr != lg).sum())} of {lr.size} "
              f"({near} logits within 0.05 of the threshold)  Dice WT/TC/ET {d['WT']:.6f} {d['TC']:.6f} {d['ET']:.6f} mean {d['mean']:.6f}", flush=True)


if __name__ == "__main__":
    main()
