#!/usr/bin/env python3
"""How stable is the fp16 path's margin over the north_star gate (Dice >= 0.999 on all voxels)?  VERDICT r4 weak #2: every
fp16 Dice on file is one seed per case.  Here: S weight-seed sets x V synthetic volumes (the first 128^3 tile of each), models A
and B, the fp16 path against the fp32 path of this library (which sits within 1.5e-4 of the CPU oracle on this tile, bench.py
parity blocks), in the four settings the suite and the bench gate:

    single forward (fold 0, no TTA) | fold 0 with 8-way TTA | 5-fold mean, no TTA | 5-fold mean with 8-way TTA (the reference's
    setting, run_brats2021_inference_singlethread.py:112-128, 161, 208-211)

Prints one line per (model, seed set, volume, setting) and the min / median / max per (model, setting).  GPU only, no oracle
forwards (the oracle import below is only for the label / Dice helpers): about a minute.

    python tests/diagnostics/f16_seed_study.py [--sets 5] [--volumes 2] [--tag NAME]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import brats_amd as amd  # noqa: E402
from brats_amd import synthetic, preprocessing, ops, predictor  # noqa: E402

PATCH = (128, 128, 128)
SETTINGS = (("single", "fold 0, no TTA        "), ("single_tta", "fold 0, 8-way TTA     "), ("mean", "5-fold mean, no TTA   "),
            ("mean_tta", "5-fold mean, 8-way TTA"))


def first_tile(seed, dev):
    data, _ = preprocessing.preprocess_case(synthetic.make_volume(seed=seed), dev)
    steps = [ops.compute_steps(PATCH[a], max(PATCH[a], data.shape[1 + a]), 0.5) for a in range(3)]
    t = data[:, steps[0][0]:steps[0][0] + 128, steps[1][0]:steps[1][0] + 128, steps[2][0]:steps[2][0] + 128]
    pad = [128 - t.shape[1 + i] for i in range(3)]
    return torch.nn.functional.pad(t, (0, pad[2], 0, pad[1], 0, pad[0])).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--models", default="A,B")
    ap.add_argument("--sets", type=int, default=5)
    ap.add_argument("--volumes", type=int, default=2)
    ap.add_argument("--noise", type=int, default=0, help="also that many 128^3 Gaussian-noise tiles (RandomState 2, 3, ...: the tile128 fixture of "
                                                        "tests/test_gpu_network.py is seed 2), listed as volumes 2, 3, ...")
    args = ap.parse_args()
    from oracle import tiler_ref
    dev = torch.device("cuda", 0)
    tiles = [(1000 + v, first_tile(1000 + v, dev)) for v in range(args.volumes)]
    tiles += [(2 + k, torch.from_numpy(np.random.RandomState(2 + k).standard_normal((4, 128, 128, 128)).astype(np.float32)).to(dev)) for k in range(args.noise)]

    def dice(a, b):
        la, lb = tiler_ref.regions_to_labels(a), tiler_ref.regions_to_labels(b)
        return tiler_ref.brats_region_dice(la, lb)["mean"], int((la != lb).sum())

    summary = {}
    for name, seed0 in (("A", 7), ("B", 8)):
        if name not in args.models.split(","):
            continue
        for s in range(args.sets):
            # set 0 = the seeds of the suite and the bench (A 7..11, B 8..12); set s shifts them by 100 s
            sds = [synthetic.make_model(name, seed=seed0 + 100 * s + k) for k in range(5)]
            res = {}
            for dtype in ("f32", "f16"):
                nets = [amd.UNet(sd, norm=m["norm"], num_groups=m["num_groups"], dtype=dtype) for sd, m in sds]
                for v, (_, tile) in enumerate(tiles):
                    run = lambda ns, tta: predictor.predict_folds(ns, tile, PATCH, 0.5, tta, (0, 1, 2), True, "sigmoid").cpu().numpy()  # noqa: E731
                    res[dtype, v, "single"] = run(nets[:1], False)
                    res[dtype, v, "single_tta"] = run(nets[:1], True)
                    res[dtype, v, "mean"] = run(nets, False)
                    res[dtype, v, "mean_tta"] = run(nets, True)
                for n in nets:
                    n.close()
            for v in range(len(tiles)):
                for key, what in SETTINGS:
                    d, nd = dice(res["f16", v, key], res["f32", v, key])
                    perr = float(np.abs(res["f16", v, key] - res["f32", v, key]).max())
                    kind = "brain" if tiles[v][0] >= 1000 else "noise"
                    summary.setdefault((name, kind, key), []).append(d)
                    print(f"SEEDSTUDY {args.tag} {name} set {s} {kind} volume {tiles[v][0]} {what}: Dice {d:.6f} ({nd} labels differ), prob err {perr:.4f}", flush=True)
    for (name, kind, key), ds in summary.items():
        what = dict(SETTINGS)[key]
        ds = np.array(ds)
        print(f"SEEDSTUDY {args.tag} SUMMARY {name} {kind} tiles {what}: n {len(ds)}  min {ds.min():.6f}  median {np.median(ds):.6f}  max {ds.max():.6f}  "
              f"below 0.999: {int((ds < 0.999).sum())}  below 0.9993: {int((ds < 0.9993).sum())}", flush=True)


if __name__ == "__main__":
    main()
