#!/usr/bin/env python3
"""fp16 margin of the FOLD MEAN (run_brats2021_inference_singlethread.py:161,:112-128) on the bench's brain tile: for models A
(seeds 7..11) and B (8..12), the 5-fold mean of the fp16 path against the fp32 path of this library (which sits within 1.5e-4
of the CPU oracle on this tile, bench.py parity blocks) - without mirrors and with the reference's 8-way TTA - and, for
comparison, each single fold.  GPU only, no oracle forwards: seconds instead of minutes.

    python tests/diagnostics/fold_mean_margin.py [--tag NAME]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="default")
    ap.add_argument("--models", default="A,B")
    args = ap.parse_args()
    from oracle import tiler_ref
    dev = torch.device("cuda", 0)
    data, _ = preprocessing.preprocess_case(synthetic.make_volume(seed=1000), dev)
    steps = [ops.compute_steps(PATCH[a], max(PATCH[a], data.shape[1 + a]), 0.5) for a in range(3)]
    t = data[:, steps[0][0]:steps[0][0] + 128, steps[1][0]:steps[1][0] + 128, steps[2][0]:steps[2][0] + 128]
    pad = [128 - t.shape[1 + i] for i in range(3)]
    tile = torch.nn.functional.pad(t, (0, pad[2], 0, pad[1], 0, pad[0])).contiguous()

    def dice(a, b):
        la, lb = tiler_ref.regions_to_labels(a), tiler_ref.regions_to_labels(b)
        return tiler_ref.brats_region_dice(la, lb)["mean"], int((la != lb).sum())

    for name, seed0 in (("A", 7), ("B", 8)):
        if name not in args.models.split(","):
            continue
        sds = [synthetic.make_model(name, seed=seed0 + k) for k in range(5)]
        res = {}
        for dtype in ("f32", "f16"):
            nets = [amd.UNet(sd, norm=m["norm"], num_groups=m["num_groups"], dtype=dtype) for sd, m in sds]
            res[dtype, "single"] = [predictor.predict_folds([n], tile, PATCH, 0.5, False, (0, 1, 2), True, "sigmoid").cpu().numpy() for n in nets]
            res[dtype, "mean"] = predictor.predict_folds(nets, tile, PATCH, 0.5, False, (0, 1, 2), True, "sigmoid").cpu().numpy()
            res[dtype, "mean_tta"] = predictor.predict_folds(nets, tile, PATCH, 0.5, True, (0, 1, 2), True, "sigmoid").cpu().numpy()
            res[dtype, "single_tta"] = predictor.predict_folds(nets[:1], tile, PATCH, 0.5, True, (0, 1, 2), True, "sigmoid").cpu().numpy()
            for n in nets:
                n.close()
        for k in range(5):
            d, nd = dice(res["f16", "single"][k], res["f32", "single"][k])
            print(f"FOLDMARGIN {args.tag} {name} fold {k} single, no TTA : Dice {d:.6f} ({nd} labels), prob err {np.abs(res['f16', 'single'][k] - res['f32', 'single'][k]).max():.4f}", flush=True)
        for key, what in (("single_tta", "fold 0, 8-way TTA     "), ("mean", "5-fold mean, no TTA   "), ("mean_tta", "5-fold mean, 8-way TTA")):
            d, nd = dice(res["f16", key], res["f32", key])
            near = int((np.abs(res["f32", key] - 0.5) < 0.0125).sum())
            print(f"FOLDMARGIN {args.tag} {name} {what}: Dice {d:.6f} ({nd} labels differ; {near} probabilities within 0.0125 of the threshold), "
                  f"prob err {np.abs(res['f16', key] - res['f32', key]).max():.4f}", flush=True)


if __name__ == "__main__":
    main()
