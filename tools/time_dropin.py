#!/usr/bin/env python3
"""End-to-end wall time of the drop-in script on one full-size synthetic case (file boundary included):
writes 4 NIfTI modalities (240x240x155, int16, gzip) + synthetic two-model checkpoints, runs
run_brats2021_inference_singlethread.py as run_full_pipeline.py:162-182 does, and reports the stages.

    python tools/time_dropin.py [--folds 1] [--workdir /tmp/dropin_timing]
"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folds", type=int, default=1)
    ap.add_argument("--workdir", default="/tmp/dropin_timing")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f16"))
    args = ap.parse_args()
    import brats_amd as amd
    work = Path(args.workdir)
    results = work / "nnUNet_results"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans((128, 128, 128))
    t0 = time.perf_counter()
    for name, preset, seed in ((amd.driver.MODEL1, "A", 7), (amd.driver.MODEL2, "B", 8)):
        if not (base / name).exists():
            sds = [amd.synthetic.make_model(preset, seed=seed + k)[0] for k in range(args.folds)]
            amd.checkpoint.save_model_folder(base / name, name.split("__")[0], sds, plans)
    case = "BraTS-GLI-00003-000"
    case_dir = work / case
    case_dir.mkdir(parents=True, exist_ok=True)
    vol = amd.synthetic.make_volume(seed=1000)
    like = amd.nifti.make_header(vol.shape[1:][::-1], zooms=(1.0, 1.0, 1.0), origin=(0.0, -239.0, 0.0))
    for c, mod in enumerate(("t1", "t1ce", "t2", "flair")):
        amd.nifti.save_like(case_dir / f"{case}_{mod}.nii.gz", np.ascontiguousarray(np.round(vol[c]).astype(np.int16).transpose(2, 1, 0)), like)
    t_setup = time.perf_counter() - t0
    out = work / "results" / case
    cmd = [sys.executable, str(ROOT / "run_brats2021_inference_singlethread.py"), "--input", str(case_dir), "--output", str(out),
           "--results_folder", str(results), "--folds", *[str(k) for k in range(args.folds)], "--dtype", args.dtype]
    for trial in range(2):  # second run: page cache warm, as in a pipeline that processes many cases
        t1 = time.perf_counter()
        res = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True)
        wall = time.perf_counter() - t1
        if res.returncode != 0:
            print(res.stdout[-3000:], res.stderr[-3000:])
            return 1
        stage = [l for l in res.stdout.splitlines() if "[OK] Completed" in l or "Loaded" in l]
        print(f"run {trial}: wall {wall:.2f} s (setup of synthetic inputs {t_setup:.1f} s, not counted)")
        for l in stage:
            print("   ", l.strip())
    seg = amd.nifti.load(out / f"{case}.nii.gz")
    print("labels:", dict(zip(*np.unique(seg.data, return_counts=True))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
