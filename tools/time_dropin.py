#!/usr/bin/env python3
"""End-to-end wall time of the drop-in script on one full-size synthetic case (file boundary included):
writes 4 NIfTI modalities (240x240x155, int16, gzip) + synthetic two-model checkpoints, runs
run_brats2021_inference_singlethread.py as run_full_pipeline.py:162-182 does, and reports the stages.

    python tools/time_dropin.py [--folds 1] [--dtype f32|f16] [--worker] [--workdir /tmp/dropin_timing]

--worker: a resident worker (python -m brats_amd.worker) is started first with both models preloaded; the timed runs are
then the thin client (what a pipeline that processes many cases pays per case).
"""
import argparse
import json
import os
import socket
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
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--runs", type=int, default=2)
    args = ap.parse_args()
    import brats_amd as amd
    work = Path(args.workdir)
    results = work / f"nnUNet_results_{args.folds}folds"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans((128, 128, 128))
    t0 = time.perf_counter()
    for name, preset, seed in ((amd.driver.MODEL1, "A", 7), (amd.driver.MODEL2, "B", 8)):
        if not (base / name).exists():
            sds = [amd.synthetic.make_model(preset, seed=seed + k)[0] for k in range(args.folds)]
            amd.checkpoint.save_model_folder(base / name, name.split("__")[0], sds, plans)
    case = "BraTS-GLI-00003-000"
    case_dir = work / case
    if not (case_dir / f"{case}_flair.nii.gz").exists():
        case_dir.mkdir(parents=True, exist_ok=True)
        vol = amd.synthetic.make_volume(seed=1000)
        like = amd.nifti.make_header(vol.shape[1:][::-1], zooms=(1.0, 1.0, 1.0), origin=(0.0, -239.0, 0.0))
        for c, mod in enumerate(("t1", "t1ce", "t2", "flair")):
            amd.nifti.save_like(case_dir / f"{case}_{mod}.nii.gz", np.ascontiguousarray(np.round(vol[c]).astype(np.int16).transpose(2, 1, 0)), like)
    t_setup = time.perf_counter() - t0
    out = work / "results" / case
    folds = [str(k) for k in range(args.folds)]
    cmd = [sys.executable, str(ROOT / "run_brats2021_inference_singlethread.py"), "--input", str(case_dir), "--output", str(out),
           "--results_folder", str(results), "--folds", *folds, "--dtype", args.dtype]
    env = dict(os.environ)
    worker = None
    if args.worker:
        sock = str(work / "worker.sock")
        env["MI355_WORKER_SOCKET"] = sock
        env["PYTHONPATH"] = str(ROOT) + os.pathsep + env.get("PYTHONPATH", "")
        tw = time.perf_counter()
        worker = subprocess.Popen([sys.executable, "-m", "brats_amd.worker", "--socket", sock, "--results_folder", str(results), "--folds", *folds,
                                   "--dtype", args.dtype], env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = ""
        for _ in range(50):  # (the ROCm runtime may print warnings of its own first)
            line = worker.stdout.readline()
            if "listening" in line or not line:
                break
        print(f"worker up in {time.perf_counter() - tw:.2f} s: {line.strip()}")
    else:
        env["MI355_NO_WORKER"] = "1"
    walls = []
    try:
        for trial in range(args.runs):  # second run: page cache warm (and, with --worker, the arena grown), as in a pipeline that processes many cases
            t1 = time.perf_counter()
            res = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t1
            if res.returncode != 0:
                print(res.stdout[-3000:], res.stderr[-3000:])
                return 1
            walls.append(wall)
            stage = [l for l in res.stdout.splitlines() if "[OK] Completed" in l or "Loaded" in l]
            print(f"run {trial}: wall {wall:.2f} s (setup of synthetic inputs {t_setup:.1f} s, not counted)")
            for l in stage:
                print("   ", l.strip())
    finally:
        if worker is not None:
            try:
                with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as sk:
                    sk.connect(env["MI355_WORKER_SOCKET"])
                    sk.sendall((json.dumps({"cmd": "shutdown"}) + "\n").encode())
                    sk.recv(100)
            except OSError:
                pass
            try:
                worker.wait(timeout=60)
            except subprocess.TimeoutExpired:
                worker.kill()
    seg = amd.nifti.load(out / f"{case}.nii.gz")
    print("labels:", {int(k): int(v) for k, v in zip(*np.unique(seg.data, return_counts=True))})
    print(json.dumps({"tool": "time_dropin", "folds": args.folds, "dtype": args.dtype, "worker": bool(args.worker), "wall_s": [round(w, 3) for w in walls],
                      "nominal_reference_s": 300.0, "speedup_vs_nominal_5min": round(300.0 / min(walls), 1)}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
