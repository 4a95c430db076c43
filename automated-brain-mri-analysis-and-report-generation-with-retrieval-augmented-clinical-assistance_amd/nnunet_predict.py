"""``nnUNet_predict`` / ``nnUNet_ensemble``-style command line.

Surface per archived/kaist_original_inference.py:30-32 and scripts/run_inference.py:89-97:

    python -m brats_amd.nnunet_predict -i IN -o OUT -t 500 -m 3d_fullres -tr <trainer> [-f 0 1 ..]
           [--save_npz] [--disable_tta] [--step_size 0.5] [-p nnUNetPlansv2.1]
    python -m brats_amd.nnunet_predict --ensemble OUT1 OUT2 -o OUT        (probability mean of the npz files)
    python -m brats_amd.nnunet_predict --postprocess FOLDER -o OUT [--threshold 200 --replace_with 2]
           [--label_format nnunet|brats2025|brats2021]                     (kaist_original_inference.py:33-34)
    python -m brats_amd.nnunet_predict --kaist -i IN -o OUT [-f ...]       (the whole script :26-38 in one process)

IN holds nnU-Net-named files ``<case>_0000..0003.nii.gz``; RESULTS_FOLDER locates the models.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np


def _cases(folder: Path):
    ids = sorted({f.name[:-len("_0000.nii.gz")] for f in folder.glob("*_0000.nii.gz")})
    return [(c, [str(folder / f"{c}_{i:04d}.nii.gz") for i in range(4)]) for c in ids]


def main(argv=None):
    from . import checkpoint, driver, nifti, ops
    ap = argparse.ArgumentParser(prog="nnUNet_predict (MI355X)")
    ap.add_argument("-i", "--input_folder")
    ap.add_argument("-o", "--output_folder", required=True)
    ap.add_argument("-t", "--task_name", default="500")
    ap.add_argument("-m", "--model", default="3d_fullres")
    ap.add_argument("-tr", "--trainer_class_name", default="nnUNetTrainerV2BraTSRegions_DA4_BN_BD")
    ap.add_argument("-p", "--plans_identifier", default="nnUNetPlansv2.1")
    ap.add_argument("-f", "--folds", nargs="+", type=int, default=[0, 1, 2, 3, 4])
    ap.add_argument("--save_npz", action="store_true")
    ap.add_argument("--disable_tta", action="store_true")
    ap.add_argument("--step_size", type=float, default=0.5)
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32", help="f16 = fp16 storage / fp32 accumulation (upstream autocast)")
    ap.add_argument("--ensemble", nargs=2, metavar=("FOLDER1", "FOLDER2"))
    ap.add_argument("--postprocess", metavar="FOLDER", help="apply_threshold_to_folder + label convention on label NIfTIs")
    ap.add_argument("--threshold", type=int, default=200)
    ap.add_argument("--replace_with", type=int, default=2)
    ap.add_argument("--label_format", choices=("nnunet", "brats2025", "brats2021"), default=None,
                    help="default: brats2021 for --postprocess / --kaist (the 2018/2019 convention of the KAIST script)")
    ap.add_argument("--kaist", action="store_true", help="both trainers + probability ensemble + post-processing")
    args = ap.parse_args(argv)
    out = Path(args.output_folder)
    out.mkdir(parents=True, exist_ok=True)
    if args.kaist:
        if not args.input_folder:
            ap.error("-i is required")
        common = ["-t", args.task_name, "-m", args.model, "-p", args.plans_identifier, "--step_size", str(args.step_size), "--dtype", args.dtype,
                  "-f", *[str(k) for k in args.folds]] + (["--disable_tta"] if args.disable_tta else [])
        raw = []
        for k, tr in enumerate((driver.MODEL1, driver.MODEL2), 1):
            raw.append(out / f"raw_output_{k}")
            rc = main(["-i", args.input_folder, "-o", str(raw[-1]), "-tr", tr.split("__")[0], "--save_npz", *common])
            if rc:
                return rc
        rc = main(["--ensemble", str(raw[0]), str(raw[1]), "-o", str(out / "ensemble")])
        if rc:
            return rc
        return main(["--postprocess", str(out / "ensemble"), "-o", str(out), "--threshold", str(args.threshold),
                     "--replace_with", str(args.replace_with), "--label_format", args.label_format or "brats2021"])
    if args.postprocess:
        import torch
        from . import evaluate
        for fpath in sorted(Path(args.postprocess).glob("*.nii.gz")):
            like = nifti.load(fpath)
            seg = torch.from_numpy(np.ascontiguousarray(like.data.astype(np.uint8))).cuda()
            seg, n3 = evaluate.apply_brats_threshold(seg, args.threshold, args.replace_with)
            seg = evaluate.convert_labels(seg, args.label_format or "brats2021")
            nifti.save_like(out / fpath.name, seg.cpu().numpy(), like)
            print(f"[OK] {out / fpath.name} (enhancing voxels: {n3}{', relabelled' if n3 < args.threshold else ''})")
        return 0
    if args.ensemble:
        import torch
        f1, f2 = (Path(p) for p in args.ensemble)
        for npz1 in sorted(f1.glob("*.npz")):
            npz2 = f2 / npz1.name
            if not npz2.exists():
                print(f"[WARNING] {npz2} missing")
                continue
            a, b = np.load(npz1), np.load(npz2)
            mean = ops.prob_mean(torch.from_numpy(a["softmax"]).cuda(), torch.from_numpy(b["softmax"]).cuda())
            like = nifti.load(f1 / npz1.name.replace(".npz", ".nii.gz"))
            lo = [int(v[0]) for v in a["crop_bbox"]]
            seg = ops.regions_to_labels(mean, (1, 2, 3), lo, like.data.shape[::-1]).cpu().numpy()
            nifti.save_like(out / npz1.name.replace(".npz", ".nii.gz"), np.ascontiguousarray(seg.transpose(2, 1, 0)), like)
            print(f"[OK] {out / npz1.name.replace('.npz', '.nii.gz')}")
        return 0
    if not args.input_folder:
        ap.error("-i is required")
    results = Path(os.environ.get("RESULTS_FOLDER", "nnUNet_results"))
    task = args.task_name if args.task_name.startswith("Task") else f"Task{int(args.task_name):03d}_BraTS2021"
    model_dir = results / args.model / task / f"{args.trainer_class_name}__{args.plans_identifier}"
    if not model_dir.exists():
        print(f"[ERROR] Model not found: {model_dir}")
        return 1
    model = driver.LoadedModel(checkpoint.load_model_folder(model_dir, args.folds), dtype=args.dtype)
    for case, files in _cases(Path(args.input_folder)):
        driver.predict_case_single_threaded(model, files, str(out / f"{case}.nii.gz"), do_tta=not args.disable_tta,
                                            step_size=args.step_size, save_npz=args.save_npz)
        print(f"[OK] {case}")
    model.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
