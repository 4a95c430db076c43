"""Drop-in mirror of the reference's hot-path driver, ``run_brats2021_inference_singlethread.py``.

Function names, argument meaning, on-disk products and exit behaviour follow the reference file
line for line (cited per function); the compute underneath is the HIP library.  The repo-root
script of the same name calls ``main()`` here, so ``run_full_pipeline.py:162-168`` works unchanged.
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import time
from pathlib import Path
from typing import List, Sequence

import numpy as np

from . import checkpoint, nifti, ops, predictor, preprocessing
from .network import UNet

#: the two ensemble members (reference :263-264)
MODEL1 = "nnUNetTrainerV2BraTSRegions_DA4_BN_BD__nnUNetPlansv2.1"
MODEL2 = "nnUNetTrainerV2BraTSRegions_DA4_BN_BD_largeUnet_Groupnorm__nnUNetPlansv2.1"
MODALITY_MAP = {"t1": "0000", "t1ce": "0001", "t2": "0002", "flair": "0003"}  # reference :48-53


def prepare_input(sample_dir, output_dir, quiet=False):
    """Reference :26-78.  BraTS names -> nnU-Net names, copied into ``output_dir``; a case with a
    missing modality is skipped with a warning.  Returns [(case, [4 paths in channel order])].
    (Cases are returned sorted; the reference iterates a Python set, i.e. in arbitrary order.)"""
    sample_dir, output_dir = Path(sample_dir), Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    cases = set()
    for file in sample_dir.glob("*.nii.gz"):
        parts = file.stem.replace(".nii", "").split("_")
        if parts[-1] in ["t1", "t1ce", "t2", "flair", "seg"]:
            cases.add("_".join(parts[:-1]))
    if not quiet:
        print(f"Found {len(cases)} cases: {cases}")
    prepared = []
    for case in sorted(cases):
        files, ok = [], True
        for mod, idx in MODALITY_MAP.items():
            src = sample_dir / f"{case}_{mod}.nii.gz"
            dst = output_dir / f"{case}_{idx}.nii.gz"
            if src.exists():
                if not dst.exists():
                    shutil.copy(src, dst)
                files.append(str(dst))
            else:
                if not quiet:
                    print(f"[WARNING] Missing {mod} for {case}")
                ok = False
                break
        if ok:
            prepared.append((case, files))
    return prepared


class LoadedModel:
    """trainer + params of the reference (:178-183): one device network per fold."""

    def __init__(self, folder: checkpoint.ModelFolder, dtype: str = "f32"):
        self.folder = folder
        self.nets: List[UNet] = [UNet(sd, norm=folder.norm, num_groups=folder.num_groups, dtype=dtype)
                                 for sd in folder.fold_state_dicts]
        self.patch_size = folder.patch_size
        self.nonlin = "sigmoid" if folder.regions else "softmax"

    def close(self):
        for n in self.nets:
            n.close()


class ModelCache:
    """Loaded models keyed by (folder, folds, dtype).  The drop-in process uses one per run; the resident worker
    (``brats_amd.worker``) keeps one for its lifetime, so a request finds both ensemble members on the device."""

    def __init__(self):
        self._models = {}

    @staticmethod
    def _stamp(model_folder, folds):
        """(mtime_ns, size) of every checkpoint file behind the key: a checkpoint replaced on disk must not be served from
        stale device weights by a long-lived worker (the per-process path re-reads the files every run)."""
        out = []
        for f in folds:
            for name in ("model_final_checkpoint.model", "model_final_checkpoint.model.pkl"):
                try:
                    st = os.stat(os.path.join(str(model_folder), f"fold_{int(f)}", name))
                    out.append((st.st_mtime_ns, st.st_size))
                except OSError:
                    out.append(None)
        return tuple(out)

    def get(self, model_folder, folds, dtype="f32") -> "LoadedModel":
        base = (str(Path(model_folder).resolve()), tuple(int(f) for f in folds), dtype)
        key = base + (self._stamp(model_folder, folds),)
        for old in [k for k in self._models if k[:3] == base and k != key]:
            self._models.pop(old).close()   # the files changed: drop the stale device copy
        if key not in self._models:
            self._models[key] = LoadedModel(checkpoint.load_model_folder(model_folder, folds, "model_final_checkpoint"), dtype=dtype)
        return self._models[key]

    def close(self):
        for m in self._models.values():
            m.close()
        self._models.clear()


def read_case(list_of_files: Sequence[str]):
    """The four modality files -> float32 [4, Z, Y, X] (SimpleITK axis order) + the first image (geometry)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(list_of_files)) as pool:  # gunzip releases the GIL: 4 files in parallel
        imgs = list(pool.map(nifti.load, list_of_files))
    shapes = {im.data.shape for im in imgs}
    if len(shapes) != 1:
        raise ValueError(f"modalities have different shapes: {shapes}")
    return np.stack([im.as_zyx().astype(np.float32) for im in imgs]), imgs[0]


def predict_case_single_threaded(model: LoadedModel, list_of_files, output_file, do_tta=True, mixed_precision=True,
                                 step_size=0.5, all_in_gpu=True, save_npz=False, timings=None):
    """Reference :81-158: preprocess once, predict with every fold, average, export with
    region_class_order=(1,2,3).  Returns (output_file, mean probabilities on the device)."""
    import torch
    t0 = time.perf_counter()
    print(f"Preprocessing {output_file}")
    raw, like = read_case(list_of_files)
    # the plans decide what preprocessing means: anything this path does not implement is refused here
    data, props = preprocessing.preprocess_case(raw, plans=model.folder.plans, spacing_zyx=tuple(reversed(like.zooms)))
    print(f"Data shape after preprocessing: {tuple(data.shape)}")
    print(f"Predicting {output_file}")
    t1 = time.perf_counter()
    probs = predictor.predict_folds(model.nets, data, model.patch_size, step_size, do_tta, (0, 1, 2), True,
                                    model.nonlin)
    print(f"Ensembling {len(model.nets)} folds")
    # save_segmentation_nifti_from_softmax resamples the probabilities back to the shape after cropping (order 1; driver :131-138):
    # a no-op unless preprocessing resampled (never for BraTS, 1 mm -> 1 mm)
    probs = preprocessing.resample_probabilities_for_export(probs, props)
    lo = [b[0] for b in props["crop_bbox"]]
    seg = ops.regions_to_labels(probs, (1, 2, 3) if model.folder.regions else None, lo, props["original_size_of_raw_data"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"Saving segmentation to {output_file}")
    seg_zyx = seg.cpu().numpy()
    nifti.save_like(output_file, np.ascontiguousarray(seg_zyx.transpose(2, 1, 0)), like)
    if save_npz:
        np.savez_compressed(str(output_file).replace(".nii.gz", ".npz"), softmax=probs.cpu().numpy(),
                            crop_bbox=np.array(props["crop_bbox"]))
    if timings is not None:
        timings.append(dict(read_preprocess_s=t1 - t0, predict_s=t2 - t1, export_s=time.perf_counter() - t2))
    return output_file, probs, props


def run_model_single_threaded(model_folder, input_folder, output_folder, folds=(0, 1, 2, 3, 4), do_tta=True,
                              step_size=0.5, save_npz=False, dtype="f32"):
    """Reference :161-214."""
    model_folder, input_folder, output_folder = Path(model_folder), Path(input_folder), Path(output_folder)
    if not model_folder.exists():
        print(f"[ERROR] Model not found: {model_folder}")
        sys.exit(1)
    print(f"Model path: {model_folder}")
    print(f"Loading model with folds: {folds}")
    model = LoadedModel(checkpoint.load_model_folder(model_folder, folds, "model_final_checkpoint"), dtype=dtype)
    print(f"Loaded {len(model.nets)} fold checkpoints")
    prepared = prepare_input(input_folder, output_folder / "temp_input")
    if not prepared:
        print("[ERROR] No valid cases found!")
        model.close()
        return []
    outputs = []
    for case_name, case_files in prepared:
        output_file = output_folder / f"{case_name}.nii.gz"
        output_folder.mkdir(parents=True, exist_ok=True)
        print(f"\n{'=' * 70}\nProcessing case: {case_name}\n{'=' * 70}")
        timings = []
        predict_case_single_threaded(model, case_files, str(output_file), do_tta=do_tta, step_size=step_size,
                                     save_npz=save_npz, timings=timings)
        t = timings[0]
        print(f"[OK] Completed: {output_file}  (read+preprocess {t['read_preprocess_s']:.2f} s, "
              f"predict {t['predict_s']:.2f} s, export {t['export_s']:.2f} s)")
        outputs.append(output_file)
    model.close()
    return outputs


def volumes_of(seg, zooms):
    """Reference :217-243 on an array - including its use of BraTS label 4 for ET on a map that holds nnU-Net label 3
    (so ET prints 0.00 before convert_labels_to_brats.py has run)."""
    voxel_volume_cm3 = float(np.prod(zooms)) / 1000.0
    ncr, ed, et = int(np.sum(seg == 1)), int(np.sum(seg == 2)), int(np.sum(seg == 4))
    return {"NCR": ncr * voxel_volume_cm3, "ED": ed * voxel_volume_cm3, "ET": et * voxel_volume_cm3,
            "TC": (ncr + et) * voxel_volume_cm3, "WT": (ncr + ed + et) * voxel_volume_cm3}


def calculate_volumes(seg_path):
    """Reference :217-243 (reads the saved file, as the reference does)."""
    img = nifti.load(seg_path)
    return volumes_of(img.data, img.zooms)


def ensemble_label_files(model1_output: Path, model2_output: Path, output_folder: Path, label_format: str = "nnunet"):
    """Reference :286-322: np.round((seg1 + seg2) / 2) per voxel, saved with seg1's header.  `label_format` other
    than "nnunet" folds convert_labels_to_brats.py:34-55 into the export (the label map is still on the device)."""
    import torch
    from . import evaluate
    finals = []
    for seg1_path in sorted(model1_output.glob("*.nii.gz")):
        case_name = seg1_path.stem.replace(".nii", "")
        seg2_path = model2_output / seg1_path.name
        if not seg2_path.exists():
            print(f"[WARNING] Missing model2 prediction for {case_name}")
            continue
        print(f"Ensembling {case_name}")
        img1, img2 = nifti.load(seg1_path), nifti.load(seg2_path)
        a = torch.from_numpy(np.ascontiguousarray(img1.data.astype(np.uint8))).cuda()
        b = torch.from_numpy(np.ascontiguousarray(img2.data.astype(np.uint8))).cuda()
        ens = evaluate.convert_labels(ops.label_ensemble(a, b), label_format).cpu().numpy()
        final_output = output_folder / f"{case_name}.nii.gz"
        nifti.save_like(final_output, ens, img1)
        print(f"[OK] Saved: {final_output}")
        v = calculate_volumes(final_output)
        print(f"\nTumor Volume Analysis for {case_name}:")
        print(f"  NCR (Necrotic Core):        {v['NCR']:.2f} cm3")
        print(f"  ED (Peritumoral Edema):     {v['ED']:.2f} cm3")
        print(f"  ET (Enhancing Tumor):       {v['ET']:.2f} cm3")
        print(f"  TC (Tumor Core):            {v['TC']:.2f} cm3")
        print(f"  WT (Whole Tumor):           {v['WT']:.2f} cm3")
        finals.append(final_output)
    return finals


def run_both_models(model_dirs, input_folder, output_folder, folds, do_tta, step_size, dtype, label_format, cache: ModelCache):
    """What reference :263-322 produces (model 1 over all cases, model 2 over all cases, label-round ensemble of the two
    NIfTI files per case), computed in ONE pass per case: the four modalities are read, cropped and normalised once
    (the reference preprocesses per model, :89), both members predict from the same device tensor, the ensemble takes
    the two label volumes where they are - on the device - and the three NIfTI products of the case
    (``temp_model1/<case>.nii.gz``, ``temp_model2/<case>.nii.gz``, ``<case>.nii.gz``) are compressed and written by
    background threads while the next case runs.  Same files, same label values, same printed volumes; checkpoint loading of
    both members and the copy of the inputs into ``temp_model{1,2}/temp_input`` overlap as well."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from . import evaluate
    output_folder = Path(output_folder)
    outs = [output_folder / f"temp_model{i}" for i in (1, 2)]
    for md in model_dirs:
        if not Path(md).exists():
            print(f"[ERROR] Model not found: {md}")
            sys.exit(1)
    pool = ThreadPoolExecutor(max_workers=6)
    t0 = time.perf_counter()
    print(f"Loading models with folds: {tuple(folds)}")
    dev = torch.cuda.current_device()

    def load(md):  # (a new thread starts on device 0: the library is bound to THIS process's device)
        torch.cuda.set_device(dev)
        return cache.get(md, folds, dtype)
    loaders = [pool.submit(load, md) for md in model_dirs]  # torch.load and the weight packing release the GIL
    prepared = prepare_input(input_folder, outs[0] / "temp_input")
    mirror = pool.submit(prepare_input, input_folder, outs[1] / "temp_input", True)
    models = [f.result() for f in loaders]
    print(f"Loaded {sum(len(m.nets) for m in models)} fold checkpoints of {len(models)} models in {time.perf_counter() - t0:.2f} s")
    if not prepared:
        print("[ERROR] No valid cases found!")
        mirror.result()
        return []
    for o in outs:
        o.mkdir(parents=True, exist_ok=True)
    writes, finals = [], []
    # Two streams (the one documented exception to "one stream at a time"): the NEXT case's host-to-device
    # copy, crop mask and z-score run on `side` while the current case's forwards run on the main stream.  Preprocessing
    # and prediction use disjoint scratch slots of the library (crop / z-score vs arena / aggregation), and the hand-off is
    # an event: the main stream waits for the case's preprocessing before its first tile gather.
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def preprocess_on_side(raw, like, plans):
        with torch.cuda.stream(side):
            data, props = preprocessing.preprocess_case(raw, plans=plans, spacing_zyx=tuple(reversed(like.zooms)))
            ev = torch.cuda.Event()
            ev.record(side)
        data.record_stream(main_stream)
        return data, props, ev

    plans0 = models[0].folder.plans
    nxt = pool.submit(read_case, prepared[0][1])
    raw, like = nxt.result()
    ready = preprocess_on_side(raw, like, plans0)
    for ci, (case_name, case_files) in enumerate(prepared):
        print(f"\n{'=' * 70}\nProcessing case: {case_name}\n{'=' * 70}")
        t1 = time.perf_counter()
        if ci + 1 < len(prepared):
            nxt = pool.submit(read_case, prepared[ci + 1][1])  # the next case's gunzip overlaps this case's prediction
        data, props, ev = ready
        cur_raw_channels, cur_like = raw.shape[0], like
        main_stream.wait_event(ev)
        print(f"Data shape after preprocessing: {tuple(data.shape)}")
        t2 = time.perf_counter()
        segs = [None] * len(models)
        # (the larger member first: the process-wide activation arena is then allocated once at its final size instead of
        #  being freed and re-allocated when the second member turns out to need more - a 20-50 GB hipMalloc each time)
        order = sorted(range(len(models)), key=lambda i: -sum(int(np.prod(v.shape)) for v in models[i].folder.fold_state_dicts[0].values()))
        for mi in order:
            model = models[mi]
            if model.folder.plans is not plans0:
                # both members predict from ONE preprocessed tensor: their plans must ask for the same preprocessing of this case
                preprocessing.check_plans(model.folder.plans, cur_raw_channels)
                zyx = tuple(reversed(cur_like.zooms))
                if preprocessing.check_spacing(model.folder.plans, zyx, props["size_after_cropping"]) != \
                        preprocessing.check_spacing(plans0, zyx, props["size_after_cropping"]):
                    raise preprocessing.UnsupportedPlansError("the two ensemble members' plans resample this case to different grids")
            print(f"Predicting {case_name} with model {mi + 1} ({len(model.nets)} folds)")
            probs = predictor.predict_folds(model.nets, data, model.patch_size, step_size, do_tta, (0, 1, 2), True, model.nonlin)
            probs = preprocessing.resample_probabilities_for_export(probs, props)   # (a no-op unless preprocessing resampled)
            lo = [b[0] for b in props["crop_bbox"]]
            segs[mi] = ops.regions_to_labels(probs, (1, 2, 3) if model.folder.regions else None, lo, props["original_size_of_raw_data"])
        ens = evaluate.convert_labels(ops.label_ensemble(segs[0], segs[1]), label_format)
        if ci + 1 < len(prepared):  # the forwards above are only enqueued: preprocess the next case beside them
            raw, like = nxt.result()
            ready = preprocess_on_side(raw, like, plans0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        host = [t.cpu().numpy() for t in (segs[0], segs[1], ens)]
        final_output = output_folder / f"{case_name}.nii.gz"
        for arr, path in zip(host, (outs[0] / f"{case_name}.nii.gz", outs[1] / f"{case_name}.nii.gz", final_output)):
            # label volumes are written (x, y, z); the reference's ensemble saves with seg1's header = the input geometry
            writes.append(pool.submit(nifti.save_like, path, np.ascontiguousarray(arr.transpose(2, 1, 0)), cur_like))
        v = volumes_of(host[2], cur_like.zooms)
        print(f"[OK] Completed: {final_output}  (wait for input {t2 - t1:.2f} s, predict {t3 - t2:.2f} s)")
        print(f"\nTumor Volume Analysis for {case_name}:")
        print(f"  NCR (Necrotic Core):        {v['NCR']:.2f} cm3")
        print(f"  ED (Peritumoral Edema):     {v['ED']:.2f} cm3")
        print(f"  ET (Enhancing Tumor):       {v['ET']:.2f} cm3")
        print(f"  TC (Tumor Core):            {v['TC']:.2f} cm3")
        print(f"  WT (Whole Tumor):           {v['WT']:.2f} cm3")
        finals.append(final_output)
    for w in writes:
        w.result()
    mirror.result()
    pool.shutdown()
    return finals


def main(argv=None, script_dir=None, model_cache: ModelCache = None):
    """Reference :246-327.  --input/--output are the whole contract; the extra flags default to the
    reference's hard-coded settings (5 folds, TTA on, step 0.5).  ``model_cache``: the resident worker's (models stay
    loaded between requests); a plain process builds its own and frees it at the end."""
    ap = argparse.ArgumentParser(description="BraTS 2021 Brain Tumor Segmentation (MI355X-native)")
    ap.add_argument("--input", type=str, required=True, help="Input directory with BraTS sample data")
    ap.add_argument("--output", type=str, required=True, help="Output directory for segmentation results")
    ap.add_argument("--results_folder", type=str, default=None, help="default: <script dir>/nnUNet_results")
    ap.add_argument("--folds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--disable_tta", action="store_true")
    ap.add_argument("--step_size", type=float, default=0.5)
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="f32 (default): what the reference computes on its CPU path; f16: fp16 storage / fp32 accumulation, "
                         "the autocast setting the upstream trainer uses on a GPU (mixed_precision=True)")
    ap.add_argument("--sequential", action="store_true",
                    help="the reference's order of work (model 1 over all cases, model 2, then the ensemble of the two files) "
                         "instead of one pass per case; same products")
    ap.add_argument("--label-format", dest="label_format", choices=("nnunet", "brats2025", "brats2021"), default="nnunet",
                    help="convention of the final <case>.nii.gz; 'nnunet' (default) is what the reference writes, the others "
                         "fold convert_labels_to_brats.py:34-55 into the export")
    args = ap.parse_args(argv)
    script_dir = Path(script_dir) if script_dir else Path(__file__).resolve().parent.parent
    results_folder = Path(args.results_folder or os.environ.get("MI355_RESULTS_FOLDER") or script_dir / "nnUNet_results")
    os.environ["RESULTS_FOLDER"] = str(results_folder)
    print("=" * 70 + "\nBraTS 2021 TUMOR SEGMENTATION (MI355X-native HIP path)\n" + "=" * 70)
    print(f"RESULTS_FOLDER: {results_folder}\n")
    base = results_folder / "3d_fullres" / "Task500_BraTS2021"
    output_folder = Path(args.output)
    if args.sequential:  # the reference's own order of work: model 1 over all cases, model 2, then the file-level ensemble
        outs = []
        for i, name in enumerate((MODEL1, MODEL2), 1):
            print("\n" + "=" * 70 + f"\nMODEL {i}: {name.split('__')[0]}\n" + "=" * 70)
            outs.append(output_folder / f"temp_model{i}")
            run_model_single_threaded(base / name, args.input, outs[-1], tuple(args.folds), not args.disable_tta,
                                      args.step_size, dtype=args.dtype)
        print("\n" + "=" * 70 + "\nENSEMBLING MODEL PREDICTIONS\n" + "=" * 70)
        ensemble_label_files(outs[0], outs[1], output_folder, args.label_format)
    else:
        print("\n" + "=" * 70 + f"\nMODELS: {MODEL1.split('__')[0]} + {MODEL2.split('__')[0]}\n" + "=" * 70)
        cache = model_cache or ModelCache()
        try:
            run_both_models([base / MODEL1, base / MODEL2], args.input, output_folder, tuple(args.folds), not args.disable_tta,
                            args.step_size, args.dtype, args.label_format, cache)
        finally:
            if model_cache is None:
                cache.close()
    print("\n" + "=" * 70 + "\nSEGMENTATION COMPLETE!\n" + "=" * 70)
    print(f"Results saved to: {output_folder}")
    return 0
