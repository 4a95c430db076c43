"""-m gpu: the drop-in boundary end to end - the repo-root ``run_brats2021_inference_singlethread.py``
invoked exactly as run_full_pipeline.py:162-182 does (subprocess, --input/--output), on a synthetic
BraTS-named case with synthetic two-model / multi-fold checkpoints, compared with the oracle's
restatement of the whole driver (preprocess -> folds -> mean -> regions -> label-round ensemble)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import driver_ref, tiler_ref, unet_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_case(amd, folder, case, shape_zyx, seed, zooms=(1.0, 1.0, 1.0)):
    vol = amd.synthetic.make_volume(seed=seed, shape=shape_zyx)
    like = amd.nifti.make_header(shape_zyx[::-1], zooms=zooms, origin=(0.0, -239.0, 0.0))
    for c, mod in enumerate(("t1", "t1ce", "t2", "flair")):
        amd.nifti.save_like(folder / f"{case}_{mod}.nii.gz", np.ascontiguousarray(np.round(vol[c]).astype(np.int16).transpose(2, 1, 0)), like)
    raw = np.stack([amd.nifti.load(folder / f"{case}_{m}.nii.gz").as_zyx().astype(np.float32) for m in ("t1", "t1ce", "t2", "flair")])
    return raw


def test_dropin_script_matches_oracle_driver(amd, gpu, tmp_path):
    patch = (32, 32, 32)
    results = tmp_path / "nnUNet_results"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans(patch)
    sds1 = [amd.synthetic.make_model("A", seed=40 + k, num_pool=2, max_feat=128)[0] for k in range(2)]
    sds2 = [amd.synthetic.make_model("B", seed=50 + k, num_pool=2, max_feat=128)[0] for k in range(2)]
    amd.checkpoint.save_model_folder(base / amd.driver.MODEL1, amd.driver.MODEL1.split("__")[0], sds1, plans)
    amd.checkpoint.save_model_folder(base / amd.driver.MODEL2, amd.driver.MODEL2.split("__")[0], sds2, plans)
    case_dir = tmp_path / "BraTS-GLI-00003-000"
    case_dir.mkdir()
    raw = _write_case(amd, case_dir, "BraTS-GLI-00003-000", (40, 56, 48), seed=77)
    out = tmp_path / "results" / "BraTS-GLI-00003-000"
    env = dict(os.environ, nnUNet_raw_data_base="x", nnUNet_preprocessed="y", RESULTS_FOLDER="z")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "run_brats2021_inference_singlethread.py"),
                          "--input", str(case_dir), "--output", str(out), "--results_folder", str(results),
                          "--folds", "0", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    final = out / "BraTS-GLI-00003-000.nii.gz"
    assert final.exists()                                              # run_full_pipeline.py:191-193
    for k in (1, 2):
        assert (out / f"temp_model{k}" / "BraTS-GLI-00003-000.nii.gz").exists()
        assert (out / f"temp_model{k}" / "temp_input" / "BraTS-GLI-00003-000_0003.nii.gz").exists()
    img = amd.nifti.load(final)
    like = amd.nifti.load(case_dir / "BraTS-GLI-00003-000_t1.nii.gz")
    assert img.data.dtype == np.uint8 and img.data.shape == like.data.shape and img.zooms == like.zooms
    assert np.allclose(img.affine, like.affine) and set(np.unique(img.data)) <= {0, 1, 2, 3}
    # oracle restatement of the whole driver
    seg1, p1, _ = driver_ref.predict_case(raw, sds1, unet_ref.default_cfg("batch"), patch)
    seg2, p2, _ = driver_ref.predict_case(raw, sds2, unet_ref.default_cfg("group", 16), patch)
    want = driver_ref.label_ensemble(seg1, seg2)
    got = img.as_zyx()
    for k, ref in ((1, seg1), (2, seg2)):
        part = amd.nifti.load(out / f"temp_model{k}" / "BraTS-GLI-00003-000.nii.gz").as_zyx()
        assert tiler_ref.brats_region_dice(part, ref)["mean"] >= 0.999
    d = tiler_ref.brats_region_dice(got, want)
    assert d["mean"] >= 0.999, d
    assert (got != want).mean() < 1e-4
    assert "Tumor Volume Analysis" in res.stdout and "SEGMENTATION COMPLETE" in res.stdout


def test_dropin_through_the_resident_worker_and_sequential_order(amd, gpu, tmp_path):
    """The drop-in script as a thin client of ``python -m brats_amd.worker``: the worker keeps both models
    resident, the client relays output and return code; its products are byte-identical to an in-process run
    (MI355_NO_WORKER=1), and to ``--sequential`` (the reference's order of work: model 1, model 2, file-level ensemble).
    A request for a missing model folder comes back as the reference's exit code 1, and the worker survives it."""
    import time
    patch = (32, 32, 32)
    results = tmp_path / "nnUNet_results"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans(patch)
    for name, preset, seed in ((amd.driver.MODEL1, "A", 60), (amd.driver.MODEL2, "B", 70)):
        amd.checkpoint.save_model_folder(base / name, name.split("__")[0], [amd.synthetic.make_model(preset, seed=seed, num_pool=2, max_feat=128)[0]], plans)
    case = "BraTS-GLI-00009-000"
    case_dir = tmp_path / case
    case_dir.mkdir()
    _write_case(amd, case_dir, case, (36, 52, 44), seed=79)
    sock = str(tmp_path / "w.sock")
    script = os.path.join(ROOT, "run_brats2021_inference_singlethread.py")
    env = dict(os.environ, MI355_WORKER_SOCKET=sock, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    worker = subprocess.Popen([sys.executable, "-m", "brats_amd.worker", "--socket", sock, "--results_folder", str(results), "--folds", "0"],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = ""
        for _ in range(50):  # (the ROCm runtime may print warnings of its own first)
            line = worker.stdout.readline()  # "[worker pid] listening on ... (2 models resident)"
            if "listening" in line or not line:
                break
        assert "listening" in line and "2 models" in line, line + (worker.stdout.read() if worker.poll() is not None else "")
        outs = {}
        for tag, extra_env, extra_args in (("worker", {}, []), ("inprocess", {"MI355_NO_WORKER": "1"}, []),
                                           ("sequential", {"MI355_NO_WORKER": "1"}, ["--sequential"])):
            out = tmp_path / f"results_{tag}" / case
            t0 = time.perf_counter()
            res = subprocess.run([sys.executable, script, "--input", str(case_dir), "--output", str(out), "--results_folder", str(results),
                                  "--folds", "0"] + extra_args, env=dict(env, **extra_env), cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
            assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
            assert "SEGMENTATION COMPLETE" in res.stdout and "Tumor Volume Analysis" in res.stdout
            print(f"{tag}: {time.perf_counter() - t0:.2f} s wall")
            outs[tag] = {rel: amd.nifti.load(out / rel).data for rel in (f"{case}.nii.gz", f"temp_model1/{case}.nii.gz", f"temp_model2/{case}.nii.gz")}
            assert (out / "temp_model2" / "temp_input" / f"{case}_0000.nii.gz").exists()
        for rel in outs["worker"]:
            assert np.array_equal(outs["worker"][rel], outs["inprocess"][rel]), rel
            # (model 2 is the GroupNorm member: its statistics are order-independent sums, so even that one is bit-stable)
            assert np.array_equal(outs["worker"][rel], outs["sequential"][rel]), rel
        # the reference's exit code for a missing model folder (:169-171) travels back through the worker
        res = subprocess.run([sys.executable, script, "--input", str(case_dir), "--output", str(tmp_path / "nope"), "--results_folder",
                              str(tmp_path / "no_such_results")], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert res.returncode == 1 and "Model not found" in res.stdout
        assert worker.poll() is None
    finally:
        sys.path.insert(0, ROOT)
        import json
        import socket as _socket
        try:
            with _socket.socket(_socket.AF_UNIX, _socket.SOCK_STREAM) as sk:
                sk.connect(sock)
                sk.sendall((json.dumps({"cmd": "shutdown"}) + "\n").encode())
                sk.recv(100)
        except OSError:
            pass
        try:
            worker.wait(timeout=30)
        except subprocess.TimeoutExpired:
            worker.kill()


def test_dropin_missing_models_exit_code(amd, gpu, tmp_path):
    (tmp_path / "in").mkdir()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "run_brats2021_inference_singlethread.py"),
                          "--input", str(tmp_path / "in"), "--output", str(tmp_path / "out"),
                          "--results_folder", str(tmp_path / "none")], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and "[ERROR] Model not found" in res.stdout


def test_dropin_resamples_a_grid_that_is_not_the_plans(amd, gpu, tmp_path):
    """A 2 mm input (e.g. an api.py upload) must not be segmented on the wrong grid.  Until round 3 the drop-in refused it; since
    round 4 it does what the reference's trainer.preprocess_patient / save_segmentation_nifti_from_softmax do (driver :89,
    :131-138): resample to the plans' 1 mm spacing (data order 3, inside mask order 1), predict, resample the probabilities back
    with order 1, and write labels on the INPUT's grid.  Checked against the oracle's restatement of those steps (PARITY UNPINNED
    like the rest of tiler_ref: nnU-Net v1 / skimage absent)."""
    patch = (32, 32, 32)
    results = tmp_path / "nnUNet_results"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans(patch)
    sds = {}
    for name, preset, seed in ((amd.driver.MODEL1, "A", 40), (amd.driver.MODEL2, "B", 50)):
        sds[name] = amd.synthetic.make_model(preset, seed=seed, num_pool=2, max_feat=64)[0]
        amd.checkpoint.save_model_folder(base / name, name.split("__")[0], [sds[name]], plans)
    case_dir = tmp_path / "case2mm"
    case_dir.mkdir()
    raw = _write_case(amd, case_dir, "BraTS-GLI-00009-000", (24, 32, 28), seed=79, zooms=(2.0, 2.0, 2.0))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "run_brats2021_inference_singlethread.py"), "--input", str(case_dir),
                          "--output", str(tmp_path / "out"), "--results_folder", str(results), "--folds", "0"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    img = amd.nifti.load(tmp_path / "out" / "BraTS-GLI-00009-000.nii.gz")
    like = amd.nifti.load(case_dir / "BraTS-GLI-00009-000_t1.nii.gz")
    assert img.data.shape == like.data.shape and img.zooms == like.zooms and set(np.unique(img.data)) <= {0, 1, 2, 3}
    # the oracle's restatement: crop, resample to 1 mm, z-score, sliding window + TTA per member, probabilities back with order 1
    data, props = tiler_ref.preprocess_case_resampled(raw, (2.0, 2.0, 2.0), (1.0, 1.0, 1.0))
    assert props["size_after_resampling"] != props["size_after_cropping"]
    segs = []
    for name, cfg in ((amd.driver.MODEL1, unet_ref.default_cfg("batch")), (amd.driver.MODEL2, unet_ref.default_cfg("group", 16))):
        probs = tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sds[name], cfg), data, patch, 3)
        back = tiler_ref.export_resample_probs(probs, props)
        segs.append(tiler_ref.paste_into_original(tiler_ref.regions_to_labels(back.astype(np.float32)), props["crop_bbox"], raw.shape[1:]))
    want = driver_ref.label_ensemble(segs[0], segs[1])
    d = tiler_ref.brats_region_dice(img.as_zyx(), want)
    assert d["mean"] >= 0.999, d


def test_nnunet_predict_cli_save_npz_and_probability_ensemble(amd, gpu, tmp_path):
    """nnUNet_predict-style surface (archived/kaist_original_inference.py:30-32): per-model prediction with
    --save_npz, then nnUNet_ensemble = mean of the two probability maps, thresholded once."""
    patch = (32, 32, 32)
    results = tmp_path / "nnUNet_results"
    base = results / "3d_fullres" / "Task500_BraTS2021"
    plans = amd.checkpoint.default_brats_plans(patch)
    tr1, tr2 = amd.driver.MODEL1.split("__")[0], amd.driver.MODEL2.split("__")[0]
    sds1 = [amd.synthetic.make_model("A", seed=60, num_pool=2, max_feat=128)[0]]
    sds2 = [amd.synthetic.make_model("B", seed=61, num_pool=2, max_feat=128)[0]]
    amd.checkpoint.save_model_folder(base / amd.driver.MODEL1, tr1, sds1, plans)
    amd.checkpoint.save_model_folder(base / amd.driver.MODEL2, tr2, sds2, plans)
    raw_dir = tmp_path / "raw"
    raw_dir.mkdir()
    raw = _write_case(amd, raw_dir, "case", (36, 44, 40), seed=78)
    in_dir = tmp_path / "imagesTs"
    amd.driver.prepare_input(raw_dir, in_dir)
    env = dict(os.environ, RESULTS_FOLDER=str(results))
    outs = []
    for k, tr in enumerate((tr1, tr2)):
        out = tmp_path / f"out{k}"
        res = subprocess.run([sys.executable, "-m", "brats_amd.nnunet_predict", "-i", str(in_dir), "-o", str(out), "-t", "500",
                              "-m", "3d_fullres", "-tr", tr, "-f", "0", "--save_npz"], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
        assert (out / "case.nii.gz").exists() and (out / "case.npz").exists()
        outs.append(out)
    ens = tmp_path / "ens"
    res = subprocess.run([sys.executable, "-m", "brats_amd.nnunet_predict", "--ensemble", str(outs[0]), str(outs[1]), "-o", str(ens)],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    got = amd.nifti.load(ens / "case.nii.gz").as_zyx()
    _, p1, props = driver_ref.predict_case(raw, sds1, unet_ref.default_cfg("batch"), patch)
    _, p2, _ = driver_ref.predict_case(raw, sds2, unet_ref.default_cfg("group", 16), patch)
    want = tiler_ref.paste_into_original(tiler_ref.regions_to_labels((p1 + p2) / 2.0), props["crop_bbox"],
                                         props["original_size_of_raw_data"])
    assert tiler_ref.brats_region_dice(got, want)["mean"] >= 0.999
    # post-processing of kaist_original_inference.py:33-34 on the ensemble folder: ET threshold, then the 2018/2019
    # label convention; once with a threshold that relabels and once with one that does not
    for thr, name in ((10 ** 9, "pp_hi"), (0, "pp_lo")):
        pp = tmp_path / name
        res = subprocess.run([sys.executable, "-m", "brats_amd.nnunet_predict", "--postprocess", str(ens), "-o", str(pp),
                              "--threshold", str(thr), "--replace_with", "2"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
        want_pp = driver_ref.convert_labels_back_to_brats(driver_ref.apply_brats_threshold(got, thr, 2))
        assert np.array_equal(amd.nifti.load(pp / "case.nii.gz").as_zyx(), want_pp)
    # the whole script in one call gives the same files as the three steps above
    kaist = tmp_path / "kaist"
    res = subprocess.run([sys.executable, "-m", "brats_amd.nnunet_predict", "--kaist", "-i", str(in_dir), "-o", str(kaist), "-f", "0",
                          "--threshold", str(10 ** 9)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert np.array_equal(amd.nifti.load(kaist / "case.nii.gz").as_zyx(), amd.nifti.load(tmp_path / "pp_hi" / "case.nii.gz").as_zyx())
    missing = subprocess.run([sys.executable, "-m", "brats_amd.nnunet_predict", "-i", str(in_dir), "-o", str(tmp_path / "x"),
                              "-tr", "nnUNetTrainerDoesNotExist"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert missing.returncode == 1 and "[ERROR] Model not found" in missing.stdout
