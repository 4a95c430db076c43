"""not gpu: the file-level boundary (NIfTI, checkpoint folders, driver host logic)."""
import os
import pickle

import numpy as np
import pytest

from oracle import driver_ref


def test_nifti_round_trip_and_geometry(amd, tmp_path):
    nifti = amd.nifti
    like = nifti.make_header((5, 6, 3), zooms=(1.0, 1.5, 2.0), origin=(-3.0, 4.0, 5.0))
    arr = (np.arange(5 * 6 * 3).reshape(5, 6, 3) * 7 % 4000).astype(np.int16)
    p = tmp_path / "vol.nii.gz"
    nifti.save_like(p, arr, like)
    im = nifti.load(p)
    assert im.data.dtype == np.int16 and np.array_equal(im.data, arr)
    assert im.zooms == (1.0, 1.5, 2.0)
    assert np.allclose(im.affine, [[1, 0, 0, -3], [0, 1.5, 0, 4], [0, 0, 2, 5], [0, 0, 0, 1]])
    assert im.as_zyx().shape == (3, 6, 5) and im.as_zyx()[2, 1, 4] == arr[4, 1, 2]
    # label file written "like" the input: uint8, same zooms / affine / shape (what downstream reads)
    seg = (arr % 4).astype(np.uint8)
    q = tmp_path / "seg.nii.gz"
    nifti.save_like(q, seg, im)
    s = nifti.load(q)
    assert s.data.dtype == np.uint8 and np.array_equal(s.data, seg) and s.zooms == im.zooms
    assert np.allclose(s.affine, im.affine)
    raw = __import__("gzip").open(q).read()
    assert len(raw) == 352 + seg.size and raw[344:348] == b"n+1\0"
    with pytest.raises(ValueError):
        (tmp_path / "bad.nii.gz").write_bytes(__import__("gzip").compress(b"\0" * 400))
        nifti.load(tmp_path / "bad.nii.gz")


def test_nifti_scaling_and_quaternion_affine(amd, tmp_path):
    import struct
    nifti = amd.nifti
    like = nifti.make_header((2, 2, 2), zooms=(1, 1, 1))
    hdr = bytearray(like.header)
    struct.pack_into("<2f", hdr, 112, 2.0, 10.0)      # scl_slope, scl_inter
    struct.pack_into("<2h", hdr, 252, 1, 0)           # qform only
    struct.pack_into("<6f", hdr, 256, 0.0, 0.0, 1.0, 10.0, 20.0, 30.0)  # 180 deg about z
    data = np.arange(8, dtype=np.int16).reshape(2, 2, 2)
    p = tmp_path / "q.nii"
    p.write_bytes(bytes(hdr) + b"\0\0\0\0" + data.tobytes(order="F"))
    im = nifti.load(p)
    assert np.allclose(im.data, data * 2.0 + 10.0)
    assert np.allclose(im.affine, [[-1, 0, 0, 10], [0, -1, 0, 20], [0, 0, 1, 30], [0, 0, 0, 1]], atol=1e-6)


def test_prepare_input_naming_and_missing_modality(amd, tmp_path, capsys):
    src = tmp_path / "case"
    src.mkdir()
    for case, mods in (("BraTS-GLI-00003-000", ("t1", "t1ce", "t2", "flair", "seg")), ("BraTS-GLI-00005-000", ("t1", "t2"))):
        for m in mods:
            (src / f"{case}_{m}.nii.gz").write_bytes(b"x" + m.encode())
    out = tmp_path / "tmp_in"
    prepared = amd.driver.prepare_input(src, out)
    assert [c for c, _ in prepared] == ["BraTS-GLI-00003-000"]
    files = prepared[0][1]
    assert [os.path.basename(f) for f in files] == [f"BraTS-GLI-00003-000_{i:04d}.nii.gz" for i in range(4)]
    assert open(files[1], "rb").read() == b"xt1ce" and open(files[3], "rb").read() == b"xflair"
    assert "Missing t1ce for BraTS-GLI-00005-000" in capsys.readouterr().out


def test_checkpoint_folder_round_trip_and_safe_unpickle(amd, tmp_path):
    ck = amd.checkpoint
    sd, _ = amd.synthetic.make_model("B", num_pool=2, max_feat=64)
    name = "nnUNetTrainerV2BraTSRegions_DA4_BN_BD_largeUnet_Groupnorm"
    folder = ck.save_model_folder(tmp_path / f"{name}__nnUNetPlansv2.1", name, [sd, sd, sd], ck.default_brats_plans((32, 32, 32)))
    m = ck.load_model_folder(folder, (0, 2))
    assert (m.norm, m.num_groups, m.patch_size, m.regions, len(m.fold_state_dicts)) == ("group", 16, (32, 32, 32), True, 2)
    assert all(np.array_equal(m.fold_state_dicts[1][k], sd[k]) for k in sd)
    assert ck.norm_from_trainer_name("nnUNetTrainerV2BraTSRegions_DA4_BN_BD")[0] == "batch"
    assert ck.norm_from_trainer_name("nnUNetTrainerV2")[0] == "instance"
    with pytest.raises(FileNotFoundError):
        ck.load_model_folder(folder, (0, 4))
    # a pickle that tries to import anything else is refused
    evil = tmp_path / "evil.pkl"

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))
    evil.write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        ck.safe_pickle_load(evil)
    ref_plans = "/root/reference/data/temp_inference_output1"
    if os.path.exists(ref_plans):  # the reference's stray plans.pkl
        p = ck.safe_pickle_load(ref_plans)
        assert list(p["plans_per_stage"][0]["patch_size"]) == [128, 128, 128] and p["base_num_features"] == 32


def test_default_plans_equal_the_references_plans_field_by_field(amd):
    """pin what the reference DOES hold.  data/temp_inference_output1 is the Task500_BraTS2021 plans file
    the reference ships; every field of checkpoint.default_brats_plans() (what synthetic model folders, the smoke run and the
    bench are built from) must equal it, and preprocessing.check_plans / check_spacing must accept it as it is - i.e. the
    path's assumptions (identity transpose, nonCT + use_mask_for_norm on four modalities, 1 mm target spacing, 128^3 patch,
    five 2x2x2 poolings, 3x3x3 convs) are the reference's own, not ours.  Build container only (the file is not on the GPU box)."""
    ref_plans = "/root/reference/data/temp_inference_output1"
    if not os.path.exists(ref_plans):
        pytest.skip("reference tree absent (GPU box)")
    ck, pre = amd.checkpoint, amd.preprocessing
    ref = ck.safe_pickle_load(ref_plans)
    ours = ck.default_brats_plans()

    def norm(v):
        if isinstance(v, dict):
            return {int(k) if not isinstance(k, str) else k: norm(x) for k, x in v.items()}
        if isinstance(v, (list, tuple, np.ndarray)):
            return [norm(x) for x in (v.tolist() if isinstance(v, np.ndarray) else v)]
        if isinstance(v, (np.integer, np.bool_)):
            return int(v)
        if isinstance(v, np.floating):
            return float(v)
        return v

    for key, val in ours.items():
        if key == "plans_per_stage":
            continue
        assert key in ref, key
        assert norm(ref[key]) == norm(val), (key, ref[key], val)
    assert sorted(ref["plans_per_stage"]) == sorted(ours["plans_per_stage"]) == [0]
    for key, val in ours["plans_per_stage"][0].items():
        assert norm(ref["plans_per_stage"][0][key]) == norm(val), (key, ref["plans_per_stage"][0][key], val)
    # fields the path relies on without carrying them in its default plans
    st = ref["plans_per_stage"][0]
    assert norm(st["original_spacing"]) == [1.0, 1.0, 1.0] and norm(st["num_pool_per_axis"]) == [5, 5, 5]
    assert ref["preprocessor_name"] == "GenericPreprocessor" and ref["conv_per_stage"] == 2
    assert ref["keep_only_largest_region"] is None and ref["min_region_size_per_class"] is None   # no plans-driven post-processing
    # the product's own gatekeepers accept the reference's plans unchanged ...
    assert list(pre.check_plans(ref, 4)) == [True, True, True, True]
    pre.check_spacing(ref, (1.0, 1.0, 1.0), (155, 240, 240))
    assert norm(pre._stage_plans(ref)["patch_size"]) == [128, 128, 128]
    # ... and refuse what they do not implement, on the reference's plans with one field changed
    bad = dict(ref, transpose_forward=[1, 0, 2])
    with pytest.raises(pre.UnsupportedPlansError):
        pre.check_plans(bad, 4)
    with pytest.raises(pre.UnsupportedPlansError):
        pre.check_plans(dict(ref, normalization_schemes={0: "CT", 1: "nonCT", 2: "nonCT", 3: "nonCT"}), 4)
    # the tile counts the plans' size statistics imply through the product's step table
    assert amd.parallel.tiles_of_shape(norm(st["median_patient_size_in_voxels"])) == 8


def test_calculate_volumes_uses_label_4(amd, tmp_path):
    seg = np.zeros((4, 4, 4), np.uint8)
    seg[0, 0, :3] = 1
    seg[1, 0, :2] = 2
    seg[2, 0, :4] = 3   # nnU-Net ET label: NOT counted (reference quirk, driver :231)
    seg[3, 0, :1] = 4
    p = tmp_path / "s.nii.gz"
    amd.nifti.save_like(p, seg, amd.nifti.make_header((4, 4, 4), zooms=(1.0, 2.0, 5.0)))
    v = amd.driver.calculate_volumes(p)
    want = driver_ref.calculate_volumes(seg, (1.0, 2.0, 5.0))
    assert v == pytest.approx(want) and v["ET"] == pytest.approx(0.01) and v["WT"] == pytest.approx(0.06)


def test_missing_model_dir_exits_1(amd, tmp_path):
    with pytest.raises(SystemExit) as e:
        amd.driver.run_model_single_threaded(tmp_path / "nope", tmp_path, tmp_path / "out")
    assert e.value.code == 1


def test_plans_are_checked_not_assumed(amd):
    """the plans decide what preprocessing means; anything this path does not implement is refused."""
    pp, ck = amd.preprocessing, amd.checkpoint
    plans = ck.default_brats_plans()
    assert pp.check_plans(plans, 4) == [True] * 4 and pp.check_plans(None, 4) == [True] * 4
    assert pp.check_spacing(plans, (1.0, 1.0, 1.0), (140, 171, 137)) is None
    assert pp.check_spacing(plans, (1.0, 1.0004, 1.0), (140, 171, 137)) is None   # nnU-Net: round(spacing ratio * shape) == shape -> no resample
    # round 4: a grid the plans would resample is no longer refused - check_spacing returns nnU-Net's resampling decisions
    assert pp.check_spacing(plans, (2.0, 2.0, 2.0), (78, 120, 120)) == ((156, 240, 240), False, None, (1.0, 1.0, 1.0))   # a 2 mm header (e.g. an api.py upload)
    assert pp.check_spacing(plans, (5.0, 1.0, 1.0), (30, 120, 100)) == ((150, 120, 100), True, 0, (1.0, 1.0, 1.0))       # thick slices: z separately
    assert pp.resample_plan((30, 120, 100), (1.25, 1.25, 0.24), (1.0, 1.0, 1.0))[1:] == (False, None)                      # two axes share the largest spacing
    with pytest.raises(pp.UnsupportedPlansError, match="transpose_forward"):
        pp.check_plans({**plans, "transpose_forward": [2, 0, 1]}, 4)
    with pytest.raises(pp.UnsupportedPlansError, match="CT"):
        pp.check_plans({**plans, "normalization_schemes": {0: "CT", 1: "nonCT", 2: "nonCT", 3: "nonCT"}}, 4)
    with pytest.raises(pp.UnsupportedPlansError, match="modalities"):
        pp.check_plans(plans, 3)
    mixed = pp.check_plans({**plans, "use_mask_for_norm": {0: True, 1: False, 2: True, 3: False}}, 4)
    assert mixed == [True, False, True, False]
    other = {**plans, "plans_per_stage": {0: {**plans["plans_per_stage"][0], "current_spacing": np.array([2.0, 1.0, 1.0])}}}
    assert pp.check_spacing(other, (2.0, 1.0, 1.0), (70, 171, 137)) is None
    assert pp.check_spacing(other, (1.0, 1.0, 1.0), (140, 171, 137)) == ((70, 171, 137), False, None, (2.0, 1.0, 1.0))


def test_worker_protocol_without_a_gpu(amd, tmp_path, monkeypatch):
    """The resident worker's request / reply protocol and the client's fallbacks, with the driver's main() replaced by a
    stand-in (no GPU here): output lines and the return code travel back, SystemExit becomes a return code, relative
    paths are resolved against the CLIENT's working directory, a missing or dead socket means "run in-process" (None)."""
    import io
    import json
    import socket
    import threading
    import time
    from brats_amd import worker, driver
    calls = []

    def fake_main(argv, script_dir=None, model_cache=None):
        print("running", " ".join(argv))
        calls.append((list(argv), script_dir, model_cache is not None))
        if "--boom" in argv:
            raise SystemExit(1)
        return 0

    monkeypatch.setattr(driver, "main", fake_main)
    sock = str(tmp_path / "w.sock")
    assert worker.request(["--input", "x"], "/sd", sock) is None          # no worker: the caller runs in-process
    t = threading.Thread(target=worker.serve, args=(sock,), daemon=True)
    t.start()
    for _ in range(100):
        if os.path.exists(sock):
            break
        time.sleep(0.05)
    out = io.StringIO()
    argv = worker.absolutise(["--input", "in/case", "--output=out/case", "--folds", "0", "1"], cwd="/work")
    assert argv == ["--input", "/work/in/case", "--output=/work/out/case", "--folds", "0", "1"]
    assert worker.request(argv, "/sd", sock, out) == 0
    assert out.getvalue() == "running " + " ".join(argv) + "\n" and calls[-1] == (argv, "/sd", True)
    assert worker.request(["--boom"], "/sd", sock, io.StringIO()) == 1
    with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as sk:            # ping, then shutdown
        sk.connect(sock)
        fh = sk.makefile("rw")
        fh.write(json.dumps({"cmd": "ping"}) + "\n")
        fh.flush()
        assert json.loads(fh.readline())["rc"] == 0
    with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as sk:
        sk.connect(sock)
        fh = sk.makefile("rw")
        fh.write(json.dumps({"cmd": "shutdown"}) + "\n")
        fh.flush()
        assert json.loads(fh.readline())["rc"] == 0
    t.join(5)
    assert not t.is_alive() and not os.path.exists(sock)
    monkeypatch.setenv("MI355_NO_WORKER", "1")
    assert worker.request(argv, "/sd", sock) is None


def test_worker_failure_classification_and_socket_directory(amd, tmp_path):
    """an ordinary error whose text happens to contain 'hip' must not end the serve loop; a socket directory that is not
    ours alone is refused."""
    from brats_amd import worker
    from brats_amd._lib import Mi355Error
    assert worker._is_device_failure(Mi355Error("mi355_sw_predict failed"), Mi355Error)
    assert worker._is_device_failure(RuntimeError("HIP error: invalid device function"), Mi355Error)
    assert worker._is_device_failure(RuntimeError("CUDA error: an illegal memory access was encountered"), Mi355Error)
    assert not worker._is_device_failure(FileNotFoundError("/data/ship01/case_t1.nii.gz not found"), Mi355Error)
    assert not worker._is_device_failure(ValueError("data must be [C=4, Z, Y, X] (membership of this hip class)"), Mi355Error)
    assert not worker._is_device_failure(RuntimeError("plans: the checkpoint was written for another patch size (hipify?)"), Mi355Error)
    open_dir = tmp_path / "shared"
    open_dir.mkdir(mode=0o755)
    os.chmod(open_dir, 0o755)
    with pytest.raises(SystemExit) as ei:
        worker.serve(str(open_dir / "w.sock"))
    assert "0700" in str(ei.value)
