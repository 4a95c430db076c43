"""-m gpu: SURVEY 8f rows (label conversion, evaluator, cosine top-k) against the reference's own
numpy expressions (oracle/extras_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import driver_ref, extras_ref, tiler_ref, unet_ref

pytestmark = pytest.mark.gpu


def _labels(rs, shape, p=(0.7, 0.1, 0.1, 0.1)):
    return rs.choice(len(p), size=shape, p=p).astype(np.uint8)


@pytest.mark.parametrize("fmt", ["brats2025", "brats2021"])
def test_convert_labels(amd, gpu, fmt):
    seg = _labels(np.random.RandomState(1), (24, 30, 28))
    got = amd.evaluate.convert_labels(torch.from_numpy(seg).to(gpu), fmt).cpu().numpy()
    assert np.array_equal(got, extras_ref.convert_labels(seg, fmt))


@pytest.mark.parametrize("n_et,threshold", [(0, 200), (37, 200), (199, 200), (200, 200), (5000, 200), (5000, 100000)])
def test_apply_brats_threshold(amd, gpu, n_et, threshold):
    """KAIST post-processing (archived/kaist_original_inference.py:33): bit-exact against the oracle's restatement,
    on both sides of the threshold, at the full raw-volume size."""
    rs = np.random.RandomState(n_et + 1)
    seg = _labels(rs, (155, 240, 240), (0.96, 0.02, 0.02, 0.0))
    flat = seg.reshape(-1)
    flat[rs.choice(flat.size, size=n_et, replace=False)] = 3
    got, n3 = amd.evaluate.apply_brats_threshold(torch.from_numpy(seg).to(gpu), threshold, 2)
    assert n3 == n_et
    want = driver_ref.apply_brats_threshold(seg, threshold, 2)
    assert np.array_equal(got.cpu().numpy(), want)
    # followed by the label convention of :34 (= convert_labels_to_brats.py:46-55)
    assert np.array_equal(amd.evaluate.convert_labels(got, "brats2021").cpu().numpy(), driver_ref.convert_labels_back_to_brats(want))


def test_evaluator_matches_reference_formulas(amd, gpu):
    rs = np.random.RandomState(2)
    gt = _labels(rs, (155, 240, 240), (0.97, 0.01, 0.012, 0.008))
    pred = gt.copy()
    flip = rs.uniform(size=gt.shape) < 0.02
    pred[flip] = _labels(rs, int(flip.sum()), (0.4, 0.2, 0.2, 0.2))
    res = amd.evaluate.evaluate(torch.from_numpy(pred).to(gpu), torch.from_numpy(gt).to(gpu))
    for lab in (1, 2, 3):
        want = extras_ref.calculate_metrics(pred, gt, lab)
        for k in ("dice", "iou", "sensitivity", "specificity", "tp", "fp", "fn", "tn"):
            assert res[lab][k] == pytest.approx(want[k], rel=1e-12, abs=1e-12), (lab, k)
    comp = extras_ref.compound(pred, gt)
    for name in ("WT", "TC", "ET"):
        for k in ("dice", "iou", "sensitivity"):
            assert res[name][k] == pytest.approx(comp[name][k], rel=1e-12)
    assert res["mean_dice"] == pytest.approx(comp["mean_dice"], rel=1e-12)
    same = amd.evaluate.evaluate(torch.from_numpy(gt).to(gpu), torch.from_numpy(gt).to(gpu))
    assert same["mean_dice"] == pytest.approx(1.0)
    with pytest.raises(ValueError):
        amd.evaluate.confusion(torch.from_numpy(gt).to(gpu), torch.from_numpy(gt[:10]).to(gpu))


def test_out_of_range_labels_never_alias_a_real_label(amd, gpu):
    """labels beyond the evaluated range go to an explicit "other" bin - the reference compares ``== label``
    (evaluate_segmentation.py:20-21) and counts ``img == 3`` only in the ET threshold."""
    rs = np.random.RandomState(5)
    gt = rs.choice([0, 1, 2, 3, 4, 7, 200], size=(40, 50, 60), p=[0.6, 0.1, 0.1, 0.1, 0.05, 0.03, 0.02]).astype(np.uint8)
    pred = gt.copy()
    flip = rs.uniform(size=gt.shape) < 0.1
    pred[flip] = rs.choice([0, 1, 2, 3, 4, 9, 255], size=int(flip.sum())).astype(np.uint8)
    res = amd.evaluate.evaluate(torch.from_numpy(pred).to(gpu), torch.from_numpy(gt).to(gpu))
    for lab in (1, 2, 3):
        want = extras_ref.calculate_metrics(pred, gt, lab)
        for k in ("tp", "fp", "fn", "tn", "dice", "specificity"):
            assert res[lab][k] == pytest.approx(want[k], rel=1e-12, abs=1e-12), (lab, k)
    cm = amd.evaluate.confusion(torch.from_numpy(pred).to(gpu), torch.from_numpy(gt).to(gpu), 6)
    assert cm.sum() == gt.size and cm[5].sum() == int((pred >= 5).sum()) and cm[:, 5].sum() == int((gt >= 5).sum())
    seg = gt.copy()
    _, n3 = amd.evaluate.apply_brats_threshold(torch.from_numpy(seg).to(gpu), 10 ** 9, 2)
    assert n3 == int((seg == 3).sum())          # labels 4, 7, 200 are not ET voxels


DOCS = [{"term": "glioma", "text": "A glioma is a tumour that starts in the glial cells of the brain or spine."},
        {"term": "edema", "text": "Peritumoral edema is swelling around a tumour caused by fluid accumulation."},
        {"term": "necrosis", "text": "Necrotic core refers to dead tissue in the centre of the tumour."},
        {"term": "enhancing", "text": "Enhancing tumour is the region that takes up contrast on T1ce MRI."},
        {"term": "flair", "text": "FLAIR is an MRI sequence that suppresses fluid signal to show edema."},
        {"term": "segmentation", "text": "Segmentation assigns every voxel of the MRI volume to a tissue label."},
        {"term": "dice", "text": "The Dice score measures the overlap between prediction and ground truth."}]


def test_dummy_vector_store_retrieve(amd, gpu):
    ref = extras_ref.DummyVectorStoreRef(DOCS)
    store = amd.retrieval.DummyVectorStore(DOCS)
    assert store.vocab == ref.vocab
    for q in ("what is edema around the tumour", "explain the dice overlap score", "which MRI sequence shows fluid", "zzz"):
        want = ref.retrieve(q, 2)
        got = store.retrieve(q, 2)
        assert [d["term"] for d, _ in got] == [d["term"] for d, _ in want], q
        assert np.allclose([s for _, s in got], [s for _, s in want], atol=1e-6)


def test_cosine_topk_large(amd, gpu):
    rs = np.random.RandomState(3)
    v = rs.standard_normal((200000, 384)).astype(np.float32)
    idx = amd.retrieval.VectorIndex(v)
    q = rs.standard_normal(384)
    q /= np.linalg.norm(q)
    got = idx.topk(q, 5)
    scores = idx.host @ q
    want = np.argsort(scores)[::-1][:5]
    assert [i for i, _ in got] == list(want)
    assert np.allclose([s for _, s in got], scores[want], atol=1e-5)


@pytest.mark.parametrize("present", [(0.95, 0.02, 0.02, 0.01, 0.0), (0.9, 0.03, 0.03, 0.02, 0.02), (1.0, 0.0, 0.0, 0.0, 0.0), (0.98, 0.0, 0.02, 0.0, 0.0)])
def test_tumor_region_features(amd, gpu, present):
    """feature_extraction/utils.py:167-216 (volumes, centroids, bounding boxes of ncr / ed / et / tc / wt) from one device
    pass: integers exact, centroids equal to numpy's float64 means (exact integer sums divided once)."""
    rs = np.random.RandomState(11)
    seg = _labels(rs, (240, 240, 155), present)          # nibabel axis order x, y, z; label 4 = BraTS-2021 ET
    seg[:40] = 0                                          # make the boxes non-trivial
    seg[:, 200:] = 0
    got = amd.evaluate.tumor_region_features(torch.from_numpy(seg).to(gpu), 0.001)
    want = extras_ref.tumor_region_features(seg, 0.001)
    assert got.keys() == want.keys()
    for name in want:
        assert got[name]["volume_cm3"] == want[name]["volume_cm3"], name
        assert got[name]["bounding_box"] == want[name]["bounding_box"], name
        if want[name]["centroid"] is None:
            assert got[name]["centroid"] is None
        else:
            for k in "xyz":
                assert abs(got[name]["centroid"][k] - want[name]["centroid"][k]) <= 1e-9, (name, k)


# --------------------------------------------------------------------------- resampling (round 4)
@pytest.mark.parametrize("order", [0, 1, 3])
@pytest.mark.parametrize("axis,n_out", [(3, 37), (3, 11), (2, 40), (1, 9), (1, 30)])
def test_resize_axis_matches_scipy_zoom(amd, gpu, order, axis, n_out):
    """mi355_resize_axis against scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) along one axis (what
    skimage.transform.resize(order, mode='edge', anti_aliasing=False) computes before its clip): up- and down-sampling along x, y
    and z of a [C, Z, Y, X] tensor, nearest / linear / cubic B-spline (prefilter with scipy's 12-sample edge padding)."""
    from scipy.ndimage import zoom
    rs = np.random.RandomState(3)
    x = (rs.standard_normal((2, 14, 19, 23)) * 50 + 100).astype(np.float32)
    factors = [1.0, 1.0, 1.0, 1.0]
    factors[axis] = n_out / x.shape[axis]
    ref = zoom(x.astype(np.float64), factors, order=order, mode="nearest", grid_mode=True)
    got = amd.ops.resize_axis(torch.from_numpy(x).to(gpu), axis, n_out, order).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= (0.0 if order == 0 else 2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("case", [
    dict(shape=(20, 24, 18), new=(31, 24, 27), sep=False, axis=None),     # isotropic-ish: 3-D cubic resize, clip per channel
    dict(shape=(12, 30, 28), new=(36, 25, 33), sep=True, axis=0),         # thick slices: cubic in plane, nearest along z
    dict(shape=(26, 9, 22), new=(20, 27, 22), sep=True, axis=1),          # the low-resolution axis is y
    dict(shape=(20, 24, 18), new=(10, 12, 9), sep=False, axis=None),      # down-sampling
])
def test_resample_data_and_mask_match_the_oracle(amd, gpu, case):
    """preprocessing.resample_data_or_seg (device) against oracle/tiler_ref.resample_data_or_seg (nnU-Net v1's function restated on
    scipy; PARITY UNPINNED): image data with order 3 (+ skimage's clip to the input's range: per channel, per slice in the
    separate-z mode) and the inside mask through resize_segmentation's rule (linear indicator >= 0.5)."""
    pp = amd.preprocessing
    rs = np.random.RandomState(11)
    from scipy.ndimage import gaussian_filter
    x = np.stack([gaussian_filter(rs.standard_normal(case["shape"]), 1.5) * 300 + 500 * (c + 1) for c in range(3)]).astype(np.float32)
    axis = None if case["axis"] is None else np.array([case["axis"]])
    ref = tiler_ref.resample_data_or_seg(x, case["new"], False, axis, 3, case["sep"], order_z=0)
    got = pp.resample_data_or_seg(torch.from_numpy(x).to(gpu), case["new"], 3, case["sep"], case["axis"], 0).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in case["shape"]], indexing="ij")
    inside = (((zz - case["shape"][0] / 2) / (case["shape"][0] * 0.4)) ** 2 + ((yy - case["shape"][1] / 2) / (case["shape"][1] * 0.35)) ** 2 +
              ((xx - case["shape"][2] / 2) / (case["shape"][2] * 0.45)) ** 2) <= 1.0
    seg = np.where(inside, 0, -1).astype(np.float32)[None]
    ref_m = tiler_ref.resample_data_or_seg(seg, case["new"], True, axis, 1, case["sep"], order_z=0)[0] >= 0
    got_m = pp.resample_data_or_seg(torch.from_numpy(inside.astype(np.float32))[None].to(gpu), case["new"], 1, case["sep"], case["axis"], 0,
                                    is_mask=True)[0].cpu().numpy().astype(bool)
    # (the indicator is resized in fp32 here and in fp64 there: a voxel whose value is 0.5 to the last bit may fall either way)
    assert (got_m != ref_m).sum() <= 2, int((got_m != ref_m).sum())


@pytest.mark.parametrize("spacing", [(2.0, 2.0, 2.0), (4.0, 1.0, 1.0)])
def test_preprocess_predict_export_with_resampling_matches_the_oracle(amd, gpu, spacing):
    """The whole path on a case whose grid is NOT the plans' grid (never BraTS; e.g. a 2 mm upload through api.py): crop ->
    resample data (order 3) + mask (order 1) to 1 mm -> masked z-score -> sliding window -> probabilities resampled back with
    order 1 (driver :131-138) -> region labels pasted at the crop box, against the oracle's restatement of the same steps."""
    sd, _ = amd.synthetic.make_model("A", seed=21, num_pool=2, max_feat=64)
    net = amd.UNet(sd, norm="batch")
    patch = (32, 32, 32)
    shape = tuple(int(round(s / f)) for s, f in zip((48, 56, 44), spacing))
    raw = amd.synthetic.make_volume(seed=9, shape=shape)
    plans = amd.checkpoint.default_brats_plans(patch)
    data, props = amd.preprocessing.preprocess_case(raw, gpu, plans=plans, spacing_zyx=spacing)
    ref_data, ref_props = tiler_ref.preprocess_case_resampled(raw, spacing, (1.0, 1.0, 1.0))
    assert tuple(data.shape) == tuple(ref_data.shape) and props["size_after_resampling"] == ref_props["size_after_resampling"]
    assert props["crop_bbox"] == ref_props["crop_bbox"] and props["size_after_cropping"] == ref_props["size_after_cropping"]
    assert np.abs(data.cpu().numpy() - ref_data).max() <= 2e-3       # z-scored values (the mask may differ in a voxel or two: see above)
    probs = amd.predictor.predict_folds([net], data, patch)
    back = amd.preprocessing.resample_probabilities_for_export(probs, props)
    ref_probs = tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")), ref_data, patch, 3)
    ref_back = tiler_ref.export_resample_probs(ref_probs, ref_props)
    assert tuple(back.shape[1:]) == props["size_after_cropping"] == tuple(ref_back.shape[1:])
    assert np.abs(back.cpu().numpy() - ref_back).max() <= 5e-3
    lo = [b[0] for b in props["crop_bbox"]]
    seg = amd.ops.regions_to_labels(back, (1, 2, 3), lo, props["original_size_of_raw_data"]).cpu().numpy()
    want = tiler_ref.paste_into_original(tiler_ref.regions_to_labels(ref_back.astype(np.float32)), ref_props["crop_bbox"], raw.shape[1:])
    assert tiler_ref.brats_region_dice(seg, want)["mean"] >= 0.999
    net.close()
