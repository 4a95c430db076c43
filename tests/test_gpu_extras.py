"""-m gpu: SURVEY 8f rows (label conversion, evaluator, cosine top-k) against the reference's own
numpy expressions (oracle/extras_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import driver_ref, extras_ref

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
