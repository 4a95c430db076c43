"""not gpu: the numpy restatement of the nnU-Net v1 tiler / export / preprocessing and of the
pure-numpy parts of the reference driver.  The upstream package is absent ("parity unpinned",
oracle/__init__.py); these tests hold the restatement to the facts the reference does fix:
the tile tables implied by its plans file, scipy's Gaussian, and the
driver's own numpy expressions."""
import os

import numpy as np
import pytest
import torch

from oracle import tiler_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_compute_steps_known_tables():
    cs = tiler_ref.compute_steps_for_sliding_window
    # median crop (140,171,137) of data/temp_inference_output1 -> 8 tiles; uncropped -> 18 tiles
    assert cs((128,) * 3, (140, 171, 137), 0.5) == [[0, 12], [0, 43], [0, 9]]
    assert cs((128,) * 3, (155, 240, 240), 0.5) == [[0, 27], [0, 56, 112], [0, 56, 112]]
    assert cs((128,) * 3, (155, 208, 177), 0.5) == [[0, 27], [0, 40, 80], [0, 49]]  # max crop -> 12 tiles
    assert cs((128,) * 3, (128, 128, 128), 0.5) == [[0], [0], [0]]
    # half-to-even rounding of the actual step (np.round)
    assert cs((4,), (9,), 0.5) == [[0, 2, 3, 5]] or cs((4,), (9,), 0.5) == [[0, 2, 4, 5]]
    for img in range(128, 300):
        s = cs((128,), (img,), 0.5)[0]
        assert s[0] == 0 and s[-1] == img - 128 and all(b - a <= 64 for a, b in zip(s, s[1:]))


def test_gaussian_is_separable_scipy_filter():
    g = tiler_ref.get_gaussian((128, 128, 128))
    prof = np.exp(-0.5 * ((np.arange(128) - 64) / 16.0) ** 2)
    outer = (prof[:, None, None] * prof[None, :, None] * prof[None, None, :]).astype(np.float32)
    assert g.dtype == np.float32 and g.max() == 1.0 and g[64, 64, 64] == 1.0
    assert np.allclose(g, outer, rtol=2e-6, atol=0)
    assert g.min() > 0 and abs(g.min() / 3.775e-11 - 1) < 1e-3
    g2 = tiler_ref.get_gaussian((32, 48, 16))
    assert g2.shape == (32, 48, 16) and g2[16, 24, 8] == 1.0


def test_mirror_schedule_order_and_flipback():
    assert tiler_ref.mirror_schedule((0, 1, 2)) == [(), (4,), (3,), (4, 3), (2,), (4, 2), (3, 2), (4, 3, 2)]
    assert tiler_ref.mirror_schedule((0, 2)) == [(), (4,), (2,), (4, 2)]
    assert tiler_ref.mirror_schedule(()) == [()]
    # an equivariant "network" (elementwise) must be unchanged by TTA; a position-dependent one
    # must come out as the mean over the 8 flips of its position ramp
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 2, 4, 6, 8)).astype(np.float32))
    same = tiler_ref.mirror_and_predict(lambda t: t * 2.0, x, (0, 1, 2), True, "identity")
    assert torch.allclose(same, x * 2.0, atol=1e-6)
    ramp = torch.arange(8, dtype=torch.float32).view(1, 1, 1, 1, 8).expand(1, 2, 4, 6, 8)
    out = tiler_ref.mirror_and_predict(lambda t: ramp.clone(), x, (0, 1, 2), True, "identity")
    assert torch.allclose(out, torch.full_like(out, 3.5))


def test_tiled_prediction_of_constant_and_identity_nets():
    vol = np.random.RandomState(1).standard_normal((2, 20, 45, 37)).astype(np.float32)
    # constant logits -> constant probabilities everywhere, whatever the tiling / padding
    p = tiler_ref.predict_3d_tiled(lambda t: torch.zeros(t.shape[0], 3, *t.shape[2:]), vol, (16, 32, 32), 3)
    assert p.shape == (3, 20, 45, 37) and np.allclose(p, 0.5, atol=1e-6)
    # identity "network" on channel 0: weighted mean of identical values = the value itself
    p, steps, agg, cnt = tiler_ref.predict_3d_tiled(lambda t: t[:, :1].repeat(1, 3, 1, 1, 1), vol, (16, 32, 32), 3,
                                                    nonlin="identity", return_parts=True)
    assert steps == [[0, 4], [0, 13], [0, 5]]
    assert np.allclose(p[0], vol[0], atol=1e-5)
    # volume smaller than the patch: padded symmetrically (below = d//2) and cropped back
    small = vol[:, :10, :20, :30]
    padded, lo = tiler_ref.pad_to_patch(small, (16, 32, 32))
    assert padded.shape == (2, 16, 32, 32) and lo == [3, 6, 1]
    assert np.array_equal(padded[:, 3:13, 6:26, 1:31], small)


def test_regions_to_labels_and_paste():
    probs = np.zeros((3, 2, 2, 2), np.float32)
    probs[0, 0, 0, 0] = 0.9                      # WT only -> 1
    probs[0, 0, 0, 1] = probs[1, 0, 0, 1] = 0.6  # WT+TC -> 2
    probs[:, 0, 1, 0] = 0.7                      # all three -> 3
    probs[2, 0, 1, 1] = 0.8                      # ET only (inconsistent) -> 3: later regions overwrite
    probs[0, 1, 0, 0] = 0.5                      # exactly 0.5 is NOT above the threshold
    seg = tiler_ref.regions_to_labels(probs)
    assert seg[0, 0, 0] == 1 and seg[0, 0, 1] == 2 and seg[0, 1, 0] == 3 and seg[0, 1, 1] == 3 and seg[1, 0, 0] == 0
    full = tiler_ref.paste_into_original(seg, [[1, 3], [2, 4], [0, 2]], (4, 5, 3))
    assert full.shape == (4, 5, 3) and full.sum() == seg.sum() and np.array_equal(full[1:3, 2:4, 0:2], seg)


def test_label_ensemble_truth_table_from_reference_expression():
    t = np.load(os.path.join(GOLD, "driver_tables.npz"))["label_round"]
    # (0,1)->0, (0,2)->1, (0,3)->2, (1,2)->2, (1,3)->2, (2,3)->2, equal -> same
    assert (t[0, 1], t[0, 2], t[0, 3], t[1, 2], t[1, 3], t[2, 3]) == (0, 1, 2, 2, 2, 2)
    assert all(t[i, i] == i for i in range(5)) and np.array_equal(t, t.T)


def test_preprocess_crop_fill_holes_and_masked_zscore(amd):
    vol = amd.synthetic.make_volume(seed=5, shape=(40, 48, 44))
    data, props = tiler_ref.preprocess_case(vol)
    bbox = props["crop_bbox"]
    assert data.shape[1:] == tuple(b[1] - b[0] for b in bbox)
    cropped, inside, bbox2 = tiler_ref.crop_to_nonzero(vol)
    assert bbox2 == bbox
    # the synthetic volume has interior zero holes: they are inside the mask after hole filling
    raw_nonzero = (cropped != 0).any(0)
    assert inside.sum() > raw_nonzero.sum()
    for c in range(4):
        assert abs(float(data[c][inside].mean())) < 1e-4 and abs(float(data[c][inside].std()) - 1) < 1e-3
        assert np.all(data[c][~inside] == 0)


def test_dice_formulas():
    a = np.array([0, 1, 2, 3, 3, 0], np.uint8)
    b = np.array([0, 1, 3, 3, 0, 0], np.uint8)
    d = tiler_ref.brats_region_dice(a, b)
    assert abs(d["WT"] - 2 * 3 / (4 + 3)) < 1e-6 and abs(d["ET"] - 2 * 1 / (2 + 2)) < 1e-6
    assert tiler_ref.brats_region_dice(a, a)["mean"] == pytest.approx(1.0)


def test_resampling_restatement_decisions_and_identities():
    """Row P step 4 / row E resampling (round 4; PARITY UNPINNED: nnU-Net v1 + skimage absent, restated on scipy).  The decision
    table of resample_patient and the properties any faithful restatement has: identity at equal shape, exactness on linear
    ramps for order 1, constants preserved and range clipped for order 3, the separate-z mode equal to per-slice 2-D resizes
    followed by a nearest pick along z."""
    T = tiler_ref
    ns, sep, axis = T.resample_plan((60, 100, 100), (5.0, 1.0, 1.0), (1.0, 1.0, 1.0))
    assert ns == (300, 100, 100) and sep and list(axis) == [0]
    assert T.resample_plan((78, 120, 120), (2.0, 2.0, 2.0), (1.0, 1.0, 1.0))[:2] == ((156, 240, 240), False)
    assert T.resample_plan((30, 120, 100), (1.25, 1.25, 0.24), (1.0, 1.0, 1.0))[1] is False          # two low-resolution axes
    rs = np.random.RandomState(0)
    x = rs.standard_normal((2, 9, 11, 10)).astype(np.float32)
    assert T.resample_data_or_seg(x, (9, 11, 10), False, None, 3, False) is x
    ramp = (np.arange(10, dtype=np.float32)[None, None, None, :] * 2.0 + 1.0) * np.ones((1, 4, 5, 1), np.float32)
    up = T.resample_data_or_seg(ramp, (4, 5, 20), False, None, 1, False)
    inner = up[0, 0, 0, 1:-1]                                                                          # edge samples replicate
    assert np.allclose(np.diff(inner), 1.0, atol=1e-5)
    const = np.full((1, 6, 7, 8), 3.25, np.float32)
    assert np.allclose(T.resample_data_or_seg(const, (9, 5, 12), False, None, 3, False), 3.25, atol=1e-6)
    y = T.resample_data_or_seg(x, (13, 17, 7), False, None, 3, False)
    for c in range(2):
        assert y[c].min() >= x[c].min() - 1e-6 and y[c].max() <= x[c].max() + 1e-6                    # skimage clips to the input's range
    sep = T.resample_data_or_seg(x, (18, 16, 7), False, np.array([0]), 3, True, order_z=0)
    per_slice = np.stack([T.skimage_resize(x[0, k], (16, 7), 3) for k in range(9)], 0)
    pick = np.floor((np.arange(18) + 0.5) * 9 / 18 - 0.5 + 0.5).clip(0, 8).astype(int)
    assert np.allclose(sep[0], per_slice[pick], atol=1e-6)
    seg = (rs.uniform(size=(1, 8, 9, 10)) > 0.5).astype(np.float32) - 1.0                             # the -1 / 0 pseudo-seg of crop_to_nonzero
    out = T.resample_data_or_seg(seg, (12, 9, 15), True, None, 1, False)
    lin = T.skimage_resize((seg[0] == 0).astype(float), (12, 9, 15), 1)
    assert np.array_equal(out[0] >= 0, lin >= 0.5)
