"""Oracle: numpy restatement of the nnU-Net v1 runtime around the network.

PARITY UNPINNED (see oracle/__init__.py): the reference driver calls these through the
un-vendored ``nnunet`` package (KAIST BraTS21 fork of nnU-Net v1; no version pin in the
reference, ``batchgenerators==0.21`` per PROJECT_DOCUMENTATION.md:195), whose source is absent
from /root/reference and for which the reference holds no tests or fixtures.  Each function
restates the published nnU-Net v1 algorithm and names the reference call site that reaches it
(run_brats2021_inference_singlethread.py = "driver").  Cross-checks available here: scipy for the
Gaussian, the tile tables implied by data/temp_inference_output1.

Test infrastructure only.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import binary_fill_holes, gaussian_filter

from . import unet_ref


# --------------------------------------------------------------------------- T2
def compute_steps_for_sliding_window(patch_size, image_size, step_size):
    """nnunet SegmentationNetwork._compute_steps_for_sliding_window (driver :97-106 via predict_3D)."""
    assert all(i >= j for i, j in zip(image_size, patch_size)), "image size must be >= patch size"
    assert 0 < step_size <= 1
    target = [i * step_size for i in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step = image_size[dim] - patch_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


# --------------------------------------------------------------------------- T3
def get_gaussian(patch_size, sigma_scale=1.0 / 8):
    """nnunet SegmentationNetwork._get_gaussian."""
    tmp = np.zeros(patch_size)
    center = [i // 2 for i in patch_size]
    sigmas = [i * sigma_scale for i in patch_size]
    tmp[tuple(center)] = 1
    g = gaussian_filter(tmp, sigmas, 0, mode="constant", cval=0)
    g = g / np.max(g) * 1
    g = g.astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_to_patch(data, patch_size):
    """batchgenerators pad_nd_image(data, patch, 'constant', {'constant_values': 0}, True, None):
    pad the trailing len(patch) axes up to the patch size, split as below//above = d//2, d//2 + d%2.
    Returns (padded, slicer-lower-bounds)."""
    shape = np.array(data.shape[-len(patch_size):])
    new_shape = np.maximum(shape, np.array(patch_size))
    diff = new_shape - shape
    below = diff // 2
    above = diff // 2 + diff % 2
    pads = [(0, 0)] * (data.ndim - len(patch_size)) + [(int(b), int(a)) for b, a in zip(below, above)]
    if diff.sum() == 0:
        return data, [0] * len(patch_size)
    return np.pad(data, pads, mode="constant", constant_values=0), [int(b) for b in below]


# --------------------------------------------------------------------------- T4
def mirror_schedule(mirror_axes):
    """Order in which _internal_maybe_mirror_and_pred_3D evaluates flips: m = 0..7 with
    m=1:(4,) 2:(3,) 3:(4,3) 4:(2,) 5:(4,2) 6:(3,2) 7:(4,3,2) on NCDHW dims, each only if all the
    spatial axes it flips are in mirror_axes (axis 0 <-> dim 2).  Returns a list of dim tuples."""
    out = []
    for m in range(8):
        dims = []
        if m & 1:
            if 2 not in mirror_axes:
                continue
            dims.append(4)
        if m & 2:
            if 1 not in mirror_axes:
                continue
            dims.append(3)
        if m & 4:
            if 0 not in mirror_axes:
                continue
            dims.append(2)
        out.append(tuple(dims))
    return out


def apply_nonlin(logits, nonlin):
    if nonlin == "sigmoid":  # BraTSRegions trainers: inference_apply_nonlin = sigmoid
        return torch.sigmoid(logits)
    if nonlin == "softmax":  # generic nnU-Net: softmax_helper = F.softmax(x, 1)
        return torch.softmax(logits, 1)
    return logits


@torch.no_grad()
def mirror_and_predict(net_fn, x, mirror_axes, do_mirroring, nonlin, mult=None):
    """_internal_maybe_mirror_and_pred_3D: result += 1/num_results * flip_back(nonlin(net(flip(x))))."""
    sched = mirror_schedule(mirror_axes) if do_mirroring else [()]
    num_results = len(sched)
    result = None
    for dims in sched:
        xi = torch.flip(x, dims) if dims else x
        pred = apply_nonlin(net_fn(xi), nonlin)
        if dims:
            pred = torch.flip(pred, dims)
        term = (1.0 / num_results) * pred
        result = term if result is None else result + term
    if mult is not None:
        result = result * mult
    return result


# --------------------------------------------------------------------------- T1 / T5
@torch.no_grad()
def predict_3d_tiled(net_fn, data, patch_size, num_classes, step_size=0.5, do_mirroring=True,
                     mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid", return_parts=False):
    """_internal_predict_3D_3Dconv_tiled: data [C,Z,Y,X] fp32 -> class probabilities [K,Z,Y,X] fp32."""
    data = np.asarray(data, dtype=np.float32)
    padded, lo = pad_to_patch(data, patch_size)
    shape = padded.shape[1:]
    steps = compute_steps_for_sliding_window(patch_size, shape, step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    if use_gaussian and num_tiles > 1:
        g = get_gaussian(patch_size, 1.0 / 8)
        add = g
    else:
        g = None
        add = np.ones(patch_size, dtype=np.float32)
    agg = np.zeros((num_classes,) + tuple(shape), dtype=np.float32)
    cnt = np.zeros((num_classes,) + tuple(shape), dtype=np.float32)
    gt = None if g is None else torch.from_numpy(g)
    for x in steps[0]:
        for y in steps[1]:
            for z in steps[2]:
                sl = (slice(None), slice(x, x + patch_size[0]), slice(y, y + patch_size[1]), slice(z, z + patch_size[2]))
                patch = torch.from_numpy(np.ascontiguousarray(padded[sl][None]))
                pred = mirror_and_predict(net_fn, patch, mirror_axes, do_mirroring, nonlin, gt)[0].numpy()
                agg[sl] += pred
                cnt[sl] += add
    crop = (slice(None),) + tuple(slice(l, l + s) for l, s in zip(lo, data.shape[1:]))
    probs = (agg[crop] / cnt[crop]).astype(np.float32)
    if return_parts:
        return probs, steps, agg, cnt
    return probs


def make_net_fn(sd, cfg):
    return lambda x: unet_ref.unet_forward(sd, x, cfg)


# --------------------------------------------------------------------------- E
def regions_to_labels(probs, region_class_order=(1, 2, 3)):
    """save_segmentation_nifti_from_softmax with region_class_order (driver :144-156):
    seg = 0; for i, c in enumerate(order): seg[probs[i] > 0.5] = c."""
    seg = np.zeros(probs.shape[1:], dtype=np.uint8)
    for i, c in enumerate(region_class_order):
        seg[probs[i] > 0.5] = c
    return seg


def paste_into_original(seg, crop_bbox, original_shape):
    """Export pastes the cropped segmentation back at crop_bbox of the raw-size array."""
    out = np.zeros(original_shape, dtype=np.uint8)
    sl = tuple(slice(b[0], b[0] + s) for b, s in zip(crop_bbox, seg.shape))
    out[sl] = seg
    return out


# --------------------------------------------------------------------------- P
def crop_to_nonzero(data):
    """nnunet.preprocessing.cropping.crop_to_nonzero: mask = OR_c(data != 0) -> fill holes -> bbox.
    Returns (cropped [C,...], inside-mask of the crop (bool), bbox [[lo,hi],...])."""
    mask = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        mask |= data[c] != 0
    mask = binary_fill_holes(mask)
    idx = np.where(mask)
    bbox = [[int(np.min(i)), int(np.max(i)) + 1] for i in idx]
    sl = tuple(slice(b[0], b[1]) for b in bbox)
    return data[(slice(None),) + sl], mask[sl], bbox


def normalize_noct_masked(data, inside_mask):
    """GenericPreprocessor nonCT scheme with use_mask_for_norm=True (plans in
    data/temp_inference_output1): per channel x[m] = (x[m]-mean)/(std+1e-8), x[~m] = 0."""
    out = np.array(data, dtype=np.float32, copy=True)
    m = inside_mask
    for c in range(out.shape[0]):
        mn = out[c][m].mean()
        sd = out[c][m].std()
        out[c][m] = (out[c][m] - mn) / (sd + 1e-8)
        out[c][~m] = 0
    return out


def preprocess_case(raw):
    """trainer.preprocess_patient (driver :89) for the BraTS plans: crop, identity transpose,
    no resampling (1 mm -> 1 mm), masked z-score.  raw: [4,Z,Y,X] float32."""
    raw = np.asarray(raw, dtype=np.float32)
    cropped, inside, bbox = crop_to_nonzero(raw)
    data = normalize_noct_masked(cropped, inside)
    props = dict(crop_bbox=bbox, original_size_of_raw_data=tuple(raw.shape[1:]), size_after_cropping=tuple(cropped.shape[1:]))
    return data, props


# --------------------------------------------------------------------------- resampling (rows P step 4 and E)
# PARITY UNPINNED twice over: nnU-Net v1's resample_patient / resample_data_or_seg (un-vendored) call
# skimage.transform.resize(order, mode='edge', anti_aliasing=False) and batchgenerators' resize_segmentation, and neither
# package is in this image.  skimage >= 0.19 implements that resize as scipy.ndimage.zoom(order, mode='nearest', grid_mode=True)
# followed by a clip to the input's range; earlier releases sample the same half-pixel-centred grid through map_coordinates.
# scipy IS here, so the restatement below calls it.
RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3


def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]  # the axis (axes) with the largest spacing


def skimage_resize(img, new_shape, order):
    """skimage.transform.resize(img, new_shape, order, mode='edge', anti_aliasing=False) (clip=True, float input)."""
    from scipy.ndimage import zoom
    img = np.asarray(img, dtype=np.float64)
    factors = [float(n) / float(o) for n, o in zip(new_shape, img.shape)]
    out = zoom(img, factors, order=order, mode="nearest", grid_mode=True)
    assert tuple(out.shape) == tuple(int(v) for v in new_shape), (out.shape, new_shape)
    return np.clip(out, img.min(), img.max())


def resize_segmentation(seg, new_shape, order):
    """batchgenerators.augmentations.utils.resize_segmentation: per label a linearly resized indicator, kept where >= 0.5."""
    if order == 0:
        return skimage_resize(seg.astype(float), new_shape, 0).astype(seg.dtype)
    out = np.zeros(new_shape, dtype=seg.dtype)
    for c in np.unique(seg):
        m = skimage_resize((seg == c).astype(float), new_shape, order)
        out[m >= 0.5] = c
    return out


def resample_data_or_seg(data, new_shape, is_seg, axis=None, order=3, do_separate_z=False, order_z=0):
    """nnunet.preprocessing.preprocessing.resample_data_or_seg ([C, z, y, x] in and out)."""
    from scipy.ndimage import map_coordinates
    assert data.ndim == 4
    resize_fn = resize_segmentation if is_seg else skimage_resize
    shape = np.array(data[0].shape)
    new_shape = np.array(new_shape)
    if not np.any(shape != new_shape):
        return data
    dtype = data.dtype
    data = data.astype(float)
    if do_separate_z:
        assert len(axis) == 1
        ax = int(axis[0])
        new_2d = [int(v) for i, v in enumerate(new_shape) if i != ax]
        chans = []
        for c in range(data.shape[0]):
            slices = [resize_fn(np.take(data[c], k, axis=ax), new_2d, order) for k in range(shape[ax])]
            vol = np.stack(slices, ax)
            if shape[ax] != new_shape[ax]:
                scale = [float(o) / float(n) for o, n in zip(vol.shape, new_shape)]
                grid = np.mgrid[:new_shape[0], :new_shape[1], :new_shape[2]].astype(float)
                coords = np.array([scale[i] * (grid[i] + 0.5) - 0.5 for i in range(3)])
                assert not is_seg or order_z == 0
                vol = map_coordinates(vol, coords, order=order_z, mode="nearest")
            chans.append(vol[None])
        return np.vstack(chans).astype(dtype)
    return np.vstack([resize_fn(data[c], new_shape, order)[None] for c in range(data.shape[0])]).astype(dtype)


def resample_plan(shape, original_spacing, target_spacing, force_separate_z=None,
                  separate_z_anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """The decisions of nnunet resample_patient: (new_shape, do_separate_z, axis)."""
    new_shape = np.round((np.array(original_spacing) / np.array(target_spacing)).astype(float) * np.array(shape)).astype(int)
    if force_separate_z is not None:
        do_separate_z = force_separate_z
        axis = get_lowres_axis(original_spacing) if force_separate_z else None
    elif get_do_separate_z(original_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(original_spacing)
    elif get_do_separate_z(target_spacing, separate_z_anisotropy_threshold):
        do_separate_z, axis = True, get_lowres_axis(target_spacing)
    else:
        do_separate_z, axis = False, None
    if axis is not None and len(axis) != 1:   # (3: isotropic; 2: e.g. (0.24, 1.25, 1.25) - not resampled separately)
        do_separate_z = False
    return tuple(int(v) for v in new_shape), bool(do_separate_z), axis


def preprocess_case_resampled(raw, original_spacing, target_spacing):
    """trainer.preprocess_patient (driver :89) INCLUDING step 4: crop, resample data (order 3) and the inside mask (order 1; the
    low-resolution axis separately with order 0 when the spacing is anisotropic beyond 3), masked z-score on the new grid."""
    raw = np.asarray(raw, dtype=np.float32)
    cropped, inside, bbox = crop_to_nonzero(raw)
    seg = np.where(inside, 0, -1).astype(np.float32)[None]
    new_shape, sep, axis = resample_plan(cropped.shape[1:], original_spacing, target_spacing)
    data = resample_data_or_seg(cropped, new_shape, False, axis, 3, sep, order_z=0)
    seg = resample_data_or_seg(seg, new_shape, True, axis, 1, sep, order_z=0)
    data = normalize_noct_masked(data, seg[0] >= 0)
    props = dict(crop_bbox=bbox, original_size_of_raw_data=tuple(raw.shape[1:]), size_after_cropping=tuple(cropped.shape[1:]),
                 size_after_resampling=tuple(new_shape), original_spacing=tuple(float(v) for v in original_spacing),
                 spacing_after_resampling=tuple(float(v) for v in target_spacing))
    return data, props


def export_resample_probs(probs, props, order=1, order_z=0):
    """The resampling inside save_segmentation_nifti_from_softmax(order=1, force_separate_z=None, interpolation_order_z=0)
    (driver :131-138, :144-156): class probabilities on the resampled grid -> the shape after cropping."""
    target = tuple(props["size_after_cropping"])
    if tuple(probs.shape[1:]) == target:
        return probs
    if get_do_separate_z(props["original_spacing"]):
        sep, axis = True, get_lowres_axis(props["original_spacing"])
    elif get_do_separate_z(props["spacing_after_resampling"]):
        sep, axis = True, get_lowres_axis(props["spacing_after_resampling"])
    else:
        sep, axis = False, None
    if axis is not None and len(axis) != 1:
        sep = False
    return resample_data_or_seg(probs, target, False, axis, order, sep, order_z=order_z)


# --------------------------------------------------------------------------- metric
def dice(a, b, eps=1e-8):
    """evaluate_segmentation.py:181-195 calculate_metrics_binary Dice."""
    a = a.astype(bool)
    b = b.astype(bool)
    inter = np.logical_and(a, b).sum()
    return float((2.0 * inter + eps) / (a.sum() + b.sum() + eps))


def brats_region_dice(pred, ref):
    """Mean of WT/TC/ET Dice on raw nnU-Net labels {1: ED, 2: NCR, 3: ET}:
    WT = {1,2,3}, TC = {2,3}, ET = {3}  (evaluate_segmentation.py:129-162 after label conversion)."""
    wt = dice(pred > 0, ref > 0)
    tc = dice(pred >= 2, ref >= 2)
    et = dice(pred == 3, ref == 3)
    return dict(WT=wt, TC=tc, ET=et, mean=(wt + tc + et) / 3.0)
