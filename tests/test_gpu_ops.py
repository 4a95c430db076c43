"""-m gpu: single-kernel parity through the C ABI against torch-CPU fp32 (the reference's own
backend, L0 in SURVEY.md section 1).  Tolerances are for fp32 summation-order differences only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(rs, *shape):
    return rs.standard_normal(shape).astype(np.float32)


def _ref_conv(x_ndhwc, w, b, stride, act, slope):
    x = torch.from_numpy(x_ndhwc).permute(0, 4, 1, 2, 3).contiguous()
    y = F.conv3d(x, torch.from_numpy(w), None if b is None else torch.from_numpy(b), stride=stride, padding=1)
    if act:
        y = F.leaky_relu(y, slope)
    return y.permute(0, 2, 3, 4, 1).contiguous().numpy()


CONV_CASES = [
    # n, d, h, w, cin, cout, stride, act
    (1, 16, 16, 32, 32, 32, 1, 0),     # the dominant shape class (Cin=Cout=32), one x-row of tiles
    (2, 8, 12, 40, 16, 64, 1, 1),      # ragged dims (partial tiles in y and x), NF=2, fused LeakyReLU, batch
    (1, 8, 8, 8, 8, 32, 1, 0),         # stem-like Cin=8 (CC=8 path)
    (1, 4, 4, 4, 64, 96, 1, 1),        # tiny volume, Cout=96 (NF=1, 3 cout blocks)
    (1, 16, 16, 32, 32, 64, 2, 0),     # stride 2
    (2, 10, 6, 14, 64, 32, 2, 1),      # stride 2, ragged / odd output dims
    (1, 32, 32, 32, 64, 64, 1, 0),     # several chunks (4 x 16 channels), many tiles
    (2, 8, 12, 40, 4, 32, 1, 1),       # the network's first layer: Cin = 4 -> stem kernel (x-taps folded into K), ragged
    (1, 16, 16, 32, 4, 64, 1, 0),      # stem with two cout blocks
    (3, 9, 7, 131, 40, 96, 2, 1),      # stride 2: odd input dims, ragged x tiles, 5 chunks of 8, 3 cout blocks
    (1, 64, 64, 64, 32, 64, 2, 1),     # stride 2 at a network-like size
    (2, 64, 64, 128, 32, 64, 2, 1),    # stride 2, >= 1024 tiles: the persistent LDS-DMA stride-2 kernel
    (3, 50, 62, 90, 16, 128, 2, 0),    # the same kernel: ragged in z, y, x (odd input dims), 2 chunks, two cout blocks, batch
    (8, 32, 30, 32, 128, 256, 2, 1),   # the same kernel on a narrow volume (Wo = 16): 2 x 4 x 16 tiles, ragged y, 16 chunks
    (8, 4, 16, 16, 128, 128, 1, 1),    # one z tile (both z borders in every brick), 2 x 2 tiles in y, x: 128 units of 8 chunks -> F(2x2x2,3x3x3)
    (4, 8, 8, 24, 128, 160, 1, 0),     # the same rule on an odd tile grid (2 x 1 x 3 tiles, linear order), 5 cout blocks, batch 4 (120 units), no activation
    (8, 8, 8, 8, 320, 320, 1, 1),      # deep level: 160 (tile, cout block) units of 20 chunks -> F(2x2x2,3x3x3) (MI355_WINO3=0: split-K over the 40 chunks)
    (8, 16, 16, 16, 256, 320, 2, 0),   # deep stride-2 level: split-K (4 slices), no activation
    (2, 4, 4, 4, 320, 320, 1, 1),      # bottleneck-sized launch: split-K with a 64-voxel volume in 256-voxel tiles
    (1, 64, 64, 64, 32, 64, 1, 1),     # >= 512 workgroups: the Winograd kernel (F(2x2,3x3) by default), two cout blocks
    (3, 30, 37, 70, 48, 32, 1, 0),     # Winograd kernel: ragged in z, y (odd: half-used row pair) and x, 3 chunks, batch
    (2, 16, 64, 128, 16, 32, 1, 1),    # Winograd kernel: ONE 16-channel chunk per tile, 4 x 16 x 4 tiles per sample -> blocked tile order 4 x 4 x 2
    (1, 24, 64, 256, 32, 32, 1, 1),    # Winograd kernel: 8 x-tiles and 6 z-tiles (not a power of two -> linear tile order), whole-line stores
    (2, 32, 64, 64, 48, 64, 1, 1),     # F(2x2x2,3x3x3) kernel: whole 4 x 8 x 8 tiles, 3 chunks, two cout blocks, batch
    (2, 30, 64, 128, 32, 32, 1, 0),     # z not a multiple of 4: stays on the F(2x2,3x3) kernel
    (1, 64, 64, 72, 16, 32, 1, 1),     # F(2x2x2,3x3x3): one chunk per tile, 9 x-tiles (linear tile order), 1152 tiles on 256 workgroups (uneven)
]

#: which kernel an fp32 case is written for (asserted through mi355_last_conv_kernel)
F32_EXPECT_KERNEL = {
    (1, 64, 64, 64, 32, 64, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (2, 16, 64, 128, 16, 32, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (1, 24, 64, 256, 32, 32, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (2, 32, 64, 64, 48, 64, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (1, 64, 64, 72, 16, 32, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (3, 30, 37, 70, 48, 32, 1, 0): "conv3_f32_wino2_kernel<0>",
    (2, 30, 64, 128, 32, 32, 1, 0): "conv3_f32_wino2_kernel<0>",
    (2, 64, 64, 128, 32, 64, 2, 1): "conv3_f32_s2dma_kernel<5>",
    (8, 4, 16, 16, 128, 128, 1, 1): "conv3_f32_wino3_kernel<0, false>",
    (4, 8, 8, 24, 128, 160, 1, 0): "conv3_f32_wino3_kernel<0, false>",
    (8, 8, 8, 8, 320, 320, 1, 1): "conv3_f32_wino3_kernel<0, false>",   # the 8^3 level of a batch of eight tiles: 160 units of 20 chunks
}


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_mfma_matches_torch(amd, gpu, case):
    n, d, h, w, cin, cout, stride, act = case
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = _rand(rs, n, d, h, w, cin)
    wt = (_rand(rs, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)
    b = _rand(rs, cout)
    ref = _ref_conv(x, wt, b, stride, act, 0.01)
    y = amd.ops.conv3d_ndhwc(torch.from_numpy(x).to(gpu), wt, b, stride=stride, act=act, slope=0.01, impl="mfma")
    ran = amd.ops.last_conv_kernel()
    y = y.cpu().numpy()
    assert y.shape == ref.shape
    err = np.abs(y - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), f"max abs err {err} ({ran})"
    import os
    if case in F32_EXPECT_KERNEL and not any(k in os.environ for k in ("MI355_CONV_IMPL", "MI355_WINOGRAD", "MI355_WINO3", "MI355_S2_DMA", "MI355_SPLITK")):
        assert ran == F32_EXPECT_KERNEL[case], ran


@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[5], (1, 6, 5, 7, 5, 7, 1, 1)])
def test_conv3d_direct_matches_torch(amd, gpu, case):
    n, d, h, w, cin, cout, stride, act = case
    rs = np.random.RandomState(11)
    x = _rand(rs, n, d, h, w, cin)
    if cin % 4:  # NDHWC tensors are channel-padded to a multiple of 4 by their producers
        pad = 4 - cin % 4
        xp = np.concatenate([x, np.zeros((n, d, h, w, pad), np.float32)], -1)
    else:
        xp = x
    wt = (_rand(rs, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)
    ref = _ref_conv(x, wt, None, stride, act, 0.01)
    wt_p = np.zeros((cout, xp.shape[-1], 3, 3, 3), np.float32)
    wt_p[:, :cin] = wt
    y = amd.ops.conv3d_ndhwc(torch.from_numpy(xp).to(gpu), wt_p, None, stride=stride, act=act, impl="direct").cpu().numpy()
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_conv3d_mfma_identity_asymmetric(amd, gpu):
    """Exact-integer check of the MFMA operand / accumulator maps: a centre-tap weight that maps
    channel c -> output (3c+1) % 32 with distinct integer gains must reproduce the input exactly."""
    rs = np.random.RandomState(3)
    x = rs.randint(-8, 9, size=(1, 8, 8, 32, 32)).astype(np.float32)
    wt = np.zeros((32, 32, 3, 3, 3), np.float32)
    for c in range(32):
        wt[(3 * c + 1) % 32, c, 1, 1, 1] = float(c + 1)
    y = amd.ops.conv3d_ndhwc(torch.from_numpy(x).to(gpu), wt, None).cpu().numpy()
    ref = np.zeros_like(y)
    for c in range(32):
        ref[..., (3 * c + 1) % 32] = x[..., c] * (c + 1)
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("case", [(1, 4, 4, 4, 32, 32), (2, 3, 5, 6, 64, 32), (1, 8, 8, 8, 320, 320), (1, 2, 2, 2, 256, 512)])
def test_tconv_matches_torch(amd, gpu, case):
    n, d, h, w, cin, cout = case
    rs = np.random.RandomState(5)
    x = _rand(rs, n, d, h, w, cin)
    wt = (_rand(rs, cin, cout, 2, 2, 2) / np.sqrt(cin)).astype(np.float32)
    ref = F.conv_transpose3d(torch.from_numpy(x).permute(0, 4, 1, 2, 3), torch.from_numpy(wt), None, stride=2)
    ref = ref.permute(0, 2, 3, 4, 1).contiguous().numpy()
    y = amd.ops.tconv3d_ndhwc(torch.from_numpy(x).to(gpu), wt).cpu().numpy()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", [(1, 33, 63, 65, 64, 32), (2, 32, 32, 64, 64, 32)])
def test_tconv_f32_persistent_kernel_matches_torch(amd, gpu, case):
    """Round 4: tconv2_f32_mfma_v3_kernel (persistent, the wave's weights in registers; Cin 64 and >= 1024 tiles of 128 input
    voxels - the largest decoder level of the fp32 path; the Cin = 128 instantiation is behind MI355_TCONV_V3=2).  One case is ragged (33 x 63 x 65 = 135135 voxels: the clamped tail
    tile and its store predicate); the kernel that ran is asserted."""
    n, d, h, w, cin, cout = case
    rs = np.random.RandomState(16)
    x = _rand(rs, n, d, h, w, cin)
    wt = (_rand(rs, cin, cout, 2, 2, 2) / np.sqrt(cin)).astype(np.float32)
    ref = F.conv_transpose3d(torch.from_numpy(x).permute(0, 4, 1, 2, 3), torch.from_numpy(wt), None, stride=2)
    ref = ref.permute(0, 2, 3, 4, 1).contiguous().numpy()
    y = amd.ops.tconv3d_ndhwc(torch.from_numpy(x).to(gpu), wt).cpu().numpy()
    if "MI355_TCONV_V3" not in os.environ and "MI355_TCONV_V1" not in os.environ:
        assert amd.ops.last_conv_kernel().startswith("tconv2_f32_mfma_v3_kernel<"), amd.ops.last_conv_kernel()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_label_ensemble_truth_table(amd, gpu):
    """np.round((s1+s2)/2) - half to even (reference driver :305)."""
    a, b = np.meshgrid(np.arange(5, dtype=np.uint8), np.arange(5, dtype=np.uint8), indexing="ij")
    want = np.round((a.astype(np.float64) + b.astype(np.float64)) / 2.0).astype(np.uint8)
    got = amd.ops.label_ensemble(torch.from_numpy(a.copy()).to(gpu), torch.from_numpy(b.copy()).to(gpu)).cpu().numpy()
    assert np.array_equal(got, want)
    assert want[0, 3] == 2 and want[0, 1] == 0 and want[1, 2] == 2 and want[2, 3] == 2


def test_regions_to_labels_and_paste(amd, gpu):
    rs = np.random.RandomState(9)
    probs = rs.uniform(0, 1, size=(3, 5, 6, 7)).astype(np.float32)
    probs[0, 0, 0, 0] = 0.5  # not > 0.5
    seg = np.zeros((5, 6, 7), np.uint8)
    for i, c in enumerate((1, 2, 3)):
        seg[probs[i] > 0.5] = c
    want = np.zeros((9, 8, 10), np.uint8)
    want[2:7, 1:7, 3:10] = seg
    got = amd.ops.regions_to_labels(torch.from_numpy(probs).to(gpu), (1, 2, 3), (2, 1, 3), (9, 8, 10)).cpu().numpy()
    assert np.array_equal(got, want)


def test_zscore_masked(amd, gpu):
    rs = np.random.RandomState(2)
    vol = (rs.standard_normal((4, 9, 10, 11)) * 300 + 1000).astype(np.float32)
    mask = rs.uniform(size=(9, 10, 11)) > 0.3
    want = vol.copy()
    for c in range(4):
        mn, sd = want[c][mask].mean(), want[c][mask].std()
        want[c][mask] = (want[c][mask] - mn) / (sd + 1e-8)
        want[c][~mask] = 0
    got = amd.ops.zscore_masked_(torch.from_numpy(vol).to(gpu), torch.from_numpy(mask.astype(np.uint8)).to(gpu)).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5
    assert np.all(got[:, ~mask] == 0)


def test_compute_steps_table(amd):
    """Host helper, no GPU needed for the call itself; kept here because it goes through the .so."""
    assert amd.ops.compute_steps(128, 140, 0.5) == [0, 12]
    assert amd.ops.compute_steps(128, 240, 0.5) == [0, 56, 112]
    assert amd.ops.compute_steps(128, 128, 0.5) == [0]


# --------------------------------------------------------------------------- fp16 storage kernels
F16_CONV_CASES = [
    (1, 8, 8, 32, 32, 32, 1, 0),
    (2, 8, 12, 40, 16, 64, 1, 1),
    (1, 16, 16, 32, 32, 64, 2, 0),
    (2, 10, 6, 14, 64, 32, 2, 1),
    (1, 32, 32, 32, 64, 64, 1, 0),
    (8, 16, 16, 32, 32, 32, 1, 1),   # enough tiles for the 512-voxel (MF=4) pipelined variant
    (2, 8, 12, 40, 4, 32, 1, 1),     # first layer, Cin = 4: stem kernel
    (1, 16, 16, 32, 4, 64, 1, 0),
    (3, 9, 7, 131, 48, 96, 2, 1),    # stride 2: odd input dims, ragged x tiles, 3 chunks of 16, 3 cout blocks
    (1, 64, 64, 64, 32, 64, 2, 1),   # stride 2 at a network-like size
    (2, 64, 64, 128, 32, 64, 2, 1),  # stride 2, Cout = 64: the LDS-DMA stride-2 kernel with 64 couts per workgroup (round 4)
    (3, 50, 62, 90, 16, 128, 2, 0),  # the same kernel: ragged in z, y, x (odd input dims), two cout blocks, batch
    (8, 32, 30, 32, 128, 256, 2, 1), # the same kernel on a narrow volume (Wo = 16): 2 x 4 x 16 tiles, 8 chunks
    (8, 8, 8, 8, 320, 320, 1, 1),    # deep level: split-K over the 20 channel chunks + finishing pass
    (8, 16, 16, 16, 256, 320, 2, 0), # deep stride-2 level: split-K, no activation
    (2, 4, 4, 4, 320, 320, 1, 1),    # bottleneck-sized launch
    (8, 32, 32, 32, 64, 64, 1, 1),   # Cout % 64 == 0, volume a whole number of 8^3 tiles, >= 512 tiles: the LDS-DMA kernel
    (2, 32, 64, 64, 48, 128, 1, 0),  # the same kernel: 3 chunks, two cout blocks, no activation
    (8, 16, 16, 16, 256, 256, 1, 1), # the same kernel at the 16^3 level: 64 tiles x 4 cout blocks = one workgroup per CU, 16 chunks (round 4: threshold 512 -> 256 units)
    (4, 24, 40, 72, 16, 64, 1, 1),   # the same kernel: one chunk, odd tile counts (3 x 5 x 9 -> linear tile order), 540 tiles (not a multiple of 8: uneven XCD split)
    (2, 64, 64, 64, 32, 128, 2, 1),  # round 3: stride 2, Cout % 128 == 0, whole 4 x 4 x 8 output tiles -> conv3_f16_s2dma_kernel, 2 chunks
    (2, 32, 32, 64, 64, 256, 2, 0),  # the same kernel: two cout blocks of 128, 4 chunks, no activation
    (6, 24, 40, 48, 16, 128, 2, 1),  # the same kernel: one chunk, 3 x 5 x 3 tiles (linear tile order), 270 tiles (uneven XCD split), batch 6
    (8, 32, 32, 32, 32, 32, 1, 1),   # round 5: Cout = 32 on whole 8^3 tiles, 512 tiles = two workgroups per CU: conv3_f16_c32_kernel, 2 chunks
    (8, 32, 32, 32, 64, 32, 1, 0),   # the same kernel: 4 chunks, no activation
    (4, 24, 40, 72, 16, 32, 1, 1),   # the same kernel: one chunk, odd tile counts (3 x 5 x 9, linear tile order), 540 tiles (uneven XCD split)
]

#: which kernel a case is written for (asserted through mi355_last_conv_kernel; ADVICE r2: a case that claims a kernel must run on it)
F16_EXPECT_KERNEL = {
    (8, 32, 32, 32, 64, 64, 1, 1): "conv3_f16_dma_kernel<false, false>",
    (2, 32, 64, 64, 48, 128, 1, 0): "conv3_f16_dma_kernel<false, false>",
    (4, 24, 40, 72, 16, 64, 1, 1): "conv3_f16_dma_kernel<false, false>",
    (8, 16, 16, 16, 256, 256, 1, 1): "conv3_f16_dma_kernel<false, false>",
    (2, 64, 64, 64, 32, 128, 2, 1): "conv3_f16_s2dma_kernel<false, 128>",
    (2, 32, 32, 64, 64, 256, 2, 0): "conv3_f16_s2dma_kernel<false, 128>",
    (6, 24, 40, 48, 16, 128, 2, 1): "conv3_f16_s2dma_kernel<false, 128>",
    (2, 64, 64, 128, 32, 64, 2, 1): "conv3_f16_s2dma_kernel<false, 64>",
    (8, 8, 8, 8, 320, 320, 1, 1): "conv3_f16_mfma_kernel<1, 2, 2> split-K",
    (8, 32, 32, 32, 32, 32, 1, 1): "conv3_f16_c32_kernel<false, false, false>",
    (8, 32, 32, 32, 64, 32, 1, 0): "conv3_f16_c32_kernel<false, false, false>",
    (4, 24, 40, 72, 16, 32, 1, 1): "conv3_f16_c32_kernel<false, false, false>",
    (8, 16, 16, 32, 32, 32, 1, 1): "conv3_f16_mfma_pipe_kernel<2, 1, false, false, 1, true>",   # (128 tiles: too few for the Cout = 32 LDS-DMA kernel)
}


@pytest.mark.parametrize("case", F16_CONV_CASES)
def test_conv3d_f16_matches_torch(amd, gpu, case):
    """fp16 inputs/weights, fp32 accumulate: compared with fp32 torch on the SAME fp16-rounded operands,
    so the only differences are summation order and the final rounding of the output to fp16."""
    n, d, h, w, cin, cout, stride, act = case
    rs = np.random.RandomState(17)
    x = _rand(rs, n, d, h, w, cin).astype(np.float16)
    wt = (_rand(rs, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float16)
    b = _rand(rs, cout)
    ref = _ref_conv(x.astype(np.float32), wt.astype(np.float32), b, stride, act, 0.01)
    y = amd.ops.conv3d_ndhwc(torch.from_numpy(x).to(gpu), wt.astype(np.float32), b, stride=stride, act=act, slope=0.01)
    ran = amd.ops.last_conv_kernel()
    y = y.float().cpu().numpy()
    assert y.shape == ref.shape
    err = np.abs(y - ref).max()
    assert err <= 2e-3 * max(1.0, np.abs(ref).max()), f"max abs err {err} ({ran})"
    import os
    if case in F16_EXPECT_KERNEL and not any(k in os.environ for k in ("MI355_CONV_IMPL", "MI355_F16_DMA", "MI355_F16_C32", "MI355_F16_S2", "MI355_S2_DMA", "MI355_SPLITK")):
        assert ran == F16_EXPECT_KERNEL[case], ran


# Norm statistics epilogue (sum y, sum y^2 per sample and channel: what InstanceNorm / GroupNorm reduce the conv output to,
# generic_UNet.py:62-72), one case per kernel instantiation that carries it.  (n, d, h, w, cin, cout, stride, act, dtype)
SUMS_CASES = [
    (8, 32, 32, 32, 64, 64, 1, 0, "f16"),    # LDS-DMA kernel, statistics, no activation (conv -> norm -> LeakyReLU)
    (4, 24, 40, 72, 16, 64, 1, 1, "f16"),    # the same kernel with an activation in front of the statistics (ConvDropoutNonlinNorm)
    (2, 64, 64, 64, 32, 128, 2, 0, "f16"),   # stride-2 LDS-DMA kernel
    (2, 64, 64, 64, 32, 64, 2, 0, "f16"),    # the same kernel with 64 couts per workgroup (waves split 2 x 2: couts x z planes)
    (8, 16, 16, 32, 32, 32, 1, 0, "f16"),    # register-staged kernel, 512-voxel tiles, Cout = 32
    (8, 32, 32, 32, 32, 32, 1, 0, "f16"),    # round 5: the Cout = 32 LDS-DMA kernel, statistics, no activation
    (4, 24, 40, 72, 16, 32, 1, 1, "f16"),    # the same kernel with an activation in front of the statistics, odd tile counts
    (2, 8, 12, 40, 16, 64, 1, 1, "f16"),     # the same family, ragged tiles: voxels beyond the edge must not count
    (3, 9, 7, 131, 48, 96, 2, 0, "f16"),     # stride 2, odd dims, ragged
    (2, 8, 12, 40, 4, 32, 1, 0, "f16"),      # first layer (Cin = 4), ragged in y
    (1, 16, 16, 32, 4, 64, 1, 1, "f16"),     # first layer, two cout blocks
    (8, 32, 32, 32, 32, 32, 1, 0, "f32"),    # F(2x2x2,3x3x3) Winograd kernel, statistics instantiation
    (8, 30, 32, 32, 32, 32, 1, 0, "f32"),    # F(2x2,3x3) Winograd kernel (z not a multiple of 4), statistics instantiation
    (2, 32, 64, 64, 48, 64, 1, 1, "f32"),    # F(2x2x2,3x3x3), statistics behind an activation, two cout blocks, 3 chunks
    (2, 16, 16, 64, 16, 64, 1, 1, "f32"),    # direct f32 kernels
    (2, 10, 6, 14, 64, 32, 2, 0, "f32"),     # stride 2, ragged
    (2, 64, 64, 128, 32, 64, 2, 0, "f32"),   # persistent stride-2 kernel
    (2, 8, 12, 40, 4, 32, 1, 0, "f32"),      # first layer
]
SUMS_EXPECT_KERNEL = {
    (8, 32, 32, 32, 64, 64, 1, 0, "f16"): "conv3_f16_dma_kernel<true, false>",
    (4, 24, 40, 72, 16, 64, 1, 1, "f16"): "conv3_f16_dma_kernel<true, false>",
    (8, 32, 32, 32, 32, 32, 1, 0, "f16"): "conv3_f16_c32_kernel<true, false, false>",
    (4, 24, 40, 72, 16, 32, 1, 1, "f16"): "conv3_f16_c32_kernel<true, false, false>",
    (2, 64, 64, 64, 32, 128, 2, 0, "f16"): "conv3_f16_s2dma_kernel<true, 128>",
    (2, 64, 64, 64, 32, 64, 2, 0, "f16"): "conv3_f16_s2dma_kernel<true, 64>",
    (8, 32, 32, 32, 32, 32, 1, 0, "f32"): "conv3_f32_wino3_kernel<2, false>",
    (8, 30, 32, 32, 32, 32, 1, 0, "f32"): "conv3_f32_wino2_kernel<2>",
    (2, 32, 64, 64, 48, 64, 1, 1, "f32"): "conv3_f32_wino3_kernel<2, false>",
    (2, 64, 64, 128, 32, 64, 2, 0, "f32"): "conv3_f32_s2dma_kernel<5>",
}


@pytest.mark.parametrize("case", SUMS_CASES)
def test_conv3d_norm_sums_match_reference(amd, gpu, case):
    """The statistics the conv kernels accumulate for the run-time norms, against fp64 sums of the reference conv on the
    same operands.  The kernels sum fp32 values in fp32 partials of 32-512 voxels, quantise each partial (common.h:
    2^-40 V for sum y, 2^-44 V for sum y^2) and add the partials exactly in fp64: expected agreement is fp32 rounding of a
    partial, ~1e-6 relative to the channel's rms; the gates below (1e-4 of the rms for the mean, 1e-4 relative for the mean
    square) leave room for the MFMA's own summation order inside y."""
    n, d, h, w, cin, cout, stride, act, dt = case
    rs = np.random.RandomState(23)
    x = (_rand(rs, n, d, h, w, cin) + 0.25).astype(np.float32)  # (non-zero mean: sum y must not be a sum of cancelling terms only)
    wt = (_rand(rs, cout, cin, 3, 3, 3) / np.sqrt(cin * 27)).astype(np.float32)
    b = _rand(rs, cout)
    if dt == "f16":
        x = x.astype(np.float16); wt = wt.astype(np.float16).astype(np.float32)
    ref = _ref_conv(x.astype(np.float32), wt, b, stride, act, 0.01).astype(np.float64)
    y, sums = amd.ops.conv3d_sums_ndhwc(torch.from_numpy(x).to(gpu), wt, b, stride=stride, act=act, slope=0.01)
    ran = amd.ops.last_conv_kernel()
    y = y.float().cpu().numpy(); sums = sums.cpu().numpy()
    assert np.abs(y - ref).max() <= (2e-3 if dt == "f16" else 2e-5) * max(1.0, np.abs(ref).max()), ran
    V = ref.shape[1] * ref.shape[2] * ref.shape[3]
    mean_ref = ref.sum(axis=(1, 2, 3)) / V
    msq_ref = (ref * ref).sum(axis=(1, 2, 3)) / V
    rms = np.sqrt(msq_ref)
    mean_err = np.abs(sums[..., 0] / V - mean_ref) / rms
    msq_err = np.abs(sums[..., 1] / V - msq_ref) / msq_ref
    assert mean_err.max() <= 1e-4 and msq_err.max() <= 1e-4, f"mean err {mean_err.max():.2e} rms, mean-square err {msq_err.max():.2e} ({ran})"
    import os
    if case in SUMS_EXPECT_KERNEL and not any(k in os.environ for k in ("MI355_CONV_IMPL", "MI355_F16_DMA", "MI355_F16_C32", "MI355_F16_S2", "MI355_S2_DMA", "MI355_WINOGRAD", "MI355_WINO3")):
        assert ran == SUMS_EXPECT_KERNEL[case], ran


def test_conv3d_f16_identity_asymmetric(amd, gpu):
    rs = np.random.RandomState(3)
    x = rs.randint(-8, 9, size=(1, 8, 8, 32, 32)).astype(np.float16)
    wt = np.zeros((32, 32, 3, 3, 3), np.float32)
    for c in range(32):
        wt[(3 * c + 1) % 32, c, 1, 1, 1] = float(c % 7 + 1)
    y = amd.ops.conv3d_ndhwc(torch.from_numpy(x).to(gpu), wt, None).float().cpu().numpy()
    ref = np.zeros_like(y)
    for c in range(32):
        ref[..., (3 * c + 1) % 32] = x[..., c].astype(np.float32) * (c % 7 + 1)
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("case", [(1, 4, 4, 4, 32, 32), (2, 3, 5, 6, 64, 32), (1, 2, 2, 2, 256, 512)])
def test_tconv_f16_matches_torch(amd, gpu, case):
    n, d, h, w, cin, cout = case
    rs = np.random.RandomState(5)
    x = _rand(rs, n, d, h, w, cin).astype(np.float16)
    wt = (_rand(rs, cin, cout, 2, 2, 2) / np.sqrt(cin)).astype(np.float16)
    ref = F.conv_transpose3d(torch.from_numpy(x.astype(np.float32)).permute(0, 4, 1, 2, 3),
                             torch.from_numpy(wt.astype(np.float32)), None, stride=2)
    ref = ref.permute(0, 2, 3, 4, 1).contiguous().numpy()
    y = amd.ops.tconv3d_ndhwc(torch.from_numpy(x).to(gpu), wt.astype(np.float32)).float().cpu().numpy()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", [(1, 33, 63, 65, 32, 32), (2, 32, 32, 64, 64, 32), (1, 32, 64, 64, 128, 64)])
def test_tconv_f16_persistent_kernel_matches_torch(amd, gpu, case):
    """the persistent tconv2_f16_mfma_v3_kernel (every fp16 decoder level with Cin 32 / 64 / 128 and >= 1024 tiles of
    128 input voxels) had no single-op case - the shapes above dispatch to v2.  These have >= 131072 input voxels, one of them
    ragged (33 x 63 x 65 = 135135 voxels, not a multiple of 128: the clamped tail tile and its store predicate), and assert
    that v3 is the kernel that ran."""
    n, d, h, w, cin, cout = case
    rs = np.random.RandomState(15)
    x = _rand(rs, n, d, h, w, cin).astype(np.float16)
    wt = (_rand(rs, cin, cout, 2, 2, 2) / np.sqrt(cin)).astype(np.float16)
    ref = F.conv_transpose3d(torch.from_numpy(x.astype(np.float32)).permute(0, 4, 1, 2, 3),
                             torch.from_numpy(wt.astype(np.float32)), None, stride=2)
    ref = ref.permute(0, 2, 3, 4, 1).contiguous().numpy()
    y = amd.ops.tconv3d_ndhwc(torch.from_numpy(x).to(gpu), wt.astype(np.float32)).float().cpu().numpy()
    assert amd.ops.last_conv_kernel().startswith("tconv2_f16_mfma_v3_kernel<"), amd.ops.last_conv_kernel()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape,seed", [((155, 240, 240), 1000), ((40, 56, 48), 5), ((33, 17, 29), 6)])
def test_crop_mask_matches_scipy_fill_holes(amd, gpu, shape, seed):
    """crop_to_nonzero on the device (nonzero mask, hole filling by border flood fill, bounding box): bit-exact against the
    oracle's numpy + scipy.ndimage.binary_fill_holes restatement, on the bench volume (interior zero holes) and on volumes
    with hand-made topology: enclosed cavities, a cavity open to the border through a tunnel, tissue touching the border."""
    from oracle import tiler_ref
    vol = amd.synthetic.make_volume(seed=seed, shape=shape).astype(np.float32)
    rs = np.random.RandomState(seed)
    z, y, x = shape
    if z < 100:
        vol[:, z // 2 - 3:z // 2 + 3, y // 2 - 4:y // 2 + 4, x // 2 - 5:x // 2 + 5] = 0          # enclosed cavity
        vol[:, z // 3, y // 3, :x // 2] = 0                                                       # tunnel to the border ...
        vol[:, z // 3 - 2:z // 3 + 2, y // 3 - 2:y // 3 + 2, x // 2 - 3:x // 2] = 0               # ... into a cavity: stays open
        vol[:, 0, :5, :5] = rs.uniform(1, 2, size=(vol.shape[0], 5, 5))                            # tissue on the border
    cropped, inside, bbox = tiler_ref.crop_to_nonzero(vol)
    mask, box = amd.ops.crop_mask(torch.from_numpy(vol).to(gpu))
    assert box == bbox
    sl = tuple(slice(lo, hi) for lo, hi in bbox)
    assert np.array_equal(mask.cpu().numpy()[sl].astype(bool), inside)
    assert mask.cpu().numpy().sum() == inside.sum()                                                # nothing outside the box
    # and the whole device preprocessing against the oracle's (crop + masked z-score)
    data, props = amd.preprocessing.preprocess_case(vol)
    want, wprops = tiler_ref.preprocess_case(vol)
    assert props["crop_bbox"] == wprops["crop_bbox"] and tuple(data.shape) == want.shape
    assert np.abs(data.cpu().numpy() - want).max() <= 2e-5


_SWITCH_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import brats_amd as amd
import torch.nn.functional as F
rs = np.random.RandomState(3)
worst = 0.0
for (n, d, h, w, cin, cout, stride) in [(1, 64, 64, 64, 32, 64, 1), (2, 64, 64, 128, 32, 64, 2), (8, 8, 8, 8, 320, 320, 1)]:
    x = rs.standard_normal((n, d, h, w, cin)).astype(np.float32)
    wt = (rs.standard_normal((cout, cin, 3, 3, 3)) / np.sqrt(cin * 27)).astype(np.float32)
    b = rs.standard_normal(cout).astype(np.float32)
    ref = F.leaky_relu(F.conv3d(torch.from_numpy(x).permute(0, 4, 1, 2, 3), torch.from_numpy(wt), torch.from_numpy(b), stride=stride, padding=1), 0.01)
    ref = ref.permute(0, 2, 3, 4, 1).numpy()
    for dt in ("f32", "f16"):
        if dt == "f32":
            y = amd.ops.conv3d_ndhwc(torch.from_numpy(x).cuda(), wt, b, stride=stride, act=1, slope=0.01, impl="mfma").cpu().numpy()
            tol = 2e-5
        else:
            y = amd.ops.conv3d_ndhwc(torch.from_numpy(x.astype(np.float16)).cuda(), wt, b, stride=stride, act=1, slope=0.01).float().cpu().numpy()
            tol = 6e-3
        err = float(np.abs(y - ref).max() / max(1.0, np.abs(ref).max()))
        assert err <= tol, (dt, n, d, cin, cout, stride, err)
        worst = max(worst, err)
print("OK", worst)
"""


@pytest.mark.parametrize("env", [{"MI355_WINOGRAD": "0"}, {"MI355_WINOGRAD": "1"}, {"MI355_WINO3": "0"}, {"MI355_S2_DMA": "0", "MI355_SPLITK": "0"},
                                 {"MI355_CONV_IMPL": "0"}, {"MI355_F16_DMA": "0"}])
def test_conv_dispatch_switches_keep_working(amd, gpu, env):
    """The A/B switches select the older kernels behind the same entry points (direct instead of Winograd, simple
    instead of the stride-2 DMA / split-K kernels, register-staged instead of LDS-DMA in fp16); they are read once per
    process, so each setting runs in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT, root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]
