"""not gpu: pins the oracle's network restatement (oracle/unet_ref.py) to the reference.

1. against golden vectors produced by importing the reference's own Generic_UNet
   (oracle/gen_golden.py -> tests/golden/net_*.npz);
2. when /root/reference is present (build container), directly against the live reference
   module on a fresh seed, so fixture and restatement cannot drift together."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_shim, unet_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _moments(t):
    t = t.double()
    return np.array([t.mean().item(), t.std(unbiased=False).item(), t.norm().item(), t.abs().max().item()])


@pytest.mark.parametrize("name", ["A", "A_in", "B"])
def test_oracle_matches_reference_golden_64(amd, name):
    g = np.load(os.path.join(GOLD, f"net_{name}.npz"))
    sd, meta = amd.synthetic.make_model(name, seed=int(g["weight_seed"]))
    x = np.random.RandomState(int(g["input_seed"])).standard_normal((1, 4, 64, 64, 64)).astype(np.float32)
    y, stages = unet_ref.unet_forward(sd, x, unet_ref.default_cfg(norm=meta["norm"], num_groups=16), return_stages=True)
    # same torch ops in the same order: the restatement is bit-identical on the generating machine,
    # allow a few ulp of the logit scale for other CPUs / oneDNN kernels
    tol = 1e-5 * float(g["logits_moments"][1])
    assert np.abs(y[:, :, ::8, ::8, ::8].numpy() - g["logits_sub"]).max() <= tol
    assert np.allclose(_moments(y), g["logits_moments"], rtol=1e-5)
    for k in ("ctx0", "ctx3", "loc4"):
        assert np.allclose(_moments(stages[k]), g[f"stage_{k}"], rtol=1e-5), k
    assert np.allclose(_moments(stages["bottleneck"]), g["stage_ctx5"], rtol=1e-5)


@pytest.mark.parametrize("norm", ["batch", "instance", "group"])
@pytest.mark.parametrize("nonlin_first", [False, True])
def test_oracle_matches_reference_golden_small(amd, norm, nonlin_first):
    g = np.load(os.path.join(GOLD, f"net_small_{norm}_{int(nonlin_first)}.npz"))
    sd, _ = amd.synthetic.make_model("A", seed=5, num_pool=2, max_feat=64, norm=norm)
    x = np.random.RandomState(6).standard_normal((2, 4, 16, 16, 32)).astype(np.float32)
    y = unet_ref.unet_forward(sd, x, unet_ref.default_cfg(norm=norm, num_groups=8, nonlin_first=nonlin_first)).numpy()
    assert y.shape == g["logits"].shape
    assert np.abs(y - g["logits"]).max() <= 1e-5 * max(1.0, float(np.abs(g["logits"]).max()))


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference absent (GPU box)")
@pytest.mark.parametrize("cfg", [
    dict(norm="batch", base=32, num_pool=3, max_feat=128, encoder_scale=1),
    dict(norm="group", base=32, num_pool=3, max_feat=128, encoder_scale=2),
    dict(norm="instance", base=32, num_pool=2, max_feat=320, encoder_scale=1),
])
def test_oracle_matches_live_reference_module(amd, cfg):
    sd = amd.synthetic.make_state_dict(seed=99, all_heads=True, **cfg)
    net = ref_shim.build_reference_net(cfg["norm"], 16, base=cfg["base"], num_pool=cfg["num_pool"],
                                       max_feat=cfg["max_feat"], encoder_scale=cfg["encoder_scale"])
    # every key / shape the synthetic generator emits is what the reference constructor builds
    ref_sd = net.state_dict()
    assert set(sd.keys()) == set(ref_sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
    ref_shim.load_numpy_state_dict(net, sd)
    x = torch.from_numpy(np.random.RandomState(3).standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    with torch.no_grad():
        want = net(x)
    got = unet_ref.unet_forward(sd, x, unet_ref.default_cfg(norm=cfg["norm"], num_groups=16))
    assert torch.equal(got, want)


def test_flop_count_matches_survey(amd):
    """model A = 965.6 GFLOP per 128^3 patch (all five seg heads), model B = 3342."""
    sd = amd.synthetic.make_state_dict(seed=1, norm="none", all_heads=True)
    assert abs(unet_ref.conv_flops(sd, (128, 128, 128)) / 1e9 - 965.6) < 0.1
    sd_b = amd.synthetic.make_state_dict(seed=1, norm="none", all_heads=True, max_feat=512, encoder_scale=2)
    assert abs(unet_ref.conv_flops(sd_b, (128, 128, 128)) / 1e9 - 3342.4) < 1.0
