"""Generates tests/golden/*.npz from the REFERENCE's own network code (run in the build
container only: needs /root/reference).  Commit the outputs; the script documents how they
were made.  Weights and inputs come from frozen np.random.RandomState streams
(brats_amd.synthetic), so only small summaries are stored:

  net_<model>.npz   logits of reference Generic_UNet on a seeded 1x4x64^3 input: every 8th voxel
                    per axis, plus mean/std/l2/absmax of the full tensor and of every stage
                    output (forward hooks on the reference module).  Models: A (BatchNorm eval),
                    A_in (InstanceNorm), B (encoder_scale=2, max 512, GroupNorm-16).
  net_A_128.npz     moments only for a 1x4x128^3 input (the bench patch size).
  net_small_*.npz   full logits of tiny variants (nonlin_first etc.) used by fast CPU tests.
  driver_tables.npz label-round ensemble truth table produced by the reference's numpy
                    expression (run_brats2021_inference_singlethread.py:305).

    python -m oracle.gen_golden
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def moments(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.std(unbiased=False).item(), t.norm().item(), t.abs().max().item()])


def run_reference(name, sd, meta, preset, x, nonlin_first=False, num_pool=5, hooks=True):
    from oracle import ref_shim
    net = ref_shim.build_reference_net(meta["norm"], meta.get("num_groups", 16), base=preset.get("base", 32),
                                       num_pool=num_pool, max_feat=preset.get("max_feat"),
                                       encoder_scale=preset.get("encoder_scale", 1), nonlin_first=nonlin_first)
    ref_shim.load_numpy_state_dict(net, sd)
    stage = {}
    if hooks:
        for d, m in enumerate(net.conv_blocks_context):
            m.register_forward_hook(lambda mod, i, o, d=d: stage.__setitem__(f"ctx{d}", moments(o)))
        for u, m in enumerate(net.tu):
            m.register_forward_hook(lambda mod, i, o, u=u: stage.__setitem__(f"tu{u}", moments(o)))
        for u, m in enumerate(net.conv_blocks_localization):
            m.register_forward_hook(lambda mod, i, o, u=u: stage.__setitem__(f"loc{u}", moments(o)))
    with torch.no_grad():
        y = net(torch.from_numpy(x))
    return y, stage


def main():
    import brats_amd
    from brats_amd import synthetic
    os.makedirs(OUT, exist_ok=True)
    x64 = np.random.RandomState(1).standard_normal((1, 4, 64, 64, 64)).astype(np.float32)
    for name in ("A", "A_in", "B"):
        sd, meta = synthetic.make_model(name, seed=7)
        y, stage = run_reference(name, sd, meta, synthetic.MODEL_PRESETS[name], x64)
        np.savez_compressed(os.path.join(OUT, f"net_{name}.npz"), logits_sub=y[:, :, ::8, ::8, ::8].numpy(),
                            logits_moments=moments(y), input_seed=1, weight_seed=7,
                            **{f"stage_{k}": v for k, v in stage.items()})
        print(name, "logits moments", moments(y))
    sd, meta = synthetic.make_model("A", seed=7)
    x128 = np.random.RandomState(2).standard_normal((1, 4, 128, 128, 128)).astype(np.float32)
    y, stage = run_reference("A", sd, meta, synthetic.MODEL_PRESETS["A"], x128)
    np.savez_compressed(os.path.join(OUT, "net_A_128.npz"), logits_sub=y[:, :, ::16, ::16, ::16].numpy(),
                        logits_moments=moments(y), input_seed=2, weight_seed=7,
                        **{f"stage_{k}": v for k, v in stage.items()})
    # tiny variants with full logits
    xs = np.random.RandomState(6).standard_normal((2, 4, 16, 16, 32)).astype(np.float32)
    for norm in ("batch", "instance", "group"):
        for nonlin_first in (False, True):
            sd, _ = synthetic.make_model("A", seed=5, num_pool=2, max_feat=64, norm=norm)
            y, _ = run_reference("small", sd, dict(norm=norm, num_groups=8), dict(base=32, max_feat=64), xs,
                                 nonlin_first=nonlin_first, num_pool=2, hooks=False)
            np.savez_compressed(os.path.join(OUT, f"net_small_{norm}_{int(nonlin_first)}.npz"), logits=y.numpy())
    # the reference's label ensemble expression (driver :305) on every label pair
    a, b = np.meshgrid(np.arange(5, dtype=np.float64), np.arange(5, dtype=np.float64), indexing="ij")
    np.savez_compressed(os.path.join(OUT, "driver_tables.npz"),
                        label_round=np.round((a + b) / 2.0).astype(np.uint8))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
