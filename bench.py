#!/usr/bin/env python
"""bench.py - BraTS volumes/s of the MI355X-native nnU-Net sliding-window predictor.

One "step" = one pass of the hot path over one synthetic BraTS-shaped volume that is already
preprocessed and resident in HBM: tile gather -> U-Net forwards -> sigmoid -> Gaussian-weighted
aggregation -> probabilities -> region labels pasted into the full 155x240x240 uint8 volume.
Default workload = BASELINE.json configs[1]: one 4x240x240x155 volume, 128^3 patches, step 0.5,
model A (base 32, BatchNorm), 1 fold, fp32, no TTA -> 8 forwards per volume.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3] [--no-cpu-baseline]

For N > 1 the driver launches it under torch.distributed.run; every rank processes its own
volume (cases are the sharding unit: SURVEY.md 8e partitioning A, no data-path collective),
timing is barrier + synchronize bracketed and the maximum over ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, same guide
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3), help="BASELINE.json config (1-based)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-tiles", type=int, default=0)
    ap.add_argument("--dtype", choices=("f32", "f16"), default=None, help="default: f32 for config 2, f16 for config 3")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import brats_amd
    from brats_amd import synthetic, predictor, preprocessing, ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run (any N)
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    # ---- workload
    patch = (128, 128, 128)
    dtype = args.dtype or ("f32" if args.config == 2 else "f16")
    if args.config == 2:
        models = [("A", 7)]
        do_mirroring = False
        workload = "BASELINE.json configs[1]: 1 volume 4x240x240x155, 128^3 patches, step 0.5, model A, 1 fold, no TTA"
    else:
        models = [("A", 7), ("B", 8)]
        do_mirroring = True
        workload = "BASELINE.json configs[2]: 1 volume, 8-way mirror TTA, models A+B, 1 fold each, label-round ensemble"
    nets = []
    for name, seed in models:
        sd, meta = synthetic.make_model(name, seed=seed)
        nets.append(brats_amd.UNet(sd, norm=meta["norm"], num_groups=meta["num_groups"], dtype=dtype))
        del sd
    raw = synthetic.make_volume(seed=1000 + rank)
    data, props = preprocessing.preprocess_case(raw, device)
    full = props["original_size_of_raw_data"]
    lo = [b[0] for b in props["crop_bbox"]]
    steps_tbl = [ops.compute_steps(patch[a], max(patch[a], data.shape[1 + a]), 0.5) for a in range(3)]
    n_tiles = int(np.prod([len(s) for s in steps_tbl]))
    n_mirrors = 8 if do_mirroring else 1
    flops_per_volume = sum(n.flops(patch) for n in nets) * n_tiles * n_mirrors

    def step():
        segs = []
        for net in nets:
            probs = predictor.predict_folds([net], data, patch, 0.5, do_mirroring, (0, 1, 2), True, "sigmoid",
                                            batch_tiles=args.batch_tiles)
            segs.append(ops.regions_to_labels(probs, (1, 2, 3), lo, full))
        return segs[0] if len(segs) == 1 else ops.label_ensemble(segs[0], segs[1])

    def sync_all():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        seg = step()
    for net in nets:
        net.profile(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seg = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {}
    for net in nets:
        for e in net.read_profile():
            p = prof.setdefault(e["name"], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            for k in ("launches", "ms", "flops", "bytes"):
                p[k] += e[k]
        net.profile(False)
    label_hist = torch.bincount(seg.flatten().to(torch.int64), minlength=4).tolist()

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events on the launch stream, this timed region)
    dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = dom["ms"] / dom["launches"]
    achieved_tflops = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get(dom_name, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_F16_MFMA_TFLOPS
    # Winograd kernels execute fewer multiplies than the algorithmic (direct-convolution) count that `achieved` uses:
    # F(2,3) along y issues 2/3 of the direct MFMAs, F(2x2,3x3) over (z,y) 4/9; frac can therefore exceed 1 and
    # frac_executed = matrix-pipe utilisation is reported beside it.
    executed_ratio = {"conv3_f32_wino_kernel": 2.0 / 3.0, "conv3_f32_wino2_kernel": 4.0 / 9.0}.get(dom_name, 1.0)
    if args.config != 2 or args.batch_tiles:
        traffic = None  # the committed PMC passes were taken on the default workload's launch sizes
    roofline = dict(bound="mfma", kernel=dom_name, achieved=round(achieved_tflops, 2), peak=peak,
                    unit="TFLOP/s", frac=round(achieved_tflops / peak, 4), traffic=traffic,
                    executed_flop_ratio=round(executed_ratio, 4),
                    frac_executed=round(achieved_tflops * executed_ratio / peak, 4),
                    launches=dom["launches"], avg_launch_ms=round(avg_ms, 4),
                    algorithmic_gflop_per_launch=round(dom["flops"] / dom["launches"] / 1e9, 3),
                    algorithmic_mb_per_launch=round(dom["bytes"] / dom["launches"] / 1e6, 2),
                    time_share=round(dom["ms"] / sum(p["ms"] for p in prof.values()), 4))
    kernels = {k: dict(launches=v["launches"], ms_total=round(v["ms"], 3),
                       tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                       gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None)
               for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- CPU baseline: the oracle on this host's cores, bounded sample of the same workload
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import unet_ref, tiler_ref
        sd, meta = synthetic.make_model(models[0][0], seed=models[0][1])
        cfg = unet_ref.default_cfg(norm=meta["norm"], num_groups=meta["num_groups"])
        # the sample is a REAL tile of this step's volume (first tile of the sliding window), so the timed CPU
        # forwards double as the parity check of the GPU path on the same input
        z0, y0, x0 = steps_tbl[0][0], steps_tbl[1][0], steps_tbl[2][0]
        tile = data[:, z0:z0 + patch[0], y0:y0 + patch[1], x0:x0 + patch[2]]
        pad = [patch[i] - tile.shape[1 + i] for i in range(3)]
        tile = torch.nn.functional.pad(tile, (0, pad[2], 0, pad[1], 0, pad[0]))[None].contiguous()
        x = tile.cpu()
        cores = torch.get_num_threads()
        unet_ref.unet_forward(sd, x[:, :, :64, :64, :64], cfg)  # page in
        n_fw = 0
        tc0 = time.perf_counter()
        ref_logits = None
        while n_fw < 2 or (time.perf_counter() - tc0 < 10.0 and n_fw < 6):
            ref_logits = unet_ref.unet_forward(sd, x, cfg)
            n_fw += 1
        per_fw = (time.perf_counter() - tc0) / n_fw
        cpu_flops_model0 = nets[0].flops(patch)
        est_volume_s = per_fw * flops_per_volume / cpu_flops_model0
        got_logits = nets[0](tile).cpu()
        pr, pg = torch.sigmoid(ref_logits)[0].numpy(), torch.sigmoid(got_logits)[0].numpy()
        dice_tile = tiler_ref.brats_region_dice(tiler_ref.regions_to_labels(pg), tiler_ref.regions_to_labels(pr))
        parity = dict(sample="first 128^3 tile of the timed volume, GPU forward vs the CPU oracle forward",
                      max_abs_prob_err=float(np.abs(pg - pr).max()),
                      max_abs_logit_err=float((got_logits - ref_logits).abs().max()), logit_std=float(ref_logits.std()),
                      dice_wt_tc_et_mean=round(dice_tile["mean"], 6))
        cpu = dict(value=round(1.0 / est_volume_s, 6), unit="volumes/s", cores=cores, kind="port",
                   sample=f"{n_fw} forwards of model {models[0][0]} on the first 1x4x128^3 tile of the timed volume with the torch-CPU fp32 oracle "
                          f"({per_fw:.2f} s each), scaled by flops to the {n_tiles * n_mirrors * len(nets)} forwards of one volume",
                   seconds_per_volume_est=round(est_volume_s, 2))

    value = world * args.steps / elapsed
    out = {
        "metric": "BraTS volumes/sec (4-modality 240x240x155)", "value": round(value, 4), "unit": "volumes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload, "patch": list(patch), "tiles_per_volume": n_tiles, "mirrors": n_mirrors,
                   "models": [m[0] for m in models], "crop": list(data.shape[1:]), "sharding": "cases (one volume per rank per step)",
                   "tflop_per_volume": round(flops_per_volume / 1e12, 3)},
        "sustained_tflops_per_gpu": round(flops_per_volume * args.steps / elapsed / 1e12, 2),
        "roofline": roofline, "cpu_baseline": cpu, "parity_vs_cpu_ref": parity, "kernels": kernels,
        "label_histogram": label_hist,
        "speedup_vs_nominal_5min": round(300.0 / (elapsed / args.steps), 1),
    }
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
