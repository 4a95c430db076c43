"""Multi-GPU sharding of the path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  SURVEY.md 8e.

* Partitioning A (throughput; BASELINE.json configs[3], bench.py): CASES are dealt to ranks - longest
  first onto the least loaded rank when their tile counts are known (real crops run from 2 to 12
  tiles), round-robin otherwise -, every rank holds the weights, no data-path
  collective; an optional all_gather of the uint8 label maps (8.9 MB each) brings results to every rank.
* Partitioning B (latency of ONE case): the tile list is dealt round-robin
  (``mi355_sw_partial``), each rank holds a Gaussian-weighted partial aggregate, and there is one
  exchange step: an all_gather of the partial aggregates (3 x Zp x Yp x Xp fp32, about 40 MB per
  rank) followed by a sum in RANK ORDER on every rank, so all ranks obtain bit-identical
  probabilities (a ring all-reduce would add in a topology-dependent order).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence


def init_distributed(backend: str = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def tiles_of_shape(shape_zyx: Sequence[int], patch_size: Sequence[int] = (128, 128, 128), step_size: float = 0.5) -> int:
    """Sliding-window tiles of a (cropped) volume: per axis ceil((max(size, patch) - patch) / (patch * step)) + 1, the count
    behind nnU-Net v1's ``_compute_steps_for_sliding_window`` - the cost of a case in forwards."""
    n = 1
    for s, p in zip(shape_zyx, patch_size):
        n *= int(math.ceil((max(int(s), int(p)) - int(p)) / (int(p) * float(step_size)))) + 1
    return n


def shard_cases(n_cases: int, rank: int, world: int, weights: Optional[Sequence[float]] = None) -> List[int]:
    """The case indices of ``rank``.  Without ``weights``: round-robin (case i goes to rank i % world).  With ``weights``
    (one cost per case, e.g. its tile count): longest-processing-time-first - cases in descending cost (ties by index) each
    go to the rank with the smallest load so far (ties to the lowest rank); every rank computes the same assignment from
    the same list, no communication.  Returned in ascending index order."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if weights is None:
        return list(range(rank, n_cases, world))
    if len(weights) != n_cases:
        raise ValueError(f"{len(weights)} weights for {n_cases} cases")
    load = [0.0] * world
    mine = []
    for i in sorted(range(n_cases), key=lambda j: (-float(weights[j]), j)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += float(weights[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def shard_tiles(n_tiles: int, rank: int, world: int) -> List[int]:
    """The tile assignment ``mi355_sw_partial`` uses (tile index % world == rank)."""
    return list(range(rank, n_tiles, world))


def shard_fold_tiles(n_folds: int, n_tiles: int, rank: int, world: int) -> List[tuple]:
    """The (fold, tile) work items of ``rank`` as ``mi355_sw_partial_folds`` deals them: item ``f * n_tiles + t`` goes to
    rank ``item % world`` (SURVEY.md 8e partitioning B over the reference's fold list, driver :161)."""
    return [(i // n_tiles, i % n_tiles) for i in range(rank, n_folds * n_tiles, world)]


def sum_partials_in_rank_order(partial, group=None):
    """all_gather the per-rank partial aggregates, then add them 0,1,2,... on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return partial
    world = dist.get_world_size(group)
    if partial.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (ranks sharing one GPU synchronise over gloo, bench.py --backend gloo --share-gpu): stage through the host
        host = partial.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        total = parts[0].to(partial.device)
        for r in range(1, world):
            total += parts[r].to(partial.device)
        return total
    parts = [torch.empty_like(partial) for _ in range(world)]
    dist.all_gather(parts, partial.contiguous(), group=group)   # RCCL over xGMI on the "nccl" backend
    total = parts[0].clone()
    for r in range(1, world):
        total += parts[r]
    return total


def gather_label_maps(seg, group=None):
    """all_gather of equally-shaped uint8 label maps -> list indexed by rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [seg]
    out = [torch.empty_like(seg) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, seg.contiguous(), group=group)
    return out


def predict_cases_sharded(models, cases, rank: int, world: int, patch_size=(128, 128, 128), step_size=0.5,
                          do_mirroring=True, mirror_axes=(0, 1, 2), nonlin="sigmoid", region_order=(1, 2, 3), weights=None):
    """Partitioning A (BASELINE.json configs[3]): this rank's share of a batch of cases, no data-path collective.

    ``models`` is a list of ensemble members, each a list of fold networks (the reference runs model 1 then model 2
    with five folds each, driver :263-264); ``cases[i]`` is ``(data, props)`` as ``preprocess_case`` returns it, or a
    callable producing that (so a rank only materialises its own cases).  Per case: fold-mean probabilities per
    member (driver :128), region labels pasted at the crop box (:144-156), label-round ensemble of two members
    (:305).  Returns ``{case index: uint8 label volume on the device}`` for this rank's cases (``shard_cases``: round-robin,
    or longest-first by ``weights`` = tile counts when given).
    """
    from . import ops, predictor
    if len(models) not in (1, 2):
        raise ValueError("the reference ensembles one or two members (label-round ensemble is pairwise)")
    out = {}
    for i in shard_cases(len(cases), rank, world, weights):
        data, props = cases[i]() if callable(cases[i]) else cases[i]
        lo = [b[0] for b in props["crop_bbox"]]
        segs = []
        for folds in models:
            probs = predictor.predict_folds(folds, data, patch_size, step_size, do_mirroring, mirror_axes, True, nonlin)
            segs.append(ops.regions_to_labels(probs, region_order, lo, props["original_size_of_raw_data"]))
        out[i] = segs[0] if len(segs) == 1 else ops.label_ensemble(segs[0], segs[1])
    return out


def predict_case_tile_sharded(net, data, patch_size=(128, 128, 128), step_size=0.5, do_mirroring=True,
                              mirror_axes=(0, 1, 2), use_gaussian=True, nonlin="sigmoid", group=None):
    """Partitioning B end to end on the GPU ranks of ``group``: identical probabilities on every rank.  ``net`` is one
    network or the fold list of one ensemble member (the reference's five folds, driver :161): the (fold, tile) work list
    is dealt over the ranks, every rank sums its own items over ALL folds locally, and there is ONE exchange per member
    (all_gather of the partial aggregates, summed in rank order) - the fold mean is linear in the aggregates."""
    import torch.distributed as dist
    from . import predictor
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_folds = len(net) if isinstance(net, (list, tuple)) else 1
    agg, cnt = predictor.predict_tile_sharded(net, data, rank, world, patch_size, step_size, do_mirroring,
                                              mirror_axes, use_gaussian, nonlin)
    agg = sum_partials_in_rank_order(agg, group)
    return predictor.finish_sharded(agg, cnt, tuple(data.shape[1:]), patch_size, n_folds)
