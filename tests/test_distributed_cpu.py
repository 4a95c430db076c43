"""not gpu: the N > 1 orchestration with world_size 2 on the gloo backend.

The compute stays on the GPU in the product; here each rank produces its tile-sharded partial
aggregate with the ORACLE (same tile assignment as mi355_sw_partial: index % world == rank) and
the product's exchange step (parallel.sum_partials_in_rank_order / gather_label_maps / shard_*)
runs for real across two processes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partial_oracle(sd, vol, patch, rank, world):
    from oracle import tiler_ref, unet_ref
    net_fn = tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch"))
    padded, lo = tiler_ref.pad_to_patch(vol, patch)
    steps = tiler_ref.compute_steps_for_sliding_window(patch, padded.shape[1:], 0.5)
    g = tiler_ref.get_gaussian(patch)
    agg = np.zeros((3,) + padded.shape[1:], np.float32)
    cnt = np.zeros(padded.shape[1:], np.float32)
    idx = 0
    for x in steps[0]:
        for y in steps[1]:
            for z in steps[2]:
                sl = (slice(x, x + patch[0]), slice(y, y + patch[1]), slice(z, z + patch[2]))
                cnt[sl] += g
                if idx % world == rank:
                    t = torch.from_numpy(np.ascontiguousarray(padded[(slice(None),) + sl][None]))
                    pred = tiler_ref.mirror_and_predict(net_fn, t, (0, 1, 2), False, "sigmoid", torch.from_numpy(g))[0].numpy()
                    agg[(slice(None),) + sl] += pred
                idx += 1
    return agg, cnt, idx


def _partial_oracle_folds(sds, vol, patch, rank, world, par):
    """This rank's share of the (fold, tile) work list exactly as mi355_sw_partial_folds deals it (parallel.shard_fold_tiles):
    agg = sum over the rank's items of gaussian x sigmoid(net_fold(tile)), cnt = the one-fold normaliser."""
    from oracle import tiler_ref, unet_ref
    padded, lo = tiler_ref.pad_to_patch(vol, patch)
    steps = tiler_ref.compute_steps_for_sliding_window(patch, padded.shape[1:], 0.5)
    g = tiler_ref.get_gaussian(patch)
    tiles = [(z, y, x) for z in steps[0] for y in steps[1] for x in steps[2]]
    agg = np.zeros((3,) + padded.shape[1:], np.float32)
    cnt = np.zeros(padded.shape[1:], np.float32)
    for z, y, x in tiles:
        cnt[z:z + patch[0], y:y + patch[1], x:x + patch[2]] += g
    fns = [tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")) for sd in sds]
    for f, t in par.shard_fold_tiles(len(sds), len(tiles), rank, world):
        z, y, x = tiles[t]
        sl = (slice(z, z + patch[0]), slice(y, y + patch[1]), slice(x, x + patch[2]))
        tin = torch.from_numpy(np.ascontiguousarray(padded[(slice(None),) + sl][None]))
        agg[(slice(None),) + sl] += tiler_ref.mirror_and_predict(fns[f], tin, (0, 1, 2), False, "sigmoid", torch.from_numpy(g))[0].numpy()
    return agg, cnt, lo


def _worker_folds(rank, world, port, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import brats_amd
    from oracle import tiler_ref, unet_ref
    brats_amd.parallel.init_distributed("gloo")
    sds = [brats_amd.synthetic.make_model("A", seed=21 + k, num_pool=2, max_feat=64)[0] for k in range(3)]
    vol = np.random.RandomState(8).standard_normal((4, 24, 40, 20)).astype(np.float32)   # 2 x 4 x 2 = 16 tiles x 3 folds over 2 ranks
    patch = (16, 16, 16)
    agg, cnt, lo = _partial_oracle_folds(sds, vol, patch, rank, world, brats_amd.parallel)
    total = brats_amd.parallel.sum_partials_in_rank_order(torch.from_numpy(agg))          # ONE exchange for all folds
    probs = (total.numpy() / cnt[None] / np.float32(len(sds)))[:, lo[0]:lo[0] + 24, lo[1]:lo[1] + 40, lo[2]:lo[2] + 20]
    want = np.mean([tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")), vol, patch, 3, 0.5,
                                               False, (0, 1, 2), True, "sigmoid") for sd in sds], axis=0)   # driver :128
    assert np.abs(probs - want).max() < 1e-5
    gathered = [torch.empty_like(total) for _ in range(world)]
    dist.all_gather(gathered, total)
    assert all(torch.equal(gathered[0], t) for t in gathered)   # rank-ordered sum: bit-identical on every rank
    np.save(os.path.join(tmp, f"okf_{rank}.npy"), np.array([1]))
    dist.destroy_process_group()


def test_two_rank_gloo_fold_list_sharding(tmp_path, amd):
    """The reference's fold list through partitioning B (driver :161): (fold, tile) items dealt over two ranks,
    one rank-ordered exchange of the summed partial aggregates, fold mean = total / cnt / n_folds."""
    par = amd.parallel
    items = [par.shard_fold_tiles(5, 8, r, 3) for r in range(3)]
    assert sorted(sum(items, [])) == [(f, t) for f in range(5) for t in range(8)]      # a partition of the work list
    assert max(len(i) for i in items) - min(len(i) for i in items) <= 1               # balanced although 8 % 3 != 0
    port = _free_port()
    mp.spawn(_worker_folds, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"okf_{r}.npy").exists() for r in range(2))


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import brats_amd
    from oracle import tiler_ref, unet_ref
    r, w, _ = brats_amd.parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    sd, _ = brats_amd.synthetic.make_model("A", seed=21, num_pool=2, max_feat=64)
    vol = np.random.RandomState(8).standard_normal((4, 24, 40, 20)).astype(np.float32)
    patch = (16, 16, 16)
    agg, cnt, n_tiles = _partial_oracle(sd, vol, patch, rank, world)
    assert brats_amd.parallel.shard_tiles(n_tiles, rank, world) == list(range(rank, n_tiles, world))
    total = brats_amd.parallel.sum_partials_in_rank_order(torch.from_numpy(agg))
    probs = (total.numpy() / cnt[None])
    full = tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")), vol, patch, 3, 0.5,
                                      False, (0, 1, 2), True, "sigmoid")
    lo = [(max(p, s) - s) // 2 for p, s in zip(patch, vol.shape[1:])]
    probs = probs[:, lo[0]:lo[0] + 24, lo[1]:lo[1] + 40, lo[2]:lo[2] + 20]
    assert np.abs(probs - full).max() < 1e-5
    # bit-identical on every rank (rank-ordered sum), and label maps gather by rank
    gathered = [torch.empty_like(total) for _ in range(world)]
    dist.all_gather(gathered, total)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    seg = torch.full((3, 4), rank, dtype=torch.uint8)
    maps = brats_amd.parallel.gather_label_maps(seg)
    assert [int(m[0, 0]) for m in maps] == list(range(world))
    assert brats_amd.parallel.shard_cases(5, rank, world) == list(range(rank, 5, world))
    np.save(os.path.join(tmp, f"ok_{rank}.npy"), np.array([1]))
    dist.destroy_process_group()


def test_two_rank_gloo_tile_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok_{r}.npy").exists() for r in range(2))


def test_longest_first_case_assignment(amd):
    """real crops run from 2 to 12 tiles; round-robin by index can leave one rank with
    all the large cases.  shard_cases(..., weights) = longest-processing-time-first: a partition, deterministic, never worse
    than round-robin on these lists and within 4/3 of the lower bound max(mean load, largest case)."""
    par = amd.parallel
    rs = np.random.RandomState(5)
    for world in (2, 3, 8):
        for trial in range(20):
            n = int(rs.randint(world, 40))
            w = [float(v) for v in rs.choice([2, 4, 6, 8, 12, 18], size=n)]
            parts = [par.shard_cases(n, r, world, w) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))                       # disjoint and complete
            assert parts == [par.shard_cases(n, r, world, list(w)) for r in range(world)]  # same answer on every rank
            span = max(sum(w[i] for i in p) for p in parts)
            rr = max(sum(w[i] for i in par.shard_cases(n, r, world)) for r in range(world))
            assert span <= rr + 1e-9
            assert span <= 4.0 / 3.0 * max(sum(w) / world, max(w)) + 1e-9
    # the adversarial list for round-robin at world 2: large cases on even indices
    w = [12, 2, 12, 2, 12, 2, 12, 2]
    assert max(sum(w[i] for i in par.shard_cases(8, r, 2, w)) for r in range(2)) == 28
    assert max(sum(w[i] for i in par.shard_cases(8, r, 2)) for r in range(2)) == 48
    # equal costs (the synthetic batch: 8 tiles each): as balanced as round-robin
    assert sorted(len(par.shard_cases(32, r, 8, [8.0] * 32)) for r in range(8)) == [4] * 8
    with pytest.raises(ValueError):
        par.shard_cases(3, 0, 2, [1.0, 2.0])
    # tile count of a crop = product of the step-table lengths
    assert par.tiles_of_shape((140, 171, 137)) == 8 and par.tiles_of_shape((155, 240, 240)) == 18
    assert par.tiles_of_shape((155, 208, 177)) == 12 and par.tiles_of_shape((98, 134, 117)) == 2
    raw = np.zeros((4, 20, 30, 40), np.float32)
    raw[1, 3:9, 5:25, 7:8] = 1.0
    raw[0, 4, 6, 30] = 2.0
    assert amd.preprocessing.nonzero_crop_shape(raw) == (6, 20, 24)


def test_shard_helpers(amd):
    assert amd.parallel.shard_cases(32, 3, 8) == [3, 11, 19, 27]
    assert sorted(sum((amd.parallel.shard_cases(10, r, 4) for r in range(4)), [])) == list(range(10))
    with pytest.raises(ValueError):
        amd.parallel.shard_cases(4, 4, 4)
