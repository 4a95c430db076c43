"""-m gpu: the two BASELINE.json configurations round 1 never exercised end to end.

* configs[3] (batch of volumes, cases sharded over ranks): the per-rank workload of ``bench.py --config 4`` run under a
  REAL process group on the "nccl" backend (= RCCL) at world size 1 - same code path as N ranks, one rank - plus the
  single-case tile-sharded latency path (``parallel.predict_case_tile_sharded``) through a real RCCL all_gather.
* configs[4] (segmentation + knowledge-base retrieval): one synthetic case is segmented, its tumour volumes computed
  on the device, and a question is answered up to the retrieved definitions through ``retrieval.DummyVectorStore``
  built from the seven knowledge-base articles - against tests/golden/rag_kb.json, which oracle/gen_golden.py produced
  by running the reference's own classes (RAG_Assistant/rag_assistant.py:131-211, :231-254).
"""
import json
import os
import socket

import numpy as np
import pytest
import torch

from oracle import tiler_ref, unet_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nccl_world1(gpu):
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=gpu)
    yield dist
    dist.destroy_process_group()


def test_config4_rank_workload_under_nccl(amd, gpu, nccl_world1):
    """Two full-size synthetic cases with config-3 settings (8-way TTA, models A + B, fp16, label-round ensemble) through
    parallel.predict_cases_sharded at world 1 under RCCL; every case equals the unsharded per-case path bit for bit, a
    2-rank split covers the batch exactly once, and the label maps travel through a real RCCL all_gather."""
    dist = nccl_world1
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    nets = []
    for name, seed in (("A", 7), ("B", 8)):
        sd, meta = amd.synthetic.make_model(name, seed=seed)
        nets.append(amd.UNet(sd, norm=meta["norm"], num_groups=meta["num_groups"], dtype="f16"))
    cases = [amd.preprocessing.preprocess_case(amd.synthetic.make_volume(seed=1000 + i)) for i in range(2)]
    patch = (128, 128, 128)
    out = amd.parallel.predict_cases_sharded([[n] for n in nets], cases, dist.get_rank(), dist.get_world_size(), patch)
    assert sorted(out) == [0, 1]
    for i, (data, props) in enumerate(cases):
        lo = [b[0] for b in props["crop_bbox"]]
        segs = [amd.ops.regions_to_labels(amd.predictor.predict_folds([n], data, patch), (1, 2, 3), lo,
                                          props["original_size_of_raw_data"]) for n in nets]
        want = amd.ops.label_ensemble(segs[0], segs[1])
        # the BN-folded model is bit-reproducible; the GroupNorm model accumulates its statistics with fp64 atomics
        # (arrival order), which can move single voxels next to the 0.5 threshold
        d = tiler_ref.brats_region_dice(out[i].cpu().numpy(), want.cpu().numpy())
        assert out[i].shape == (155, 240, 240) and d["mean"] >= 0.9999, d
        assert set(np.unique(out[i].cpu().numpy())) <= {0, 1, 2, 3}
    # the same call as ranks 0 and 1 of a 2-rank job: disjoint, complete
    parts = [amd.parallel.predict_cases_sharded([[n] for n in nets[:1]], cases, r, 2, patch, do_mirroring=False) for r in range(2)]
    assert sorted(parts[0]) == [0] and sorted(parts[1]) == [1]
    gathered = amd.parallel.gather_label_maps(out[0])           # RCCL all_gather (world 1)
    assert len(gathered) == 1 and torch.equal(gathered[0], out[0])
    for n in nets:
        n.close()


def test_tile_sharded_latency_path_through_rccl(amd, gpu, nccl_world1):
    """parallel.predict_case_tile_sharded (mi355_sw_partial -> RCCL all_gather of the partial aggregates -> rank-ordered
    sum -> mi355_sw_finish) at world 1 equals mi355_sw_predict."""
    sd, _ = amd.synthetic.make_model("A", seed=21, num_pool=2, max_feat=128)
    net = amd.UNet(sd, norm="batch")
    vol = np.random.RandomState(8).standard_normal((4, 40, 48, 36)).astype(np.float32)
    patch = (32, 32, 32)
    data = torch.from_numpy(vol).to(gpu)
    got = amd.parallel.predict_case_tile_sharded(net, data, patch)
    want = amd.predictor.predict_folds([net], data, patch, lanes=1)   # (one lane = mi355_sw_predict: the same in-order sum)
    assert torch.equal(got, want)
    # the default (two lanes of one GPU: the same dealing as two ranks) differs by the fp32 rounding of the summation order only
    assert float((amd.predictor.predict_folds([net], data, patch) - want).abs().max()) <= 2e-6
    ref = tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")), vol, patch, 3)
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-3
    net.close()


def test_config5_segmentation_then_knowledge_base_retrieval(amd, gpu, tmp_path):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rag_kb.json"), encoding="utf-8"))
    for name, text in fx["kb_files"].items():
        (tmp_path / name).write_text(text, encoding="utf-8")
    # --- segmentation of one synthetic case (sliding window + TTA on the HIP path), checked against the oracle
    sd, _ = amd.synthetic.make_model("A", seed=21, num_pool=2, max_feat=128)
    net = amd.UNet(sd, norm="batch")
    raw = amd.synthetic.make_volume(seed=5, shape=(40, 56, 44))
    data, props = amd.preprocessing.preprocess_case(raw)
    patch = (32, 32, 32)
    probs = amd.predictor.predict_folds([net], data, patch)
    lo = [b[0] for b in props["crop_bbox"]]
    seg = amd.ops.regions_to_labels(probs, (1, 2, 3), lo, props["original_size_of_raw_data"])
    want_data, want_props = tiler_ref.preprocess_case(raw)
    ref = tiler_ref.predict_3d_tiled(tiler_ref.make_net_fn(sd, unet_ref.default_cfg("batch")), want_data, patch, 3)
    want_seg = np.zeros(raw.shape[1:], np.uint8)
    sl = tuple(slice(b[0], b[1]) for b in want_props["crop_bbox"])
    want_seg[sl] = tiler_ref.regions_to_labels(ref)
    assert tiler_ref.brats_region_dice(seg.cpu().numpy(), want_seg)["mean"] >= 0.999
    feats = amd.evaluate.tumor_region_features(amd.evaluate.convert_labels(seg, "brats2021"), 0.001)
    assert feats["wt"]["volume_cm3"] == pytest.approx(float((want_seg > 0).sum()) * 0.001, rel=2e-3)
    # --- retrieval over the knowledge base: the GPU cosine top-k behind the reference's DummyVectorStore contract
    store = amd.retrieval.DummyVectorStore(amd.retrieval.load_knowledge_base(tmp_path))
    n_gated = 0
    for e in fx["expected"]:
        got = amd.retrieval.retrieve_for_query(store, e["query"], top_k=2)
        if e["clinical"]:
            assert got is None                     # refused before retrieval (rag_assistant.py:517-518)
            n_gated += 1
            got = store.retrieve(e["query"], 2)    # the store itself still answers like the reference's
        assert [d["source"] for d, _ in got] == e["top"], e["query"]
        assert np.allclose([s for _, s in got], e["scores"], atol=1e-6), e["query"]   # fp32 dot products vs the reference's fp64
    assert n_gated == 2
    block = amd.retrieval.definitions_block(store.retrieve("What does midline shift mean in my report?", 2))
    assert block.startswith("- Midline Shift: Title: Midline Shift") and "\n\n- " in block
    # a larger index through the same kernel keeps the same answers: the 7 articles hidden among 50 000 random rows
    rs = np.random.RandomState(3)
    noise = rs.standard_normal((50000, store.vectors.shape[1])) * 1e-3
    big = amd.retrieval.VectorIndex(np.concatenate([noise, store.vectors]), normalise=False)
    top = big.topk(store._query_vector("Why is there edema around the tumor?"), 2)
    assert [i - 50000 for i, _ in top] == [[d["source"] for d in fx["docs"]].index(s) for s in fx["expected"][1]["top"]]
    net.close()


def test_bench_config4_under_the_launcher(gpu):
    """The exact command shape the driver's SCALE run uses (python -m torch.distributed.run ... bench.py --gpus N --config 4),
    at N = 1 with two cases and one step: a real `nccl` process group, the tile-count exchange, longest-first sharding and
    the one-line JSON contract.  Runs as a child process (the parent keeps its own HIP context; two processes on the card)."""
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "4", "--cases", "2", "--steps", "1",
           "--warmup", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["scaling"] == "strong" and out["dtype"] == "f16"
    assert out["config"]["cases"] == 2 and out["config"]["cases_per_rank"] == 2 and out["config"]["tiles_in_batch"] == 16
    assert out["value"] > 0 and out["roofline"]["kernel"].startswith("conv3_f16")
    assert sum(out["label_histogram"]) == 155 * 240 * 240


def test_bench_config3_tile_sharded_under_the_launcher(gpu):
    """`bench.py --config 3 --shard tiles` (SURVEY.md 8e partitioning B: ONE case, the (fold, tile) work list of each ensemble
    member dealt over the ranks, one RCCL all_gather of the partial aggregates per member, strong scaling) exactly as the
    driver would start it for N ranks - at N = 1, two folds per member, one step: a real `nccl` process group, the exchange on
    the data path, the one-line JSON contract with "scaling": "strong"."""
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "3", "--shard", "tiles", "--folds", "2",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["scaling"] == "strong" and out["dtype"] == "f16"
    assert out["config"]["folds_per_member"] == 2 and out["config"]["sharding"].startswith("tiles")
    assert out["value"] > 0 and sum(out["label_histogram"]) == 155 * 240 * 240
