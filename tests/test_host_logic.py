"""not gpu: host-side logic of the product and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import tiler_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(amd):
    hdr = open(os.path.join(ROOT, "include", "mi355_nnunet.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(amd._lib.EXPORTS), declared ^ set(amd._lib.EXPORTS)
    lib = ctypes.CDLL(str(amd._lib.lib_path()))
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} not exported"
    assert lib.mi355_version() >= 100


def test_compute_steps_in_library_matches_oracle(amd):
    """mi355_compute_steps is host code in the .so: callable without a GPU."""
    for img in list(range(128, 262)) + [300, 511]:
        want = tiler_ref.compute_steps_for_sliding_window((128,), (img,), 0.5)[0]
        assert amd.ops.compute_steps(128, img, 0.5) == want, img
    for patch, img, step in [(32, 70, 0.5), (64, 64, 0.5), (96, 200, 0.25), (128, 240, 1.0), (40, 57, 0.75)]:
        assert amd.ops.compute_steps(patch, img, step) == tiler_ref.compute_steps_for_sliding_window((patch,), (img,), step)[0]
    with pytest.raises(amd._lib.Mi355Error):
        amd.ops.compute_steps(128, 100, 0.5)  # image smaller than the patch: caller must pad first


def test_topology_from_state_dict_model_a_and_b(amd):
    sd, _ = amd.synthetic.make_model("A")
    t = amd.topology_from_state_dict(sd)
    assert (t.in_channels, t.num_classes, t.num_pool, t.head_cin) == (4, 3, 5, 32)
    assert [[(c.cin, c.cout, c.stride) for c in st] for st in t.enc] == [
        [(4, 32, 1), (32, 32, 1)], [(32, 64, 2), (64, 64, 1)], [(64, 128, 2), (128, 128, 1)],
        [(128, 256, 2), (256, 256, 1)], [(256, 320, 2), (320, 320, 1)], [(320, 320, 2), (320, 320, 1)]]
    assert t.tu == [(320, 320), (320, 256), (256, 128), (128, 64), (64, 32)]
    assert [(st[0].cin, st[-1].cout) for st in t.dec] == [(640, 320), (512, 256), (256, 128), (128, 64), (64, 32)]
    assert t.has_batchnorm_stats
    assert abs(t.conv_flops((128, 128, 128)) / 1e9 - 965.47) < 0.05
    sd_b, _ = amd.synthetic.make_model("B")
    tb = amd.topology_from_state_dict({"module." + k: v for k, v in sd_b.items()})  # DataParallel prefix
    # irregular decoder of encoder_scale=2: tu.1 is 256->512, loc.1 is 1024->512->256
    assert tb.tu[1] == (256, 512) and [(c.cin, c.cout) for c in tb.dec[1]] == [(1024, 512), (512, 256)]
    assert tb.head_cin == 32 and not tb.has_batchnorm_stats
    with pytest.raises(ValueError):
        amd.topology_from_state_dict({**sd, "axial_attention.0.to_q.weight": np.zeros(1)})


def test_synthetic_generators_are_frozen(amd):
    sd, _ = amd.synthetic.make_model("A", seed=7)
    w = sd["conv_blocks_context.0.blocks.0.conv.weight"]
    assert w.dtype == np.float32 and w.shape == (32, 4, 3, 3, 3)
    assert abs(float(w[0, 0, 0, 0, 0]) - 0.23003991) < 1e-6  # RandomState(7) stream
    v = amd.synthetic.make_volume(seed=3, shape=(24, 32, 28))
    assert v.shape == (4, 24, 32, 28) and v.dtype == np.float32 and (v == 0).any() and v.max() > 1000
    v2 = amd.synthetic.make_volume(seed=3, shape=(24, 32, 28))
    assert np.array_equal(v, v2)


def test_no_gpu_means_loud_failure_not_fallback(amd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sd, meta = amd.synthetic.make_model("A", num_pool=2, max_feat=64)
    with pytest.raises(amd._lib.Mi355Error, match="no HIP device|CPU fallback"):
        amd.UNet(sd, norm="batch")
    with pytest.raises(ValueError):
        amd.UNet(amd.synthetic.make_model("A_in", num_pool=2, max_feat=64)[0], norm="auto")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    for f in ("run_brats2021_inference_singlethread.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(p).read(), flags=re.M)


def test_oracle_kaist_postprocessing_truth_table():
    """oracle/driver_ref.py: ET threshold (kaist_original_inference.py:33) and label convention (:34), on hand-made cases."""
    import numpy as np
    from oracle import driver_ref, extras_ref
    seg = np.zeros((4, 5, 6), np.uint8)
    seg[0, 0, :3] = 3
    seg[1, 1, :4] = 1
    seg[2, 2, :2] = 2
    low = driver_ref.apply_brats_threshold(seg, threshold=4, replace_with=2)       # 3 voxels < 4 -> relabelled
    assert (low == 3).sum() == 0 and (low == 2).sum() == 5 and (low == 1).sum() == 4
    same = driver_ref.apply_brats_threshold(seg, threshold=3, replace_with=2)      # 3 voxels, not < 3 -> kept
    assert np.array_equal(same, seg)
    conv = driver_ref.convert_labels_back_to_brats(seg)
    assert np.array_equal(conv, extras_ref.convert_labels(seg, "brats2021"))       # the in-repo converter pins it
    assert set(np.unique(conv)) == {0, 1, 2, 4} and (conv == 4).sum() == 3 and (conv == 2).sum() == 4 and (conv == 1).sum() == 2


def test_knowledge_base_loader_and_gating_match_the_reference_fixture(amd, tmp_path):
    """BASELINE.json configs[4], host side: the seven knowledge-base articles parsed by the product give the documents,
    vocabulary and question gating the reference's own code gave (tests/golden/rag_kb.json, made by oracle/gen_golden.py
    from the reference's DummyVectorStore / parse_md_file / is_clinical_query)."""
    import hashlib
    import json
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "rag_kb.json"), encoding="utf-8"))
    for name, text in fx["kb_files"].items():
        (tmp_path / name).write_text(text, encoding="utf-8")
    docs = amd.retrieval.load_knowledge_base(tmp_path)
    assert [d["source"] for d in docs] == [d["source"] for d in fx["docs"]]
    assert [d["term"] for d in docs] == [d["term"] for d in fx["docs"]]
    assert [hashlib.sha256(d["text"].encode()).hexdigest() for d in docs] == [d["text_sha256"] for d in fx["docs"]]
    store = amd.retrieval.DummyVectorStore(docs)          # index built on the host; nothing touches the device yet
    assert len(store.vocab) == fx["vocab_size"]
    assert hashlib.sha256("\n".join(store.vocab).encode()).hexdigest() == fx["vocab_sha256"]
    for e in fx["expected"]:
        assert amd.retrieval.is_clinical_query(e["query"]) == e["clinical"], e["query"]
        assert np.allclose(store.vectors @ store._query_vector(e["query"]), e["all_scores"], atol=1e-15)
    with pytest.raises(FileNotFoundError):
        amd.retrieval.load_knowledge_base(tmp_path / "empty")


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without a launcher starts torch.distributed.run as a CHILD (never exec) before importing torch,
    with the exact command line the driver uses, and relays its exit code."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                       "--master-port", "29511", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    calls = []

    class Done:
        returncode = 7
    monkeypatch.setattr(bench.subprocess, "run", lambda c, env=None: calls.append((c, env)) or Done())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse_args(["--gpus", "2", "--config", "4"])
    with pytest.raises(SystemExit) as e:
        bench.maybe_self_launch(args, ["--gpus", "2", "--config", "4"])
    assert e.value.code == 7 and calls and calls[0][0][-4:] == ["--gpus", "2", "--config", "4"]
    assert calls[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")                                  # already under a launcher: nothing is started
    bench.maybe_self_launch(args, ["--gpus", "2"])
    bench.maybe_self_launch(bench.parse_args([]), [])                      # N = 1: nothing is started
    assert not calls


# ---------------------------------------------------------------- ISA gate of the build (round 5; _isa_gate.py)
_GATE_KERNEL = """\t.text
_ZN5mi35520conv3_f16_dma_kernelILb0ELb0EEEvNS_9ConvArgsHE:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_writelane_b32 v254, s0, 0
\tv_writelane_b32 v254, s1, 1
%s
\ts_endpgm
\t.amdgpu_metadata
amdhsa.kernels:
  - .agpr_count:     128
    .name:           _ZN5mi35520conv3_f16_dma_kernelILb0ELb0EEEvNS_9ConvArgsHE
    .private_segment_fixed_size: %d
    .sgpr_count:     100
    .sgpr_spill_count: 2
    .vgpr_count:     300
    .vgpr_spill_count: %d
"""


def test_isa_gate_flags_the_hazards_it_is_there_for(amd):
    """The three ways a compiler bump or an edit can silently break the hand-counted kernels (the cause of round 4's
    wino3 two-body fault): a spilled SGPR reloaded straight in front of an inline-asm load that uses it as scalar base, a compiler
    copy of a register an inline-asm load is still in flight to, scratch in a hand-counted kernel."""
    gate = amd._isa_gate
    hazard = ("\tv_readlane_b32 s0, v254, 0\n\tv_readlane_b32 s1, v254, 1\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v1, s[0:1] offset:0\n\t;;#ASMEND\n"
              "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND")
    f = gate.check_asm_text(_GATE_KERNEL % (hazard, 0, 0))
    assert len(f) == 2 and all(x.startswith("H1") for x in f), f
    padded = hazard.replace("\tglobal_load_dwordx4", "\ts_nop 4\n\tglobal_load_dwordx4")
    assert gate.check_asm_text(_GATE_KERNEL % (padded, 0, 0)) == []
    # the same reload in front of a COMPILER-emitted load is the hazard recogniser's business, not the gate's
    plain = hazard.replace("\t;;#ASMSTART\n\tglobal_load", "\tglobal_load").replace("offset:0\n\t;;#ASMEND", "offset:0")
    assert gate.check_asm_text(_GATE_KERNEL % (plain, 0, 0)) == []
    copy = ("\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v1, s[0:1] offset:0\n\t;;#ASMEND\n\tv_mov_b32_e32 v9, v5\n"
            "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND")
    f = gate.check_asm_text(_GATE_KERNEL % (copy, 0, 0))
    assert len(f) == 1 and f[0].startswith("H2"), f
    waited = copy.replace("\tv_mov_b32_e32 v9, v5\n", "").replace("vmcnt(0)\n\t;;#ASMEND", "vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v9, v5")
    assert gate.check_asm_text(_GATE_KERNEL % (waited, 0, 0)) == []
    f = gate.check_asm_text(_GATE_KERNEL % ("\ts_nop 0", 12, 2))
    assert len(f) == 1 and f[0].startswith("R "), f


def test_built_library_passes_the_isa_gate(amd):
    """Every listing the build left beside its objects passes the gate (the build itself refuses a listing that does not)."""
    import glob
    lst = sorted(glob.glob(os.path.join(os.path.dirname(str(amd._lib.lib_path())), "obj", "*.gfx950.s")))
    if not lst:
        pytest.skip("library built elsewhere: no listings beside it")
    gate = amd._isa_gate
    names = set()
    for path in lst:
        assert gate.check_asm_file(path) == [], path
        names |= {gate.demangle(n) for n in gate.resources(open(path).read())}
    # the gate knows the hand-counted kernels by name: a rename must not silently drop one out of the scratch check
    for prefix in ("conv3_f32_wino3_kernel", "conv3_f32_wino2_kernel", "conv3_f32_s2dma_kernel", "conv3_f16_dma_kernel", "conv3_f16_s2dma_kernel"):
        assert any(prefix in n for n in names), prefix
