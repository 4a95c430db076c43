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
  rag_kb.json       config 5: the seven knowledge-base articles and, per query, what the reference's own
                    DummyVectorStore / is_clinical_query return (see gen_rag_fixture).

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
    gen_rag_fixture()
    print("wrote", sorted(os.listdir(OUT)))


RAG_QUERIES = [
    "What does midline shift mean in my report?", "Why is there edema around the tumor?", "what is an enhancing tumor",
    "Which MRI sequences show the peritumoral swelling best", "How are tumor volumes measured in cubic centimeters",
    "explain non-enhancing tumor core and necrosis", "what is a glioma", "T1ce contrast gadolinium enhancement",
    "zzz qqq", "", "What treatment do I need for this glioma?", "is the prognosis bad given the midline shift",
]


def _import_by_path(name, path, stand_ins=None):
    import importlib.util
    import types
    saved = {k: sys.modules.get(k) for k in (stand_ins or {})}
    try:
        for mod, attrs in (stand_ins or {}).items():
            m = types.ModuleType(mod)
            m.__dict__.update(attrs)
            sys.modules[mod] = m
        spec = importlib.util.spec_from_file_location(name, path)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return module


def gen_rag_fixture():
    """tests/golden/rag_kb.json - BASELINE.json configs[4]: the reference's OWN DummyVectorStore
    (RAG_Assistant/rag_assistant.py:131-211, imported unmodified; numpy only) built over the seven knowledge_base
    articles as the reference's own parse_md_file (RAG_Assistant/vector_store_builder.py:71-141, imported with inert
    stand-ins for the absent chromadb package, which parse_md_file never touches) turns them into documents.  Stored:
    the article files (input DATA of the knowledge base), and per query the gating decision, the top-2 document ids and
    their float64 cosine scores."""
    import glob
    import hashlib
    import json
    ref = os.environ.get("REFERENCE_ROOT", "/root/reference")
    rag = _import_by_path("_reference_rag_assistant", os.path.join(ref, "RAG_Assistant", "rag_assistant.py"))
    vsb = _import_by_path("_reference_vector_store_builder", os.path.join(ref, "RAG_Assistant", "vector_store_builder.py"),
                          {"chromadb": {"Collection": object}, "chromadb.utils": {"embedding_functions": object()}})
    files = sorted(glob.glob(os.path.join(vsb.KNOWLEDGE_BASE_DIR, "*.md")))
    docs, kb = [], {}
    for path in files:
        text, meta = vsb.parse_md_file(path)
        kb[os.path.basename(path)] = open(path, "r", encoding="utf-8").read()
        docs.append({"term": meta["title"], "text": text, "source": meta["source"]})
    store = rag.DummyVectorStore(documents=docs)
    expected = []
    for q in RAG_QUERIES:
        got = store.retrieve(q, top_k=2)
        expected.append({"query": q, "clinical": bool(rag.is_clinical_query(q)),
                         "top": [d["source"] for d, _ in got], "scores": [s for _, s in got],
                         "all_scores": [float(v) for v in (store.vectors @ store._query_vector(q))]})
    out = {"generator": "oracle/gen_golden.py:gen_rag_fixture (reference classes imported from /root/reference)",
           "kb_files": kb,
           "docs": [{"term": d["term"], "source": d["source"], "text_sha256": hashlib.sha256(d["text"].encode()).hexdigest()} for d in docs],
           "vocab_size": len(store.vocab), "vocab_sha256": hashlib.sha256("\n".join(store.vocab).encode()).hexdigest(),
           "expected": expected}
    with open(os.path.join(OUT, "rag_kb.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("rag fixture:", len(docs), "docs,", len(store.vocab), "terms;", [(e["query"][:20], e["top"]) for e in expected[:3]])


if __name__ == "__main__":
    main()
