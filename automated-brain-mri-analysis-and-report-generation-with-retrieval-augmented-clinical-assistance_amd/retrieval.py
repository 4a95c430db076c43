"""GPU cosine top-k behind the reference's fallback vector store (SURVEY.md 8f row 3,
BASELINE.json configs[4]).

``DummyVectorStore`` (RAG_Assistant/rag_assistant.py:131-211) is mirrored: tokenisation and the
term-frequency index are built on the host exactly as the reference does (:161-185), the L2-normalised
matrix lives in HBM as fp32 and ``retrieve`` is one ``mi355_cosine_topk`` call (GEMV + top-k).
An external ``[N, D]`` matrix (e.g. sentence-transformer vectors exported to .npy) can be used through
``VectorIndex``; the embedding model itself (hub fetch) is out of reach offline.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import re
from typing import List, Tuple

import numpy as np

from . import _lib


class VectorIndex:
    """Row-normalised fp32 matrix on the device + top-k by dot product."""

    def __init__(self, vectors: np.ndarray, device="cuda", normalise=True):
        import torch
        v = np.asarray(vectors, dtype=np.float64)
        if v.ndim != 2:
            raise ValueError("vectors must be [N, D]")
        if normalise:
            n = np.linalg.norm(v, axis=1, keepdims=True)
            n[n == 0] = 1
            v = v / n
        self.host = v
        self.n, self.d = v.shape
        self.dev = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(device)

    def topk(self, query: np.ndarray, k: int = 2) -> List[Tuple[int, float]]:
        import torch
        q = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32)).to(self.dev.device)
        if q.numel() != self.d:
            raise ValueError(f"query has {q.numel()} dims, index has {self.d}")
        idx = (C.c_int32 * 64)()
        sc = (C.c_float * 64)()
        stream = torch.cuda.current_stream(self.dev.device).cuda_stream
        n = _lib.load().mi355_cosine_topk(self.dev.data_ptr(), q.data_ptr(), self.n, self.d, int(k), idx, sc, stream)
        _lib.check(n, "mi355_cosine_topk")
        return [(int(idx[i]), float(sc[i])) for i in range(n)]


class DummyVectorStore:
    """Same constructor / ``retrieve`` contract as the reference class; ``vocab`` and the L2-normalised ``vectors``
    matrix are built on the host exactly as there, the matrix is uploaded once (at the first ``retrieve``)."""

    def __init__(self, documents: List[dict] | None = None, device="cuda"):
        self.documents = documents or []
        self.vocab: List[str] = []
        self.vectors = np.array([])
        self.index = None
        self._device = device
        if self.documents:
            self._build_index()

    @staticmethod
    def _tokenize(text: str) -> List[str]:
        return re.findall(r"[a-z]+", text.lower())  # rag_assistant.py:154-157

    def _build_index(self):
        all_tokens = [self._tokenize(d["text"]) for d in self.documents]
        self.vocab = sorted(set(t for toks in all_tokens for t in toks))
        self._w2i = {w: i for i, w in enumerate(self.vocab)}
        m = np.zeros((len(self.documents), len(self.vocab)))
        for r, toks in enumerate(all_tokens):
            for t in toks:
                m[r, self._w2i[t]] += 1
        n = np.linalg.norm(m, axis=1, keepdims=True)
        n[n == 0] = 1
        self.vectors = m / n

    def _query_vector(self, query: str) -> np.ndarray:
        vec = np.zeros(len(self.vocab))
        for t in self._tokenize(query):
            if t in self._w2i:
                vec[self._w2i[t]] += 1
        n = np.linalg.norm(vec)
        return vec / n if n > 0 else vec

    def retrieve(self, query: str, top_k: int = 2) -> List[Tuple[dict, float]]:
        if not self.documents:
            return []
        if self.index is None:
            self.index = VectorIndex(self.vectors, device=self._device, normalise=False)
        return [(self.documents[i], s) for i, s in self.index.topk(self._query_vector(query), top_k)]


# ---- the knowledge base behind BASELINE.json configs[4] ---------------------------------------------------------
#: question gating of answer_query (rag_assistant.py:61-64, 231-254): a query containing one of these is refused
#: before any retrieval happens
BLOCKED_KEYWORDS = ("treatment", "therapy", "surgery", "medication", "drug", "prognosis", "survival", "outcome",
                    "chemotherapy", "radiation")


def is_clinical_query(user_query: str) -> bool:
    q = user_query.lower()
    return any(k in q for k in BLOCKED_KEYWORDS)


def parse_md_document(raw: str, source: str = "") -> Tuple[str, dict]:
    """One knowledge-base article -> (text that is embedded, metadata): the header block up to the first ``---``
    holds ``TITLE:`` / ``KEYWORDS:`` / ``VERSION:`` lines, and the embedded text is
    ``"Title: ..." + "Keywords: ..." + body`` joined by blank lines (RAG_Assistant/vector_store_builder.py:71-141)."""
    meta = {"title": "", "keywords": "", "version": "", "source": source}
    body = raw
    if "---" in raw:
        head, _, rest = raw.partition("---")
        body = rest.strip()
        for line in head.strip().splitlines():
            for key in ("TITLE", "KEYWORDS", "VERSION"):
                if line.startswith(key + ":"):
                    meta[key.lower()] = line[len(key) + 1:].strip()
    parts = []
    if meta["title"]:
        parts.append(f"Title: {meta['title']}")
    if meta["keywords"]:
        parts.append(f"Keywords: {meta['keywords']}")
    parts.append(body)
    return "\n\n".join(parts), meta


def load_knowledge_base(directory) -> List[dict]:
    """Every ``*.md`` of a knowledge-base folder, sorted by file name, one document per file
    (vector_store_builder.py:183-206), as the ``{"term", "text"}`` dicts DummyVectorStore takes."""
    docs = []
    for path in sorted(glob.glob(os.path.join(str(directory), "*.md"))):
        with open(path, "r", encoding="utf-8") as f:
            text, meta = parse_md_document(f.read(), os.path.basename(path))
        docs.append({"term": meta["title"] or os.path.splitext(meta["source"])[0], "text": text, "source": meta["source"],
                     "id": os.path.splitext(meta["source"])[0]})
    if not docs:
        raise FileNotFoundError(f"no .md files found in {directory!r}")
    return docs


def definitions_block(retrieved: List[Tuple[dict, float]]) -> str:
    """CONTEXT 2 of the assistant's prompt for DummyVectorStore results (rag_assistant.py:400-407)."""
    lines = [f"- {doc.get('term', 'Definition')}: {doc['text']}" for doc, _ in retrieved]
    return "\n\n".join(lines) if lines else "No definitions retrieved."


def retrieve_for_query(store: "DummyVectorStore", user_query: str, top_k: int = 2):
    """Steps 1-2 of answer_query (rag_assistant.py:494-540): keyword gating, then top-k retrieval on the GPU.
    Returns None for a gated (clinical) query, else the list of (document, score)."""
    if is_clinical_query(user_query):
        return None
    return store.retrieve(user_query, top_k)
