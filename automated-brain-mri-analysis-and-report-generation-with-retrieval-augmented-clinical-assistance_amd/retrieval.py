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
    """Same constructor / ``retrieve`` contract as the reference class."""

    def __init__(self, documents: List[dict] | None = None, device="cuda"):
        self.documents = documents or []
        self.vocab: List[str] = []
        self.index = None
        if self.documents:
            self._build_index(device)

    @staticmethod
    def _tokenize(text: str) -> List[str]:
        return re.findall(r"[a-z]+", text.lower())  # rag_assistant.py:154-157

    def _build_index(self, device):
        all_tokens = [self._tokenize(d["text"]) for d in self.documents]
        self.vocab = sorted(set(t for toks in all_tokens for t in toks))
        self._w2i = {w: i for i, w in enumerate(self.vocab)}
        m = np.zeros((len(self.documents), len(self.vocab)))
        for r, toks in enumerate(all_tokens):
            for t in toks:
                m[r, self._w2i[t]] += 1
        self.index = VectorIndex(m, device=device, normalise=True)

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
        return [(self.documents[i], s) for i, s in self.index.topk(self._query_vector(query), top_k)]
