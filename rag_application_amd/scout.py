"""`search_across_spaces` of the IndexerAPI on the dense top-k kernels.

Mirrors `Neo4jHandler.search_across_spaces` (IndexerAPI/src/core/storage/neo4j_handler.py:809-827)
and its four `_search_*` helpers (:829-1047): one cosine vector index per SPACE ("page", "entity",
"column", "relationship"); each index answers top_k, the tenant predicate
`user_id = $user_id AND org_id = $org_id` is applied AFTER the top-k (`WITH node, score WHERE ...`,
:834-836 -- a space may therefore return fewer than top_k rows, or none), the four lists are
concatenated in that order, stably sorted by score descending (:826) and cut to `limit` (:827).

Assumed upstream behaviour, unverifiable offline (Neo4j is not in the image): the score of a cosine
vector index is (1 + cos) / 2, and `db.index.vector.queryNodes` is read as EXACT top-k (the engine
has no approximate mode).  Both are switches of the oracle (oracle/oracle.py: scout_*).

Each space is an `engine.HxIndex` (dense rows only); the top-k runs on the GPU (scan + exact fp32
re-score, certified), the score transform, tenant filter and merge of <= 4*top_k rows are host work.
Arithmetic: cos = the engine's spec_dot of the normalised vectors; score = fp32((1 + cos) * 0.5)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import engine as eng

SPACES = ("page", "entity", "column", "relationship")     # the order of neo4j_handler.py:813-818


def neo4j_cosine_score(cos: np.ndarray) -> np.ndarray:
    """(1 + cos) / 2 in fp32: one add, one multiply by the exact constant 0.5."""
    return ((np.float32(1.0) + cos.astype(np.float32)) * np.float32(0.5)).astype(np.float32)


class _Space:
    def __init__(self, dim: int, device: int):
        self.ix = eng.HxIndex(dim, (), device=device)
        self.user: List[str] = []
        self.org: List[str] = []
        self.props: List[Dict[str, Any]] = []
        self._codes = None          # (user codes, org codes) as int32 arrays, rebuilt after add()

    def codes(self, intern: Dict[str, int]):
        if self._codes is None or len(self._codes[0]) != len(self.user):
            self._codes = (np.fromiter((intern.setdefault(u, len(intern)) for u in self.user), np.int32, len(self.user)),
                           np.fromiter((intern.setdefault(o, len(intern)) for o in self.org), np.int32, len(self.org)))
        return self._codes


class ScoutIndex:
    """Four vector spaces with per-row tenant tags and properties (kept on the host, as the
    reference keeps them on the node)."""

    def __init__(self, dim: int, device: int = 0):
        self.dim = int(dim)
        self.device = int(device)
        self.spaces = {s: _Space(self.dim, self.device) for s in SPACES}
        self._intern: Dict[str, int] = {}       # tenant string -> small integer (host-side filter)

    def add(self, space: str, embeddings: np.ndarray, user_ids: Sequence[str], org_ids: Sequence[str],
            props: Optional[Sequence[Dict[str, Any]]] = None) -> None:
        sp = self.spaces[space]
        X = np.ascontiguousarray(embeddings, np.float32)
        if X.ndim != 2 or X.shape[1] != self.dim:
            raise ValueError(f"embeddings must be [n, {self.dim}]")
        n = X.shape[0]
        if len(user_ids) != n or len(org_ids) != n or (props is not None and len(props) != n):
            raise ValueError("one user_id / org_id / props entry per row")
        sp.ix.add(X)
        sp.user.extend(str(u) for u in user_ids)
        sp.org.extend(str(o) for o in org_ids)
        sp.props.extend(dict(p) for p in props) if props is not None else sp.props.extend({} for _ in range(n))

    def count(self, space: str) -> int:
        return len(self.spaces[space].user)

    def search_across_spaces_batch(self, query_embeddings, top_k: int, user_id: str, org_id: str
                                   ) -> List[List[Dict[str, Any]]]:
        """B queries at once (additive: the reference is one query per call)."""
        Q = np.ascontiguousarray(query_embeddings, np.float32)
        if Q.ndim == 1:
            Q = Q[None, :]
        limit = max(1, int(top_k))                                  # :812
        qd = torch.from_numpy(Q).to(torch.device("cuda", self.device))
        B = Q.shape[0]
        ucode, ocode = self._intern.get(user_id, -1), self._intern.get(org_id, -1)
        S, I, M, W = [], [], [], []          # scores, rows, keep mask, space index: [B, <= 4*limit]
        for k, name in enumerate(SPACES):                           # asyncio.gather keeps this order (:819)
            sp = self.spaces[name]
            if not sp.user:
                continue
            users, orgs = sp.codes(self._intern)
            ucode, ocode = self._intern.get(user_id, -1), self._intern.get(org_id, -1)
            keys, cnt = sp.ix.search_dense(qd, min(limit, eng_max_limit()))
            s, i = eng.unpack(keys)
            s, i, c = neo4j_cosine_score(s.cpu().numpy()), i.cpu().numpy(), cnt.cpu().numpy()
            valid = np.arange(s.shape[1])[None, :] < c[:, None]
            ii = np.where(valid, i, 0)
            keep = valid & (users[ii] == ucode) & (orgs[ii] == ocode)          # post-filter (:834-836)
            S.append(s); I.append(ii); M.append(keep); W.append(np.full(s.shape, k, np.int8))
        if not S:
            return [[] for _ in range(B)]
        S, I, M, W = (np.concatenate(x, axis=1) for x in (S, I, M, W))
        # stable sort by score descending, as list.sort(reverse=True) (:826); dropped rows sink
        order = np.argsort(np.where(M, -S, np.float32(np.inf)), axis=1, kind="stable")
        out: List[List[Dict[str, Any]]] = []
        for b in range(B):
            n = min(limit, int(M[b].sum()))                                    # :827
            items = []
            for j in order[b, :n]:
                name, row = SPACES[int(W[b, j])], int(I[b, j])
                items.append({"space": name, "score": float(S[b, j]), "row": row, **self.spaces[name].props[row]})
            out.append(items)
        return out

    def search_across_spaces(self, query_embedding, top_k: int, user_id: str, org_id: str) -> List[Dict[str, Any]]:
        return self.search_across_spaces_batch(np.asarray(query_embedding, np.float32)[None, :], top_k, user_id,
                                               org_id)[0]

    def close(self) -> None:
        for sp in self.spaces.values():
            sp.ix.close()


def eng_max_limit() -> int:
    return 2048      # MAX_LIMIT of the engine (hx_common.hpp)
