"""Drop-in for the reference's `QdrantHandler`
(app/core/vector_store/qdrant/qdrant_handler.py:14-481): same class name, same async
methods, same argument meaning and the same error conventions -- search and count
never raise (they return [] / 0, :384-386, :479-481), mutations re-raise (:196-198,
:265-267, :437-439), `create_collection` raises ValueError on an empty user id
(:39-40).  Where the reference ships a Prefetch tree to a Qdrant server over HTTP
(:363-372), this class calls the HIP engine through the C ABI (include/hx.h).

Additive: `hybrid_search_batch` (the reference is strictly one query per call) and a
configurable dense size (the reference hard-codes 768, :138-139; that stays the
default)."""
from __future__ import annotations

import asyncio
import itertools
import json
import logging
import os
import threading
import uuid
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

from . import engine as _engine
from . import filters as _filters
from ._lib import HX_MODE_H1, HX_MODE_TREE


@dataclass
class ScoredPoint:
    """What callers duck-type on (qdrant_handler.py:377;
    app/services/agents/search_orchestration_workflow.py:70-73, 168-175)."""
    id: str
    version: int
    score: float
    payload: Optional[Dict[str, Any]] = None
    vector: Optional[Any] = None
    shard_key: Optional[Any] = None
    order_value: Optional[Any] = None

    def dict(self):
        return asdict(self)

    model_dump = dict


@dataclass
class SparseVector:
    """Shape-compatible with qdrant_client.http.models.SparseVector."""
    indices: List[int] = field(default_factory=list)
    values: List[float] = field(default_factory=list)


def _sparse_parts(sv):
    """Accept {"indices": [...], "values": [...]} or an object with .indices/.values
    (qdrant_handler.py:348-351)."""
    if isinstance(sv, dict):
        return sv["indices"], sv["values"]
    return sv.indices, sv.values


class _Collection:
    """One user collection: the engine index (anything with HxIndex's add / hybrid_query_host / count / save /
    close: sharded.ShardedHandler plugs in a row-sharded one) plus the point ids and payloads, which never cross
    the C ABI and are indexed by the engine's row id = insertion order."""

    def __init__(self, dim, msizes, device, index=None):
        self.dim = dim
        self.msizes = tuple(msizes)
        self.index = index if index is not None else _engine.HxIndex(dim, self.msizes, device=device)
        self.ids: List[str] = []
        self.payloads: List[Dict[str, Any]] = []
        self.sparse_enabled = True

    def close(self):
        self.index.close()

    # on-disk form (the reference asks Qdrant for on_disk storage, qdrant_handler.py:47-55, 62):
    # <dir>/<user>.hx = the engine's file, <dir>/<user>.json = point ids + payloads
    def save(self, base: str):
        self.index.save(base + ".hx")
        with open(base + ".json", "w") as f:
            json.dump({"dim": self.dim, "msizes": list(self.msizes), "sparse_enabled": self.sparse_enabled,
                       "ids": self.ids, "payloads": self.payloads}, f)

    @classmethod
    def load(cls, base: str, device: int, index_loader=None):
        with open(base + ".json") as f:
            meta = json.load(f)
        self = cls.__new__(cls)
        self.dim = int(meta["dim"])
        self.msizes = tuple(meta["msizes"])
        self.index = (index_loader(base + ".hx", meta) if index_loader is not None
                      else _engine.HxIndex.load(base + ".hx", device=device))
        self.ids = list(meta["ids"])
        self.payloads = list(meta["payloads"])
        self.sparse_enabled = bool(meta["sparse_enabled"])
        return self


class QdrantHandler:
    """Handles vector operations for hybrid search with dense and sparse vectors."""

    def __init__(self, reranker=None, device: int = 0, persist_dir: Optional[str] = None):
        # The reference loads jinaai/jina-colbert-v2 here (:17-22) and falls back to the
        # un-reranked list whenever reranking raises (:410-412).  `reranker` is any object
        # with rerank_documents(query, documents, max_tokens) -> list of indices.
        self.reranker = reranker
        self.device = device
        # persist_dir (additive): collections found there are loaded on first use, `save_collection`
        # writes them back.  None = in-memory only.
        self.persist_dir = persist_dir
        self._collections: Dict[str, _Collection] = {}
        self._lock = threading.Lock()

    async def _run(self, fn, *a):
        loop = asyncio.get_running_loop()

        def locked():
            with self._lock:
                return fn(*a)
        return await loop.run_in_executor(None, locked)

    # ---------------------------------------------------------------- create_collection
    async def create_collection(self, user_id: str, dense_vector_size: int = 768,
                                matryoshka_sizes: list = [64, 128, 256], quantized_size: int = 768,
                                sparse_enabled: bool = True, force_recreate: bool = False):
        try:
            if not user_id:
                raise ValueError("user_id cannot be empty")
            user_id = str(user_id)
            if user_id in self._collections and not force_recreate:
                logging.info("create_collection: %s is already there", user_id)
                return
            if quantized_size != dense_vector_size:
                raise ValueError("quantized_size must equal dense_vector_size")

            def make():
                old = self._collections.pop(user_id, None)
                if old is not None:
                    self._drop(user_id, old)
                self._collections[user_id] = self._open_collection(
                    user_id, int(dense_vector_size), [int(m) for m in matryoshka_sizes], bool(sparse_enabled),
                    bool(force_recreate))
            await self._run(make)
            logging.info("create_collection: new collection for %s", user_id)
        except ValueError as ve:
            logging.error("create_collection(%s) refused: %s", user_id, ve)
            raise
        except Exception as e:
            logging.critical(f"Collection creation failed for user {user_id}: {str(e)}")
            raise

    # the two places a collection's engine index is made and dropped: sharded.ShardedHandler overrides them
    def _open_collection(self, user_id, dim, msizes, sparse_enabled, force_recreate) -> _Collection:
        base = self._base(user_id)
        if base and not force_recreate and os.path.exists(base + ".hx") and os.path.exists(base + ".json"):
            col = _Collection.load(base, self.device)
            if col.dim != dim:
                col.close()
                raise ValueError("stored collection has another vector size")
        else:
            col = _Collection(dim, msizes, self.device)
            col.sparse_enabled = sparse_enabled
        return col

    def _drop(self, user_id, col: _Collection) -> None:
        col.close()

    def _base(self, user_id: str) -> Optional[str]:
        if not self.persist_dir:
            return None
        safe = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in str(user_id))
        return os.path.join(self.persist_dir, safe)

    async def save_collection(self, user_id: str) -> None:
        """Write the collection to persist_dir (additive; Qdrant persists on its own)."""
        if not self.persist_dir:
            raise ValueError("handler was created without persist_dir")
        col = self._collections[str(user_id)]
        os.makedirs(self.persist_dir, exist_ok=True)
        await self._run(col.save, self._base(user_id))

    # --------------------------------------------------------------------------- upserts
    async def _store(self, user_id, items, emb_key_payload):
        if str(user_id) not in self._collections:
            await self.create_collection(user_id=user_id)
        col = self._collections[str(user_id)]
        dense, indptr, idx, val, ids, payloads = [], [0], [], [], [], []
        for item in items:
            if len(item["dense_embedding"]) != col.dim:
                raise ValueError(
                    f"Dense vector dimension mismatch. Expected {col.dim}, got {len(item['dense_embedding'])}")
            dense.append(np.asarray(item["dense_embedding"], dtype=np.float32))
            si, sv = _sparse_parts(item["sparse_embedding"]) if col.sparse_enabled else ([], [])
            idx.extend(int(i) for i in si)
            val.extend(float(v) for v in sv)
            indptr.append(len(idx))
            ids.append(str(uuid.uuid4()))
            payloads.append(emb_key_payload(item))
        if not dense:
            return 0

        def add():
            col.index.add(np.stack(dense), np.asarray(indptr, np.int64), np.asarray(idx, np.int32),
                          np.asarray(val, np.float32))
            col.ids.extend(ids)
            col.payloads.extend(payloads)
        await self._run(add)
        return len(dense)

    async def store_document_vectors(self, embedded_chunks: List[Dict[str, Any]], user_id: str):
        """Stores document chunks with multi-stage embeddings (qdrant_handler.py:120-198)."""
        try:
            def payload(chunk):
                metadata = chunk["chunk_metadata"]
                return {
                    "document_id": metadata["document_id"],
                    "user_id": metadata["user_id"],
                    "file_name": metadata["file_name"],
                    "mime_type": metadata["mime_type"],
                    "file_size": metadata["file_size"],
                    "file_description": metadata["description"],
                    "file_path": metadata["file_path"],
                    "context_version": metadata["context_version"],
                    "chunk_number": metadata["chunk_number"],
                    "entities": metadata.get("entities"),
                    "relationships": metadata.get("relationships"),
                    "context": metadata.get("context"),
                    "document_summary": metadata["doc_summary"],
                    "content": str(chunk["content"]),
                    "page_number": metadata.get("page_number"),
                    "languages": metadata.get("languages"),
                    "element_id": metadata.get("element_id"),
                    "is_continuation": metadata.get("is_continuation"),
                    "category": metadata.get("category"),
                }
            n = await self._store(user_id, embedded_chunks, payload)
            logging.info("store_document_vectors: %d chunks added for %s", n, user_id)
        except Exception as e:
            logging.error("store_document_vectors failed: %s", e)
            raise

    async def store_chat_vectors(self, embedded_payload: List[Dict[str, Any]], user_id: str):
        """Stores chat message vectors (qdrant_handler.py:200-267)."""
        try:
            def payload(chat):
                ts = chat["timestamp"]
                return {
                    "chat_id": chat["chat_id"],
                    "user_id": user_id,
                    "message_type": chat["message_type"],
                    "timestamp": ts.isoformat() if hasattr(ts, "isoformat") else ts,
                    "entities": chat["entities"],
                    "relationships": chat["relationships"],
                    "chat_summary": chat["chat_summary"],
                    "content": chat["message"],
                    "is_chat": True,
                }
            n = await self._store(user_id, embedded_payload, payload)
            logging.info("store_chat_vectors: %d messages added for %s", n, user_id)
        except Exception as e:
            logging.error("store_chat_vectors failed: %s", e)
            raise

    # ---------------------------------------------------------------------------- search
    def _search_sync(self, user_id, dense_vectors, sparse_vectors, search_params, filters, mode="tree"):
        col = self._collections[str(user_id)]
        q = np.asarray(dense_vectors, dtype=np.float32).reshape(len(sparse_vectors), -1)
        if q.shape[1] != col.dim:
            raise ValueError(f"query dimension {q.shape[1]} != collection dimension {col.dim}")
        parts = [_sparse_parts(sv) for sv in sparse_vectors]
        indptr = np.zeros(len(parts) + 1, np.int64)
        np.cumsum([len(si) for si, _ in parts], out=indptr[1:])
        nnz = int(indptr[-1])
        # (one pass over the batch's terms, no per-element int() / float() calls: the packing of a 1024-query batch is host
        # time the GPU waits for)
        idx = np.fromiter(itertools.chain.from_iterable(si for si, _ in parts), dtype=np.int64, count=nnz)
        val = np.fromiter(itertools.chain.from_iterable(vv for _, vv in parts), dtype=np.float64, count=nnz)
        if nnz and (idx.min() < -2 ** 31 or idx.max() >= 2 ** 31):
            raise ValueError("sparse index out of range")
        if mode not in ("tree", "h1"):
            raise ValueError("mode must be 'tree' (the reference query) or 'h1'")
        # KeyError/TypeError like the reference when search_params lacks a key / is None
        hp = _engine.make_params(search_params, mode=HX_MODE_TREE if mode == "tree" else HX_MODE_H1)
        final_limit = int(hp.final_limit)
        if filters and mode != "tree":
            raise ValueError("filters belong to the reference query's root (:297, :371): use mode='tree'")
        if filters:
            # query_filter belongs to the ROOT query only (:297, :371): the union of the branches'
            # candidates (<= dense_limit + the fusion's 10) is re-scored, filtered, cut to final_limit.
            # So: ask the engine for the whole re-scored union and filter it here.
            _filters.matches({}, filters)                  # validates the clause names before any GPU work
            hp.final_limit = min(int(hp.dense_limit) + int(hp.rrf_limit), 2048)
        scores, ids, counts = col.index.hybrid_query_host(q, indptr, idx.astype(np.int32), val.astype(np.float32), hp)
        out = []
        cids, cpay = col.ids, col.payloads
        for sc, rw, n in zip(scores.tolist(), ids.tolist(), counts.tolist()):     # (python floats / ints in one go)
            pts = [ScoredPoint(id=cids[r], version=0, score=s, payload=cpay[r]) for s, r in zip(sc[:n], rw[:n])]
            if filters:
                pts = [p for p in pts if _filters.matches(p.payload, filters, p.id)][:final_limit]
            out.append(pts)
        return out

    async def hybrid_search(self, user_id: str, query_text: str, dense_vector: List[float],
                            sparse_vector: Dict[str, List[float]], image_embedding: Optional[List[float]] = None,
                            top_k: int = 10, search_params: Optional[Dict[str, Any]] = None,
                            filters: Optional[Dict] = None) -> List[Dict]:
        """Matryoshka cascade, quantized + dense refinement, sparse, RRF, dense root
        re-score, reranking hook (qdrant_handler.py:269-386)."""
        try:
            results = (await self._run(self._search_sync, user_id, [dense_vector], [sparse_vector],
                                       search_params, filters))[0]
            max_tokens_per_doc = 8000 // top_k
            documents = [res.payload["content"] for res in results
                         if hasattr(res, "payload") and res.payload and "content" in res.payload]
            reranked_results = await self.rerank_with_colbert(query_text, documents, results, max_tokens_per_doc)
            return reranked_results[:top_k]
        except Exception as e:
            logging.error("hybrid search for %s failed: %s", user_id, e)
            return []

    async def hybrid_search_batch(self, user_id: str, dense_vectors, sparse_vectors, top_k: int = 10,
                                  search_params: Optional[Dict[str, Any]] = None,
                                  filters: Optional[Dict] = None, mode: str = "tree") -> List[List[ScoredPoint]]:
        """B queries in one engine call (additive; no reranking hook).  mode "tree" = the reference query
        (:305-372), "h1" = dense top-dense_limit (+) sparse top-sparse_limit -> RRF -> final_limit."""
        try:
            res = await self._run(self._search_sync, user_id, dense_vectors, sparse_vectors, search_params, filters,
                                  mode)
            return [r[:top_k] for r in res]
        except Exception as e:
            logging.error("hybrid search for %s failed: %s", user_id, e)
            return []

    async def rerank_with_colbert(self, query: str, documents: List[str], results: List[Dict],
                                  max_tokens: int) -> List[Dict]:
        """qdrant_handler.py:388-412: reorder by reranker indices; any failure keeps the order."""
        try:
            client = getattr(self.reranker, "client", self.reranker)
            ranked_indices = client.rerank_documents(query, documents, max_tokens)
            if not ranked_indices:
                return results
            return [results[i] for i in ranked_indices]
        except Exception as e:
            logging.error("reranker raised, order kept: %s", e)
            return results

    # ------------------------------------------------------------------------ collections
    async def get_all_containers(self) -> List[str]:
        try:
            return list(self._collections.keys())
        except Exception as e:
            logging.error("get_all_containers failed: %s", e)
            return []

    async def delete_collection(self, user_id: str):
        try:
            def drop():
                col = self._collections.pop(str(user_id))   # KeyError if absent: re-raised
                self._drop(str(user_id), col)
            await self._run(drop)
            logging.info("delete_collection: %s dropped", user_id)
        except Exception as e:
            logging.error("delete_collection(%s) failed: %s", user_id, e)
            raise

    async def get_collection_chunk_count(self, user_id: str, filters: Optional[Dict] = None) -> int:
        try:
            if str(user_id) not in self._collections:
                logging.warning("get_collection_chunk_count: no collection for %s", user_id)
                return 0
            col = self._collections[str(user_id)]
            if filters:   # :464-470: count the points the filter keeps
                return await self._run(lambda: sum(1 for i, p in zip(col.ids, col.payloads)
                                                   if _filters.matches(p, filters, i)))
            return await self._run(col.index.count)
        except Exception as e:
            logging.error("get_collection_chunk_count(%s) failed: %s", user_id, e)
            return 0
