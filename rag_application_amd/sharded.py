"""The multi-GPU front end of the drop-in: one user collection row-sharded over the ranks of a process group
(one process per GPU, RCCL over xGMI), driven from ONE rank the way the reference drives Qdrant.

The reference is single-process: `QdrantHandler.hybrid_search` (app/core/vector_store/qdrant/qdrant_handler.py:
269-386) and `store_document_vectors` (:120-198) are one call each.  Here they stay one call on the front rank
(`src`, rank 0): the query batch -- dense rows and the sparse CSR -- travels to the other ranks in ONE broadcast of
a packed buffer (SURVEY.md section 2, C2 `bcast_queries`), every rank runs the stages on its shard with one
all-gather per ranking stage (distributed.ShardedIndex, C1), and the front rank returns `ScoredPoint`s.  Ingest
deals a batch's chunks to the ranks in contiguous blocks (point-to-point), every rank derives and indexes its own
block -- with its own encoder replica when texts are dealt (data parallel, no gradient, no collective) -- and one
all-gather of the row counts tells every rank where each block sits in the collection's insertion order
("Ingest (cfg5)").

SPMD: every rank of the group calls every method in the same order; ranks other than `src` pass None for the data
and get None back.  `ShardedHandler.serve()` is that loop for the worker ranks of an application whose front
rank simply uses the handler.  Row ids inside the engine are `rank * stride + local row` (the engine's keys carry
global ids, distributed.py); the front rank maps them back to insertion order and payloads.  Equal scores break by
engine row id: for a collection ingested as one batch that is insertion order (as in the unsharded engine and
the oracle); with several batches, rank r's rows of a later batch precede rank r + 1's rows of an earlier one.
(The reference's point ids are uuid4, qdrant_handler.py:142: it defines no tie order at all.)
"""
from __future__ import annotations

import asyncio
import logging
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .distributed import ShardedIndex

ID_SPACE = 0xFFFFFFFE            # engine row ids stay below 2^32 - 1 (include/hx.h)


def _dev_of(group) -> torch.device:
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def bcast_queries(q, q_indptr, q_idx, q_val, src: int = 0, group=None, device: Optional[torch.device] = None,
                  header_group=None):
    """C2: the query batch from `src` to every rank.  A 3-word header (B, D, nnz), then ONE broadcast of a packed
    byte buffer [indptr int64 | Q float32 | idx int32 | val float32].  Returns the four tensors on every rank.
    `header_group`: a host-side (gloo) group with the same ranks for the header.  Over RCCL the header broadcast is
    queued behind everything the device still has to do, and reading its three words would park the host until
    the previous batch's kernels are through -- with a stream of batches in flight (distributed.H1Pipeline) the
    device would then idle while the host catches up.  Over the host group only the payload touches the device."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = device or _dev_of(group)
    if world == 1:
        return (torch.as_tensor(q, dtype=torch.float32, device=dev), torch.as_tensor(q_indptr, dtype=torch.int64, device=dev),
                torch.as_tensor(q_idx, dtype=torch.int32, device=dev), torch.as_tensor(q_val, dtype=torch.float32, device=dev))
    # the collective runs where the backend works: on the device for RCCL, on the host for gloo (CPU tests, rehearsals)
    cdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    hdev = torch.device("cpu") if header_group is not None else cdev
    head = torch.zeros(3, dtype=torch.int64, device=hdev)
    if rank == src:
        q = torch.as_tensor(q, dtype=torch.float32).contiguous()
        q_indptr = torch.as_tensor(q_indptr, dtype=torch.int64).contiguous()
        q_idx = torch.as_tensor(q_idx, dtype=torch.int32).contiguous()
        q_val = torch.as_tensor(q_val, dtype=torch.float32).contiguous()
        head = torch.tensor([q.shape[0], q.shape[1], q_idx.shape[0]], dtype=torch.int64, device=hdev)
    dist.broadcast(head, src, group=header_group if header_group is not None else group)
    B, D, nnz = (int(x) for x in head.tolist())
    sizes = [(B + 1) * 8, B * D * 4, nnz * 4, nnz * 4]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=cdev)
    if rank == src:
        o = 0
        for t, n in zip((q_indptr, q, q_idx, q_val), sizes):
            if n:
                buf[o:o + n] = t.to(cdev).reshape(-1).view(torch.uint8)
            o += n
    dist.broadcast(buf, src, group=group)
    buf = buf.to(dev)
    o0, o1, o2, o3 = np.cumsum([0] + sizes[:3]).tolist()
    return (buf[o1:o1 + sizes[1]].view(torch.float32).view(B, D), buf[o0:o0 + sizes[0]].view(torch.int64),
            buf[o2:o2 + sizes[2]].view(torch.int32), buf[o3:o3 + sizes[3]].view(torch.float32))


class ShardedCollection:
    """One collection over the ranks of `group`.  `index_factory(dim, msizes, id_base)` builds this rank's shard
    (engine.HxIndex by default; the gloo tests inject an oracle-backed stand-in), `ops` as in ShardedIndex."""

    def __init__(self, dim: int = 768, msizes: Sequence[int] = (64, 128, 256), group=None, index_factory=None,
                 ops=None, src: int = 0):
        self.dim, self.msizes, self.group, self.src = int(dim), tuple(msizes), group, src
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dev = _dev_of(group)
        self.stride = ID_SPACE // self.world
        self.id_base = self.rank * self.stride
        if index_factory is None:
            from . import engine as _engine

            def index_factory(dim, msizes, id_base):
                return _engine.HxIndex(dim, msizes, device=self.dev.index or 0, id_base=id_base)
        self.local = index_factory(self.dim, self.msizes, self.id_base)
        self.sh = ShardedIndex(self.local, group, ops)
        self.counts = np.zeros(self.world, np.int64)      # rows per rank (every rank knows them all)
        self.total = 0
        # front rank: insertion-order position of (rank, local row): one array per rank
        self.seq_of: List[List[np.ndarray]] = [[] for _ in range(self.world)]

    # ---------------------------------------------------------------------------------- ingest
    def _deal(self, n: int):
        """contiguous blocks: rank r owns batch rows [n r / W, n (r + 1) / W)"""
        return [n * r // self.world for r in range(self.world + 1)]

    def store(self, dense=None, sp_indptr=None, sp_idx=None, sp_val=None, texts=None, encoder=None, sparse_embed=None):
        """One batch.  The front rank passes either `dense` [n x dim] (+ optional sparse CSR) or `texts`; with
        `texts` every rank encodes its own block (`encoder.encode(list[str]) -> [m x dim]` tensor or array on this
        rank's device, `sparse_embed(list[str]) -> (indptr, idx, val)`).  Returns, on the front rank, the
        insertion-order positions of the batch's rows (others: None)."""
        W, r = self.world, self.rank
        head = torch.zeros(3, dtype=torch.int64, device=self.dev)
        if r == self.src:
            n = len(texts) if texts is not None else int(np.asarray(dense).shape[0])
            head = torch.tensor([n, 1 if texts is not None else 0, 1 if (sp_indptr is not None) else 0],
                                dtype=torch.int64, device=self.dev)
        if W > 1:
            dist.broadcast(head, self.src, group=self.group)
        n, is_text, has_sp = (int(x) for x in head.tolist())
        cut = self._deal(n)
        m = cut[r + 1] - cut[r]
        # ---- the rank's block
        if is_text:
            if W > 1:
                out = [None]
                dist.scatter_object_list(out, [texts[cut[j]:cut[j + 1]] for j in range(W)] if r == self.src else None,
                                         src=self.src, group=self.group)
                block = out[0]
            else:
                block = list(texts)
            mine_dense = encoder.encode(block) if m else np.zeros((0, self.dim), np.float32)
            mine_sp = sparse_embed(block) if (sparse_embed is not None and m) else None
        else:
            mine_dense, mine_sp = self._deal_arrays(dense, sp_indptr, sp_idx, sp_val, cut, has_sp)
        # ---- derive + index locally (K1/K2 on ingest, K9 on the next search)
        if m:
            sp = mine_sp if mine_sp is not None else (None, None, None)
            if isinstance(mine_dense, torch.Tensor) and mine_dense.is_cuda and hasattr(self.local, "add_device"):
                self.local.add_device(mine_dense.float().contiguous(), *sp)
            else:
                self.local.add(np.asarray(mine_dense.cpu() if isinstance(mine_dense, torch.Tensor) else mine_dense,
                                          np.float32), *sp)
        # ---- one all-gather of the row counts: where every block sits in the insertion order
        cnt = torch.tensor([m], dtype=torch.int64, device=self.dev)
        if W > 1:
            if self.dev.type == "cuda":
                allc = torch.empty(W, dtype=torch.int64, device=self.dev)
                dist.all_gather_into_tensor(allc, cnt, group=self.group)
            else:                                   # gloo (CPU tests / rehearsals)
                parts = [torch.empty(1, dtype=torch.int64) for _ in range(W)]
                dist.all_gather(parts, cnt, group=self.group)
                allc = torch.cat(parts)
            allc = allc.cpu().numpy()
        else:
            allc = np.asarray([m], np.int64)
        assert int(allc.sum()) == n and (self.counts + allc < self.stride).all(), "shard id space exhausted"
        base = self.total + np.concatenate([[0], np.cumsum(allc)[:-1]])      # id base of each rank's block
        if r == self.src:
            for j in range(W):
                self.seq_of[j].append(base[j] + np.arange(allc[j], dtype=np.int64))
        self.counts += allc
        first = self.total
        self.total += n
        return np.arange(first, first + n) if r == self.src else None

    def _deal_arrays(self, dense, ip, ix, v, cut, has_sp):
        W, r = self.world, self.rank
        m = cut[r + 1] - cut[r]
        if W == 1:
            return np.asarray(dense, np.float32), ((np.asarray(ip, np.int64), np.asarray(ix, np.int32), np.asarray(v, np.float32))
                                                    if has_sp else None)
        meta = torch.zeros(1, dtype=torch.int64, device=self.dev)           # nnz of the rank's block
        if r == self.src:
            dense = torch.as_tensor(np.asarray(dense, np.float32))
            if has_sp:
                ip = np.asarray(ip, np.int64)
                ix_t, v_t = torch.as_tensor(np.asarray(ix, np.int32)), torch.as_tensor(np.asarray(v, np.float32))
            metas = [torch.tensor([int(ip[cut[j + 1]] - ip[cut[j]]) if has_sp else 0], dtype=torch.int64, device=self.dev)
                     for j in range(W)]
            dist.scatter(meta, metas, src=self.src, group=self.group)
        else:
            dist.scatter(meta, None, src=self.src, group=self.group)
        nnz = int(meta.item())
        my_d = torch.empty((m, self.dim), dtype=torch.float32, device=self.dev)
        my_ip = torch.empty(m + 1, dtype=torch.int64, device=self.dev)
        my_ix = torch.empty(nnz, dtype=torch.int32, device=self.dev)
        my_v = torch.empty(nnz, dtype=torch.float32, device=self.dev)
        if r == self.src:
            reqs = []
            for j in range(W):
                a, b = cut[j], cut[j + 1]
                parts = [dense[a:b]]
                if has_sp:
                    parts += [torch.as_tensor(ip[a:b + 1] - ip[a]), ix_t[ip[a]:ip[b]], v_t[ip[a]:ip[b]]]
                if j == r:
                    for dst, p in zip((my_d, my_ip, my_ix, my_v), parts):
                        dst.copy_(p)
                else:
                    reqs += [dist.isend(p.contiguous().to(self.dev), j, group=self.group) for p in parts if p.numel()]
            for q in reqs:
                q.wait()
        else:
            for t in ((my_d, my_ip, my_ix, my_v) if has_sp else (my_d,)):
                if t.numel():
                    dist.recv(t, self.src, group=self.group)
        sp = (my_ip.cpu().numpy(), my_ix.cpu().numpy(), my_v.cpu().numpy()) if has_sp else None
        return (my_d if my_d.is_cuda else my_d.numpy()), sp

    # ---------------------------------------------------------------------------------- search
    def search(self, q=None, q_indptr=None, q_idx=None, q_val=None, params: Optional[dict] = None, mode: str = "tree",
               rrf_k: float = 2.0, rank_base: int = 0):
        """The query batch of the front rank through the sharded path: (keys [B x final_limit], counts [B]) on
        every rank (replicated).  mode "tree" = the reference query, "h1" = dense (+) sparse -> RRF."""
        if self.world > 1:
            meta = [params, mode] if self.rank == self.src else [None, None]
            dist.broadcast_object_list(meta, self.src, group=self.group)
            params, mode = meta
        q, ip, ix, v = bcast_queries(q, q_indptr, q_idx, q_val, self.src, self.group, self.dev)
        if mode == "h1":
            return self.sh.hybrid_h1(q, ip, ix, v, params["dense_limit"], params["sparse_limit"], params["final_limit"],
                                     rrf_k, rank_base)
        return self.sh.hybrid_tree(q, ip, ix, v, params, self.msizes, rrf_k, rank_base)

    def resolve(self, keys: torch.Tensor, counts: Optional[torch.Tensor]):
        """Front rank: engine keys -> per query [(insertion-order position, score)], best first."""
        k = keys.cpu().numpy().view(np.uint64)
        ids = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
        u = (k >> np.uint64(32)).astype(np.uint32)
        u = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
        sc = u.view(np.float32)
        seqs = [np.concatenate(s) if s else np.zeros(0, np.int64) for s in self.seq_of]
        out = []
        for b in range(k.shape[0]):
            n = int(counts[b]) if counts is not None else int((k[b] != 0).sum())
            row = []
            for j in range(n):
                r, loc = divmod(int(ids[b, j]), self.stride)
                row.append((int(seqs[r][loc]), float(sc[b, j])))
            out.append(row)
        return out

    def count(self) -> int:
        return int(self.total)

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()


class ShardedHandler:
    """`QdrantHandler`'s methods over a row-sharded collection per user (same names, argument meaning and error
    conventions as qdrant_handler.py:14-481; see handler.QdrantHandler for the single-GPU form).  The front rank
    holds point ids and payloads; worker ranks hold only their shard and run `serve()`."""

    def __init__(self, group=None, index_factory=None, ops=None, src: int = 0, dense_vector_size: int = 768,
                 matryoshka_sizes: Sequence[int] = (64, 128, 256)):
        self.group, self.factory, self.ops, self.src = group, index_factory, ops, src
        self.dim, self.msizes = dense_vector_size, tuple(matryoshka_sizes)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._col: Dict[str, ShardedCollection] = {}
        self._payloads: Dict[str, List[Dict[str, Any]]] = {}
        self._ids: Dict[str, List[str]] = {}

    # ---- the command channel: the front rank announces every call so that worker ranks can follow
    def _announce(self, op: str, user_id: Optional[str]):
        msg = [op, user_id] if self.rank == self.src else [None, None]
        if self.world > 1:
            dist.broadcast_object_list(msg, self.src, group=self.group)
        return msg

    def serve(self):
        """Worker ranks: follow the front rank's calls until it calls shutdown()."""
        assert self.rank != self.src
        while True:
            op, user = self._announce(None, None)
            if op == "shutdown":
                return
            if op == "create":
                self._create(user)
            elif op == "store":
                self._col[user].store()
            elif op == "search":
                self._col[user].search()
            elif op == "delete":
                self._delete(user)

    def shutdown(self):
        self._announce("shutdown", None)

    def _create(self, user_id):
        if user_id not in self._col:
            self._col[user_id] = ShardedCollection(self.dim, self.msizes, self.group, self.factory, self.ops, self.src)
            self._payloads[user_id], self._ids[user_id] = [], []

    def _delete(self, user_id):
        col = self._col.pop(user_id, None)
        if col is not None:
            col.close()
        self._payloads.pop(user_id, None)
        self._ids.pop(user_id, None)

    # ---- QdrantHandler surface (front rank) -------------------------------------------------------------------
    async def create_collection(self, user_id: str, **_kw) -> None:
        if not user_id:
            raise ValueError("user_id cannot be empty")                    # qdrant_handler.py:39-40
        self._announce("create", str(user_id))
        self._create(str(user_id))

    async def store_document_vectors(self, embedded_chunks: List[Dict[str, Any]], user_id: str) -> None:
        """qdrant_handler.py:120-198: the batch is dealt to the ranks in contiguous blocks."""
        try:
            import uuid
            user_id = str(user_id)
            if user_id not in self._col:
                await self.create_collection(user_id)
            dense = np.asarray([c["dense_embedding"] for c in embedded_chunks], np.float32)
            if dense.ndim != 2 or dense.shape[1] != self.dim:
                raise ValueError(f"Dense vector dimension mismatch. Expected {self.dim}, got {dense.shape[-1]}")   # :138-139
            ip, ix, v = [0], [], []
            for c in embedded_chunks:
                sv = c.get("sparse_embedding")
                si, vv = (sv["indices"], sv["values"]) if isinstance(sv, dict) else ((sv.indices, sv.values) if sv is not None else ([], []))
                ix.extend(int(i) for i in si)
                v.extend(float(x) for x in vv)
                ip.append(len(ix))
            self._announce("store", user_id)
            self._col[user_id].store(dense, np.asarray(ip, np.int64), np.asarray(ix, np.int32), np.asarray(v, np.float32))
            for c in embedded_chunks:
                md = c["chunk_metadata"]
                self._ids[user_id].append(str(uuid.uuid4()))                # :142
                self._payloads[user_id].append({**{k: md.get(k) for k in md}, "content": str(c["content"]),
                                                "document_summary": md.get("doc_summary"),
                                                "file_description": md.get("description")})
        except Exception as e:
            logging.error("store_document_vectors failed: %s", e)
            raise                                                           # :196-198

    async def hybrid_search_batch(self, user_id: str, dense_vectors, sparse_vectors, top_k: int = 10,
                                  search_params: Optional[Dict[str, Any]] = None, mode: str = "tree"):
        from .handler import ScoredPoint, _sparse_parts
        try:
            user_id = str(user_id)
            col = self._col[user_id]
            params = {k: int(search_params[k]) for k in ("matryoshka_64_limit", "matryoshka_128_limit",
                                                         "matryoshka_256_limit", "dense_limit", "quantized_limit",
                                                         "sparse_limit", "final_limit", "hnsw_ef")}
            q = np.asarray(dense_vectors, np.float32).reshape(len(sparse_vectors), -1)
            ip, ix, v = [0], [], []
            for sv in sparse_vectors:
                si, vv = _sparse_parts(sv)
                order = np.argsort(np.asarray(si, np.int64), kind="stable")     # ascending term id: the sum's order
                ix.extend(int(si[o]) for o in order)
                v.extend(float(vv[o]) for o in order)
                ip.append(len(ix))
            self._announce("search", user_id)
            keys, cnt = col.search(q, np.asarray(ip, np.int64), np.asarray(ix, np.int32), np.asarray(v, np.float32), params, mode)
            out = []
            for row in col.resolve(keys, cnt):
                out.append([ScoredPoint(id=self._ids[user_id][s], version=0, score=sc, payload=self._payloads[user_id][s])
                            for s, sc in row][:top_k])
            return out
        except Exception as e:
            logging.error("hybrid search failed: %s", e)
            return [[] for _ in sparse_vectors] if sparse_vectors is not None else []     # :384-386

    async def hybrid_search(self, user_id: str, query_text: str, dense_vector, sparse_vector, image_embedding=None,
                            top_k: int = 10, search_params: Optional[Dict[str, Any]] = None, filters=None):
        if search_params is None:
            return []                                                       # :314 indexes None -> TypeError -> []
        res = await self.hybrid_search_batch(user_id, [dense_vector], [sparse_vector], top_k, search_params)
        return res[0] if res else []

    async def get_collection_chunk_count(self, user_id: str, filters=None) -> int:
        col = self._col.get(str(user_id))
        return col.count() if col is not None else 0                       # :479-481

    async def get_all_containers(self) -> List[str]:
        return list(self._col)

    async def delete_collection(self, user_id: str) -> None:
        self._announce("delete", str(user_id))
        self._delete(str(user_id))
