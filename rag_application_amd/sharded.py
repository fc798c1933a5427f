"""The multi-GPU front end of the drop-in: one user collection row-sharded over the ranks of a process group
(one process per GPU, RCCL over xGMI), driven from ONE rank the way the reference drives Qdrant.

The reference is single-process: `QdrantHandler.hybrid_search` (app/core/vector_store/qdrant/qdrant_handler.py:
269-386) and `store_document_vectors` (:120-198) are one call each.  Here they stay one call on the front rank
(`src`, rank 0): the query batch -- dense rows and the sparse CSR -- travels to the other ranks in ONE broadcast of
a packed buffer (SURVEY.md section 2, C2 `bcast_queries`), every rank runs the stages on its shard with one
all-gather per ranking stage (distributed.ShardedIndex, C1), and the front rank returns `ScoredPoint`s.  Ingest
deals a batch's chunks to the ranks in contiguous blocks (point-to-point), every rank derives and indexes its own
block -- with its own encoder replica when texts are dealt (data parallel, no gradient, no collective) -- and one
all-gather of the blocks' outcomes tells every rank that the batch is stored ("Ingest (cfg5)").

Row ids.  The reference upserts a collection in MANY batches (one per document, app/services/file_processor/
text_processor.py:357 -> qdrant_handler.py:190-193), so a shard holds a slice of every batch.  A row's id is its
position in the collection's INSERTION ORDER, on every rank: before a rank stores its block of a batch it names the
block's first id (`hx_set_next_id`: total rows so far + the block's offset in the batch), and every key that leaves
a shard carries those ids.  The total order (score desc, id asc) of the sharded collection is therefore the order
of the same collection on one GPU -- ties included -- whatever the number of batches or ranks.

Failures (the reference: search never raises, mutations re-raise, :384-386, :196-198).  Commands and outcomes travel
over a host-side control group with a timeout.  Every call is: announce -> every rank acknowledges (or reports why
it cannot take part: nobody then enters a data-path collective) -> the data path, in which a rank whose local work
raises goes on with empty lists so that no collective is left waiting -> one exchange of the outcomes.  A failed
rank makes the call fail on EVERY rank: a batch is rolled back where it had been stored (`hx_truncate`) and the
front rank raises; a search raises on the front rank, which `hybrid_search` turns into [].  A rank that died or hangs
shows as a control-group timeout on the front rank within `timeout` seconds.

SPMD: every rank of the group calls every method of `ShardedCollection` in the same order; ranks other than `src`
pass None for the data.  `ShardedHandler.serve()` is that loop for the worker ranks of an application whose front
rank simply uses the handler.
"""
from __future__ import annotations

import datetime
import json
import logging
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .distributed import ShardedIndex
from .handler import QdrantHandler, _Collection

ID_SPACE = 0xFFFFFFFE            # row ids stay below 2^32 - 1 (include/hx.h)
SPARSE_ABS_MAX = 1.0e18          # what hx_add_sparse accepts (engine.hip)


class ShardError(RuntimeError):
    """A call failed on some rank (or a rank did not answer in time): it failed on every rank."""


class ShardOutOfStep(ShardError):
    """A collective step itself failed (a rank died or timed out inside the deal of a batch): the ranks no longer
    agree on where they are, so the handler that owns the group closes (searches return [], mutations raise)."""


def _dev_of(group) -> torch.device:
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _control_group(group, timeout: Optional[float] = None):
    """The host-side group commands and outcomes travel over: `group` itself when it is a gloo group and no timeout
    is asked for, else a new gloo group over the same ranks (collective: every rank calls it)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    if timeout is None and dist.get_backend(group) == "gloo":
        return group
    ranks = dist.get_process_group_ranks(group) if group is not None else None
    kw = dict(timeout=datetime.timedelta(seconds=float(timeout))) if timeout is not None else {}
    return dist.new_group(ranks=ranks, backend="gloo", **kw)


def check_sparse_rows(indptr: np.ndarray, idx: np.ndarray, val: np.ndarray) -> None:
    """What hx_add_sparse checks (engine.hip: add_sparse_host), on the front rank BEFORE a batch is announced: a
    batch no shard would accept never reaches the ranks.  Monotone indptr, term ids in [0, 2^31) and unique within a
    vector (Qdrant rejects duplicates), finite values with |v| <= 1e18."""
    indptr = np.asarray(indptr, np.int64)
    if indptr.ndim != 1 or indptr.size < 1 or indptr[0] != 0 or (np.diff(indptr) < 0).any() or indptr[-1] != len(idx):
        raise ValueError("sparse indptr must start at 0, be monotone and end at nnz")
    if len(idx) == 0:
        return
    ix = np.asarray(idx, np.int64)
    v = np.asarray(val, np.float64)
    if (ix < 0).any() or (ix >= 2 ** 31).any():
        raise ValueError("sparse index out of range [0, 2^31)")
    if not np.isfinite(v).all() or (np.abs(v) > SPARSE_ABS_MAX).any():
        raise ValueError("sparse values must be finite and at most 1e18 in magnitude")
    rows = np.repeat(np.arange(indptr.size - 1), np.diff(indptr))
    o = np.lexsort((ix, rows))
    if ((rows[o][1:] == rows[o][:-1]) & (ix[o][1:] == ix[o][:-1])).any():
        raise ValueError("sparse indices must be unique within a vector")


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
    (engine.HxIndex by default; the gloo tests inject an oracle-backed stand-in), `ops` as in ShardedIndex, `ctl` the
    host-side control group (made here when not given)."""

    def __init__(self, dim: int = 768, msizes: Sequence[int] = (64, 128, 256), group=None, index_factory=None,
                 ops=None, src: int = 0, ctl=None, local=None, counts=None, device: Optional[torch.device] = None):
        self.dim, self.msizes, self.group, self.src = int(dim), tuple(msizes), group, src
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # `dev`: where this rank's shard lives and the stages run; `cdev`: where the collectives run (the device for
        # RCCL, the host for gloo -- rehearsals and tests, also with real shards on a GPU: device=cuda over a gloo group)
        self.cdev = _dev_of(group)
        self.dev = device if device is not None else self.cdev
        self.ctl = ctl if ctl is not None else _control_group(group)
        # C2 gets a communicator of its own over RCCL: ProcessGroupNCCL runs a group's collectives in order on one
        # internal stream, so the broadcast of batch i + 1 issued on the stages' group would queue behind the
        # all-gather of batch i (collective: every rank makes it here, in the same order)
        self.bgroup = group
        if dist.is_initialized() and self.world > 1 and dist.get_backend(group) == "nccl":
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.bgroup = dist.new_group(ranks=ranks, backend="nccl")
        if local is None:
            if index_factory is None:
                from . import engine as _engine

                def index_factory(dim, msizes, id_base):
                    return _engine.HxIndex(dim, msizes, device=self.dev.index or 0, id_base=id_base)
            local = index_factory(self.dim, self.msizes, 0)     # ids are named batch by batch (set_next_id)
        self.local = local
        self.sh = ShardedIndex(self.local, group, ops)
        self.counts = np.zeros(self.world, np.int64) if counts is None else np.asarray(counts, np.int64).copy()
        self.total = int(self.counts.sum())                      # rows of the collection = the next insertion id

    # ---------------------------------------------------------------------------------- control plane
    def _bcast_obj(self, obj):
        if self.world == 1:
            return obj
        box = [obj if self.rank == self.src else None]
        dist.broadcast_object_list(box, self.src, group=self.ctl)
        return box[0]

    def _outcomes(self, mine):
        """every rank's outcome of the step, on every rank (host-side, small, with the control group's timeout)"""
        if self.world == 1:
            return [mine]
        out = [None] * self.world
        dist.all_gather_object(out, mine, group=self.ctl)
        return out

    # ---------------------------------------------------------------------------------- ingest
    def _deal(self, n: int):
        """contiguous blocks: rank r owns batch rows [n r / W, n (r + 1) / W)"""
        return [n * r // self.world for r in range(self.world + 1)]

    def store(self, dense=None, sp_indptr=None, sp_idx=None, sp_val=None, texts=None, encoder=None, sparse_embed=None):
        """One batch, all ranks or none.  The front rank passes either `dense` [n x dim] (+ optional sparse CSR) or
        `texts`; with `texts` every rank encodes its own block (`encoder.encode(list[str]) -> [m x dim]` tensor or
        array on this rank's device, `sparse_embed(list[str]) -> (indptr, idx, val)`).  Returns, on the front rank,
        the insertion-order ids of the batch's rows (others: None); raises ShardError on every rank when any rank
        could not store its block (the others roll theirs back)."""
        W, r = self.world, self.rank
        head = None
        if r == self.src:
            # a batch the front rank itself refuses still has to reach the ranks that wait for its header: as an error
            try:
                n = len(texts) if texts is not None else int(np.asarray(dense).shape[0])
                if texts is None and np.asarray(dense).shape[1:] != (self.dim,):
                    raise ValueError(f"Dense vector dimension mismatch. Expected {self.dim}, got {np.asarray(dense).shape[-1]}")
                if sp_indptr is not None:
                    check_sparse_rows(sp_indptr, sp_idx, sp_val)
                    if len(sp_indptr) != n + 1:
                        raise ValueError("sparse indptr must have n+1 entries")
                if self.total + n >= ID_SPACE:
                    raise ValueError("row ids must stay below 2^32 - 1")
                head = (n, texts is not None, sp_indptr is not None)
            except Exception as e:
                head = ("refused", f"{type(e).__name__}: {e}", None)
        head = self._bcast_obj(head)
        if head[0] == "refused":
            raise ShardError("batch refused by the front rank: " + head[1])
        n, is_text, has_sp = head
        cut = self._deal(n)
        m = cut[r + 1] - cut[r]
        rows_before = int(self.counts[r])
        err = None
        # ---- the rank's block (the point-to-point deal is a collective step: every rank takes part)
        try:
            if is_text:
                if W > 1:
                    out = [None]
                    dist.scatter_object_list(out, [texts[cut[j]:cut[j + 1]] for j in range(W)] if r == self.src else None,
                                             src=self.src, group=self.ctl)
                    block = out[0]
                else:
                    block = list(texts)
                mine_dense, mine_sp = None, None
            else:
                mine_dense, mine_sp = self._deal_arrays(dense, sp_indptr, sp_idx, sp_val, cut, has_sp)
        except Exception as e:                # a broken deal leaves the ranks out of step: nothing to salvage here
            raise ShardOutOfStep(f"rank {r}: dealing the batch failed: {e}") from e
        # ---- derive + index locally (K1/K2 on ingest, K9 on the next search); a failure is reported, not raised yet
        try:
            if is_text and m:
                mine_dense = encoder.encode(block)
                mine_sp = sparse_embed(block) if sparse_embed is not None else None
            if m:
                sp = mine_sp if mine_sp is not None else (None, None, None)
                self.local.set_next_id(self.total + cut[r])
                if isinstance(mine_dense, torch.Tensor) and mine_dense.is_cuda and hasattr(self.local, "add_device"):
                    self.local.add_device(mine_dense.float().contiguous(), *sp)
                else:
                    self.local.add(np.asarray(mine_dense.cpu() if isinstance(mine_dense, torch.Tensor) else mine_dense,
                                              np.float32), *sp)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        # ---- one exchange of the outcomes: the batch is stored everywhere or nowhere
        outs = self._outcomes(err)
        bad = [j for j, o in enumerate(outs) if o is not None]
        if bad:
            if err is None and m:
                self.local.truncate(rows_before)                   # roll this rank's block back
            raise ShardError("batch not stored: " + "; ".join(f"rank {j}: {outs[j]}" for j in bad))
        self.counts += np.diff(cut)
        first = self.total
        self.total += n
        return np.arange(first, first + n) if r == self.src else None

    def _deal_arrays(self, dense, ip, ix, v, cut, has_sp):
        W, r = self.world, self.rank
        m = cut[r + 1] - cut[r]
        if W == 1:
            return np.asarray(dense, np.float32), ((np.asarray(ip, np.int64), np.asarray(ix, np.int32), np.asarray(v, np.float32))
                                                    if has_sp else None)
        nnzs = None
        if r == self.src:
            dense = torch.as_tensor(np.asarray(dense, np.float32))
            if has_sp:
                ip = np.asarray(ip, np.int64)
                ix_t, v_t = torch.as_tensor(np.asarray(ix, np.int32)), torch.as_tensor(np.asarray(v, np.float32))
            nnzs = [int(ip[cut[j + 1]] - ip[cut[j]]) if has_sp else 0 for j in range(W)]
        nnz = self._bcast_obj(nnzs)[r]                                      # nnz of the rank's block
        my_d = torch.empty((m, self.dim), dtype=torch.float32, device=self.cdev)
        my_ip = torch.empty(m + 1, dtype=torch.int64, device=self.cdev)
        my_ix = torch.empty(nnz, dtype=torch.int32, device=self.cdev)
        my_v = torch.empty(nnz, dtype=torch.float32, device=self.cdev)
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
                    reqs += [dist.isend(p.contiguous().to(self.cdev), j, group=self.group) for p in parts if p.numel()]
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
               rrf_k: float = 2.0, rank_base: int = 0, rrf_limit: int = 10):
        """The query batch of the front rank through the sharded path: (keys [B x final_limit], counts [B]) on
        every rank (replicated); ids are insertion-order positions.  mode "tree" = the reference query, "h1" = dense
        (+) sparse -> RRF.  `q_idx` strictly ascending within a query.  Raises ShardError on every rank when a
        rank's local stage failed."""
        head = None
        if self.rank == self.src:          # (a bad batch on the front rank reaches the waiting ranks as an error)
            try:
                q = torch.as_tensor(np.asarray(q, np.float32)).reshape(-1, self.dim)
                q_indptr = torch.as_tensor(np.asarray(q_indptr, np.int64))
                q_idx, q_val = torch.as_tensor(np.asarray(q_idx, np.int32)), torch.as_tensor(np.asarray(q_val, np.float32))
                if q_indptr.shape[0] != q.shape[0] + 1 or q_idx.shape != q_val.shape or int(q_indptr[-1]) != q_idx.shape[0]:
                    raise ValueError("sparse query CSR does not match the batch")
                for k in ("dense_limit", "sparse_limit", "final_limit"):
                    int(params[k])
                if mode not in ("tree", "h1"):
                    raise ValueError("mode must be 'tree' or 'h1'")
                head = (params, mode, rrf_k, rank_base, rrf_limit)
            except Exception as e:
                head = ("refused", f"{type(e).__name__}: {e}")
        head = self._bcast_obj(head)
        if head[0] == "refused":
            raise ShardError("query batch refused by the front rank: " + head[1])
        params, mode, rrf_k, rank_base, rrf_limit = head
        q, ip, ix, v = bcast_queries(q, q_indptr, q_idx, q_val, self.src, self.bgroup, self.dev,
                                     header_group=self.ctl if self.world > 1 else None)
        if mode == "h1":
            out = self.sh.hybrid_h1(q, ip, ix, v, params["dense_limit"], params["sparse_limit"], params["final_limit"],
                                    rrf_k, rank_base)
        else:
            out = self.sh.hybrid_tree(q, ip, ix, v, params, self.msizes, rrf_k, rank_base, rrf_limit)
        e = self.sh.take_error()
        outs = self._outcomes(None if e is None else f"{type(e).__name__}: {e}")
        bad = [j for j, o in enumerate(outs) if o is not None]
        if bad:
            raise ShardError("search failed: " + "; ".join(f"rank {j}: {outs[j]}" for j in bad))
        return out

    @staticmethod
    def unpack(keys: torch.Tensor, counts: Optional[torch.Tensor]):
        """engine keys -> (scores f32 [B x L], ids i64 [B x L], counts i32 [B]) as numpy (empty slots: -inf, -1)"""
        k = keys.cpu().numpy().view(np.uint64)
        ids = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
        u = (k >> np.uint64(32)).astype(np.uint32)
        u = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
        sc = u.view(np.float32).copy()
        cnt = (k != 0).sum(axis=1).astype(np.int32) if counts is None else counts.cpu().numpy().astype(np.int32)
        sc[k == 0] = -np.inf
        ids[k == 0] = -1
        return sc, ids, cnt

    def resolve(self, keys: torch.Tensor, counts: Optional[torch.Tensor]):
        """engine keys -> per query [(insertion-order position, score)], best first"""
        sc, ids, cnt = self.unpack(keys, counts)
        return [[(int(ids[b, j]), float(sc[b, j])) for j in range(int(cnt[b]))] for b in range(ids.shape[0])]

    def count(self) -> int:
        return int(self.total)

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()


class _ShardedBackend:
    """What handler._Collection holds as its `index` on the FRONT rank: HxIndex's add / hybrid_query_host / count /
    save / close, each announced to the worker ranks and run by all ranks together."""

    def __init__(self, handler: "ShardedHandler", user_id: str, col: ShardedCollection):
        self.h, self.user, self.col = handler, user_id, col

    def _together(self, fn):
        """a step all ranks run together: what escapes it other than a refusal is a collective that failed (a rank
        died or timed out) -- the ranks are out of step from then on and the handler closes"""
        try:
            return fn()
        except ShardOutOfStep as e:
            self.h.broken = f"{e}; the sharded handler is closed"
            raise
        except (ShardError, ValueError, KeyError, TypeError):
            raise
        except Exception as e:
            self.h.broken = f"a rank did not answer ({type(e).__name__}: {e}); the sharded handler is closed"
            raise ShardError(self.h.broken) from e

    def add(self, dense, sp_indptr=None, sp_idx=None, sp_val=None):
        dense = np.ascontiguousarray(dense, np.float32)
        if dense.ndim != 2 or dense.shape[1] != self.col.dim:
            raise ValueError(f"Dense vector dimension mismatch. Expected {self.col.dim}, got {dense.shape[-1]}")
        if sp_indptr is not None:
            check_sparse_rows(sp_indptr, sp_idx, sp_val)
        self.h._command("store", self.user)
        self._together(lambda: self.col.store(dense, sp_indptr, sp_idx, sp_val))

    def hybrid_query_host(self, q, q_indptr, q_idx, q_val, hp):
        """hx_hybrid_query_host's contract: terms sorted by id per query (the sum's order), duplicates refused."""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, self.col.dim)
        q_indptr = np.asarray(q_indptr, np.int64)
        q_idx, q_val = np.asarray(q_idx, np.int64), np.asarray(q_val, np.float32)
        if (q_idx < 0).any() or (q_idx >= 2 ** 31).any() or not np.isfinite(q_val).all():
            raise ValueError("sparse query: indices in [0, 2^31), finite values")
        rows = np.repeat(np.arange(q_indptr.size - 1), np.diff(q_indptr))
        o = np.lexsort((q_idx, rows))
        q_idx, q_val = q_idx[o], q_val[o]
        if ((rows[o][1:] == rows[o][:-1]) & (q_idx[1:] == q_idx[:-1])).any():
            raise ValueError("duplicate sparse index in query")
        params = {k: int(getattr(hp, k)) for k in ("matryoshka_64_limit", "matryoshka_128_limit", "matryoshka_256_limit",
                                                   "dense_limit", "quantized_limit", "sparse_limit", "final_limit", "hnsw_ef")}
        self.h._command("search", self.user)
        keys, cnt = self._together(lambda: self.col.search(
            q, q_indptr, q_idx.astype(np.int32), q_val, params, "h1" if int(hp.mode) == 1 else "tree",
            float(hp.rrf_k), int(hp.rrf_rank_base), int(hp.rrf_limit)))
        return ShardedCollection.unpack(keys, cnt)

    def count(self) -> int:
        return self.col.count()

    def save(self, path: str):
        self.h._command("save", self.user, path)
        self._together(lambda: self.h._save_shard(self.col, path))

    def close(self):
        self.col.close()


class ShardedHandler(QdrantHandler):
    """`QdrantHandler` over a row-sharded collection per user: the SAME class on the front rank -- payload building
    (the reference's 19 fields, qdrant_handler.py:165-185), `store_chat_vectors` (:200-267), `filters` on the root query
    (:297, :371), the rerank hook (:380), `persist_dir`, the error conventions -- with the engine index replaced by a
    sharded one.  The front rank holds point ids and payloads; worker ranks hold only their shard and run `serve()`.
    `timeout` (seconds) bounds every wait on another rank: a worker that died or hangs makes the front rank's call
    fail (search -> [], mutation -> raise) instead of hanging it."""

    def __init__(self, group=None, index_factory=None, ops=None, src: int = 0, dense_vector_size: int = 768,
                 matryoshka_sizes: Sequence[int] = (64, 128, 256), reranker=None, persist_dir: Optional[str] = None,
                 timeout: float = 300.0, index_loader=None, device: Optional[torch.device] = None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = device if device is not None else _dev_of(group)
        self.shard_device = device
        super().__init__(reranker=reranker, device=dev.index or 0, persist_dir=persist_dir)
        self.factory, self.loader, self.ops, self.src = index_factory, index_loader, ops, src
        self.dim, self.msizes = int(dense_vector_size), tuple(matryoshka_sizes)
        self.ctl = _control_group(group, timeout)
        # over gloo (CPU tests, rehearsals) the data path runs on the control group too and is bounded by `timeout`;
        # over RCCL the data path keeps the caller's group, whose collectives the RCCL watchdog bounds
        # (init_process_group(timeout=...)): this class bounds what it owns
        self.group = self.ctl if (self.ctl is not None and dist.get_backend(group) == "gloo") else group
        self.broken: Optional[str] = None        # set when a rank stopped answering: the group is no longer usable
        self._shards: Dict[str, ShardedCollection] = {}      # every rank; the front rank's _collections wrap them

    # ---- the command channel: the front rank announces every call, every rank acknowledges ---------------------
    def _command(self, op: Optional[str], user_id: Optional[str] = None, meta: Any = None):
        """Front rank: announce (op, user, meta) and collect the acknowledgements.  Worker ranks (op None): receive
        the next command and acknowledge it.  A rank that cannot take part says why, and then NO rank enters the
        call's data path: returns (op, user, meta) or raises ShardError on every rank alike."""
        if self.broken:
            raise ShardError(self.broken)
        if self.world == 1:
            return op, user_id, meta
        if self.dev_is_cuda() or (self.shard_device is not None and self.shard_device.type == "cuda"):
            torch.cuda.set_device(self.device)       # executor threads start on device 0
        try:
            box = [op, user_id, meta] if self.rank == self.src else [None, None, None]
            dist.broadcast_object_list(box, self.src, group=self.ctl)
            op, user_id, meta = box
            why = None
            if op in ("store", "search", "save", "delete") and user_id not in self._shards:
                why = f"no collection for {user_id!r}"
            acks = [None] * self.world
            dist.all_gather_object(acks, why, group=self.ctl)
        except ShardError:
            raise
        except Exception as e:                    # timeout / lost peer: the ranks are out of step from here on
            self.broken = f"a rank did not answer ({type(e).__name__}: {e}); the sharded handler is closed"
            raise ShardError(self.broken) from e
        bad = [f"rank {j}: {a}" for j, a in enumerate(acks) if a is not None]
        if bad:
            raise ShardError(f"{op} refused: " + "; ".join(bad))
        return op, user_id, meta

    def dev_is_cuda(self) -> bool:
        return dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def serve(self):
        """Worker ranks: follow the front rank's calls until it calls shutdown().  A call that fails is logged and
        the loop goes on (the front rank has seen the failure through the outcome exchange)."""
        assert self.rank != self.src
        while True:
            try:
                op, user, meta = self._command(None)
            except ShardError as e:
                logging.error("sharded handler, rank %d: %s", self.rank, e)
                if self.broken:
                    return
                continue
            if op == "shutdown":
                return
            try:
                if op == "create":
                    self._open_shard(user, **meta)
                elif op == "store":
                    self._shards[user].store()
                elif op == "search":
                    self._shards[user].search()
                elif op == "save":
                    self._save_shard(self._shards[user], meta)
                elif op == "delete":
                    self._shards.pop(user).close()
            except ShardOutOfStep as e:       # a collective step itself failed: this rank leaves the loop too
                self.broken = f"{e}; the sharded handler is closed"
                logging.error("sharded handler, rank %d: %s", self.rank, self.broken)
                return
            except Exception as e:
                logging.error("sharded handler, rank %d: %s(%s) failed: %s", self.rank, op, user, e)

    def shutdown(self):
        self._command("shutdown")

    # ---- shards ----------------------------------------------------------------------------------------------------
    def _shard_path(self, path: str) -> str:
        return f"{path}.r{self.rank}of{self.world}"

    def _save_shard(self, col: ShardedCollection, path: str):
        err = None
        try:
            col.local.save(self._shard_path(path))
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        bad = [f"rank {j}: {o}" for j, o in enumerate(col._outcomes(err)) if o is not None]
        if bad:
            raise ShardError("save failed: " + "; ".join(bad))

    def _open_shard(self, user_id, dim, msizes, load=None, counts=None) -> ShardedCollection:
        """every rank: this rank's shard of a new (or stored: `load` = path, `counts` = rows per rank) collection"""
        old = self._shards.pop(user_id, None)
        if old is not None:
            old.close()
        local, err = None, None
        if load:
            try:
                path = self._shard_path(load)
                if self.loader is not None:
                    local = self.loader(path)
                else:
                    from . import engine as _engine
                    local = _engine.HxIndex.load(path, device=self.device)
                if local.count() != int(counts[self.rank]):
                    raise ValueError("stored shard holds another number of rows than the collection's record says")
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
            outs = [err]
            if self.world > 1:
                outs = [None] * self.world
                try:
                    dist.all_gather_object(outs, err, group=self.ctl)
                except Exception as e:        # timeout / lost peer inside the exchange: out of step, as in _command
                    self.broken = f"a rank did not answer ({type(e).__name__}: {e}); the sharded handler is closed"
                    raise ShardOutOfStep(self.broken) from e
            bad = [f"rank {j}: {o}" for j, o in enumerate(outs) if o is not None]
            if bad:
                if local is not None and hasattr(local, "close"):
                    local.close()
                raise ShardError("stored collection not loaded: " + "; ".join(bad))
        col = ShardedCollection(dim, msizes, self.group, self.factory, self.ops, self.src, ctl=self.ctl, local=local,
                                counts=counts if load else None, device=self.shard_device)
        self._shards[user_id] = col
        return col

    async def create_collection(self, user_id: str, dense_vector_size: Optional[int] = None,
                                matryoshka_sizes: Optional[list] = None, quantized_size: Optional[int] = None,
                                sparse_enabled: bool = True, force_recreate: bool = False):
        """qdrant_handler.py:24-32, with the handler's own vector sizes as the defaults"""
        d = self.dim if dense_vector_size is None else int(dense_vector_size)
        return await super().create_collection(
            user_id, d, list(self.msizes) if matryoshka_sizes is None else matryoshka_sizes,
            d if quantized_size is None else quantized_size, sparse_enabled, force_recreate)

    # ---- QdrantHandler's two hooks -------------------------------------------------------------------------------
    def _open_collection(self, user_id, dim, msizes, sparse_enabled, force_recreate) -> _Collection:
        base = self._base(user_id)
        meta = None
        if base and not force_recreate and os.path.exists(base + ".json"):
            with open(base + ".json") as f:
                meta = json.load(f)
            if int(meta.get("world", 1)) != self.world:
                raise ValueError(f"stored collection was sharded over {meta.get('world', 1)} ranks, this group has {self.world}")
            if int(meta["dim"]) != dim:
                raise ValueError("stored collection has another vector size")
            msizes = [int(m) for m in meta["msizes"]]
        args = dict(dim=dim, msizes=list(msizes), load=(base + ".hx") if meta else None,
                    counts=meta["shard_rows"] if meta else None)
        self._command("create", user_id, args)
        shard = self._open_shard(user_id, **args)
        col = _Collection(dim, msizes, self.device, index=_ShardedBackend(self, user_id, shard))
        col.sparse_enabled = bool(meta["sparse_enabled"]) if meta else sparse_enabled
        if meta:
            col.ids, col.payloads = list(meta["ids"]), list(meta["payloads"])
        return col

    def _drop(self, user_id, col: _Collection) -> None:
        self._command("delete", user_id)
        self._shards.pop(user_id).close()

    async def save_collection(self, user_id: str) -> None:
        """persist_dir: every rank writes its shard (<user>.hx.r<rank>of<world>), the front rank the record of point
        ids, payloads and rows per rank."""
        if not self.persist_dir:
            raise ValueError("handler was created without persist_dir")
        col = self._collections[str(user_id)]
        os.makedirs(self.persist_dir, exist_ok=True)
        base = self._base(user_id)

        def save():
            col.index.save(base + ".hx")
            with open(base + ".json", "w") as f:
                json.dump({"dim": col.dim, "msizes": list(col.msizes), "sparse_enabled": col.sparse_enabled,
                           "ids": col.ids, "payloads": col.payloads, "world": self.world,
                           "shard_rows": [int(c) for c in col.index.col.counts]}, f)
        await self._run(save)
