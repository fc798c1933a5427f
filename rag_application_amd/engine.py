"""Thin Python face of libhx: one `HxIndex` = one user collection on one GPU.

torch is plumbing only (device buffers, the current stream); every computation is a
HIP kernel behind the C ABI of include/hx.h.  Ranked lists travel between stages as
int64 tensors holding the engine's 64-bit keys (see hx.h)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import HX_MODE_H1, HX_MODE_TREE, HxError, HxParams, HxProf, HxStats, check

SEARCH_PARAM_KEYS = ("matryoshka_64_limit", "matryoshka_128_limit", "matryoshka_256_limit",
                     "dense_limit", "quantized_limit", "sparse_limit", "final_limit", "hnsw_ef")


def make_params(search_params: dict, mode: int = HX_MODE_TREE, rrf_k: float = 2.0,
                rrf_rank_base: int = 0, rrf_limit: int = 10) -> HxParams:
    """search_params dict (hybrid_search_workflow.py:8-19) -> hx_params.  Indexes the
    dict unconditionally like the reference (qdrant_handler.py:314-369): a missing
    key raises KeyError, None raises TypeError."""
    p = HxParams()
    for k in SEARCH_PARAM_KEYS:
        setattr(p, k, int(search_params[k]))
    p.rrf_k = float(rrf_k)
    p.rrf_rank_base = int(rrf_rank_base)
    p.rrf_limit = int(rrf_limit)
    p.mode = int(mode)
    return p


def _ptr(t) -> int:
    if t is None:
        return 0
    if isinstance(t, torch.Tensor):
        return t.data_ptr()
    return t.ctypes.data


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype):
        raise HxError(f"{name} must be a cuda tensor of dtype {dtype}")
    return t.contiguous()


class HxIndex:
    def __init__(self, dim: int = 768, matryoshka_sizes: Sequence[int] = (64, 128, 256),
                 device: int = 0, id_base: int = 0):
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.msizes = tuple(int(m) for m in matryoshka_sizes)
        self.device = int(device)
        self.id_base = int(id_base)
        ms = (C.c_int32 * max(len(self.msizes), 1))(*self.msizes)
        check(_lib.lib().hx_create(self.dim, ms, len(self.msizes), self.device, self.id_base,
                                   C.byref(self._h)))
        self._tdev = torch.device("cuda", self.device)

    # -- persistence (hx.h: hx_save / hx_load) -----------------------------------------
    def save(self, path: str) -> None:
        check(_lib.lib().hx_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path: str, device: int = 0) -> "HxIndex":
        """A new index from a file written by save(): same rows, same search results bit for bit."""
        import struct
        with open(path, "rb") as f:
            head = f.read(8 + 6 * 4 + 4 * 8)
        magic, dim, n_pre, p0, p1, p2, _r, id_base, _n, _sr, _nnz = struct.unpack("<8s6i4q", head)
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        check(_lib.lib().hx_load(str(path).encode(), int(device), C.byref(self._h)))
        self.dim = int(dim)
        self.msizes = tuple(int(m) for m in (p0, p1, p2)[:n_pre])
        self.device = int(device)
        self.id_base = int(id_base)
        self._tdev = torch.device("cuda", self.device)
        return self

    # -- lifecycle -------------------------------------------------------------------
    def close(self):
        if self._h:
            check(_lib.lib().hx_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ingest ----------------------------------------------------------------------
    def reserve(self, n_rows: int, nnz: int = 0):
        check(_lib.lib().hx_reserve(self._h, n_rows, nnz))

    def add(self, dense: np.ndarray, sp_indptr=None, sp_idx=None, sp_val=None):
        dense = np.ascontiguousarray(dense, dtype=np.float32)
        if dense.ndim != 2 or dense.shape[1] != self.dim:
            raise ValueError(f"Dense vector dimension mismatch. Expected {self.dim}, got {dense.shape[-1]}")
        n = dense.shape[0]
        if sp_indptr is None:
            sp_indptr = np.zeros(n + 1, dtype=np.int64)
            sp_idx = np.zeros(0, dtype=np.int32)
            sp_val = np.zeros(0, dtype=np.float32)
        sp_indptr = np.ascontiguousarray(sp_indptr, dtype=np.int64)
        sp_idx = np.ascontiguousarray(sp_idx, dtype=np.int32)
        sp_val = np.ascontiguousarray(sp_val, dtype=np.float32)
        if sp_indptr.shape[0] != n + 1:
            raise ValueError("sparse indptr must have n+1 entries")
        # dense and sparse vectors of a batch are committed together or not at all (hx_add_rows)
        check(_lib.lib().hx_add_rows(self._h, _ptr(dense), _ptr(sp_indptr), _ptr(sp_idx), _ptr(sp_val), n))

    def add_device(self, dense: torch.Tensor, sp_indptr=None, sp_idx=None, sp_val=None):
        """`add` for dense rows that already live on the GPU (an encoder's output): hx_add_dense_dev."""
        dense = _need_cuda(dense, torch.float32, "dense")
        if dense.ndim != 2 or dense.shape[1] != self.dim:
            raise ValueError(f"Dense vector dimension mismatch. Expected {self.dim}, got {dense.shape[-1]}")
        n = dense.shape[0]
        if sp_indptr is None:
            sp_indptr = np.zeros(n + 1, dtype=np.int64)
            sp_idx = np.zeros(0, dtype=np.int32)
            sp_val = np.zeros(0, dtype=np.float32)
        sp_indptr = np.ascontiguousarray(sp_indptr, dtype=np.int64)
        sp_idx = np.ascontiguousarray(sp_idx, dtype=np.int32)
        sp_val = np.ascontiguousarray(sp_val, dtype=np.float32)
        if sp_indptr.shape[0] != n + 1:
            raise ValueError("sparse indptr must have n+1 entries")
        # committed together or not at all, like `add` (hx_add_rows_dev)
        check(_lib.lib().hx_add_rows_dev(self._h, _ptr(dense), _ptr(sp_indptr), _ptr(sp_idx), _ptr(sp_val), n, _stream()))

    def set_next_id(self, first_id: int):
        """Row sharding: the next add's rows get the global ids first_id, first_id + 1, ... (hx_set_next_id)."""
        check(_lib.lib().hx_set_next_id(self._h, int(first_id)))

    def truncate(self, n_rows: int):
        """Roll back to the first n_rows rows (hx_truncate)."""
        check(_lib.lib().hx_truncate(self._h, int(n_rows)))

    def synth_fill(self, n: int, seed_dense: int, seed_sparse: int = 0, tables=None):
        if tables is not None:
            cdf = np.ascontiguousarray(tables[0], dtype=np.uint32)
            lens = np.ascontiguousarray(tables[1], dtype=np.uint16)
            check(_lib.lib().hx_synth_fill(self._h, n, seed_dense, seed_sparse, _ptr(cdf), cdf.shape[0],
                                           _ptr(lens), 1))
        else:
            check(_lib.lib().hx_synth_fill(self._h, n, seed_dense, 0, 0, 0, 0, 0))

    def finalize(self):
        check(_lib.lib().hx_finalize(self._h))

    def rebuild_sparse(self):
        """Build the inverted index again (hx_rebuild_sparse): a measurement aid."""
        check(_lib.lib().hx_rebuild_sparse(self._h))

    def count(self) -> int:
        n = C.c_int64()
        check(_lib.lib().hx_count(self._h, C.byref(n)))
        return n.value

    def stats(self) -> dict:
        s = HxStats()
        check(_lib.lib().hx_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in HxStats._fields_}

    def set_dense_candidates(self, kind: str):
        """'i8' (default) or 'f16': which copy nominates the dense stage's candidates (hx_set_dense_candidates)."""
        check(_lib.lib().hx_set_dense_candidates(self._h, {"f16": 0, "i8": 1}[kind]))

    def set_stream_overlap(self, on: bool):
        """False: every stage of a hybrid call on the caller's stream (a kernel's profiled duration is then its own)."""
        check(_lib.lib().hx_set_stream_overlap(self._h, 1 if on else 0))

    def profile(self, enable: bool):
        check(_lib.lib().hx_profile(self._h, 1 if enable else 0))

    def profile_read(self) -> dict:
        p = HxProf()
        check(_lib.lib().hx_profile_read(self._h, C.byref(p)))
        names = ("scan_f16", "scan_i8", "sparse", "scan_cand8", "prep_rows")
        return {n: dict(launches=p.launches[i], ms=p.ms[i], flops=p.flops[i], bytes=p.bytes[i])
                for i, n in enumerate(names)}

    def debug_row(self, which: int, row: int) -> np.ndarray:
        if which in (4, 5):
            out = np.zeros(self.dim, dtype=np.int8)
        elif which == 6:
            out = np.zeros(1, dtype=np.float32)
        elif which == 0:
            out = np.zeros(self.dim, dtype=np.float32)
        else:
            out = np.zeros(self.msizes[which - 1], dtype=np.float32)
        check(_lib.lib().hx_debug_row(self._h, which, row, _ptr(out)))
        return out

    # -- stages (device tensors) -----------------------------------------------------
    def _out(self, B: int, L: int):
        return (torch.empty((B, L), dtype=torch.int64, device=self._tdev),
                torch.empty((B,), dtype=torch.int32, device=self._tdev))

    # `flag` (a zeroed int32 device tensor of one element): the stage is only ENQUEUED -- no host round trip -- and
    # adds to flag[0] the number of queries whose lists are not final (hx_*_async); the caller reads the word once
    # behind all the stages of its query and redoes the batch without `flag` when it is not zero.
    deferred_stages = True

    def search_dense(self, q: torch.Tensor, limit: int, prefix: int = 0, flag: Optional[torch.Tensor] = None):
        q = _need_cuda(q, torch.float32, "q")
        keys, cnt = self._out(q.shape[0], limit)
        if flag is not None:
            check(_lib.lib().hx_search_dense_async(self._h, _ptr(q), q.shape[0], prefix, limit, _ptr(keys), _ptr(cnt),
                                                   _ptr(_need_cuda(flag, torch.int32, "flag")), _stream()))
        else:
            check(_lib.lib().hx_search_dense(self._h, _ptr(q), q.shape[0], prefix, limit, _ptr(keys), _ptr(cnt),
                                             _stream()))
        return keys, cnt

    def search_i8(self, q: torch.Tensor, limit: int, flag: Optional[torch.Tensor] = None):
        q = _need_cuda(q, torch.float32, "q")
        keys, cnt = self._out(q.shape[0], limit)
        if flag is not None:
            check(_lib.lib().hx_search_i8_async(self._h, _ptr(q), q.shape[0], limit, _ptr(keys), _ptr(cnt),
                                                _ptr(_need_cuda(flag, torch.int32, "flag")), _stream()))
        else:
            check(_lib.lib().hx_search_i8(self._h, _ptr(q), q.shape[0], limit, _ptr(keys), _ptr(cnt), _stream()))
        return keys, cnt

    def search_sparse(self, q_indptr: torch.Tensor, q_idx: torch.Tensor, q_val: torch.Tensor, limit: int,
                      flag: Optional[torch.Tensor] = None):
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        B = q_indptr.shape[0] - 1
        keys, cnt = self._out(B, limit)
        if flag is not None:
            check(_lib.lib().hx_search_sparse_async(self._h, _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, limit,
                                                    _ptr(keys), _ptr(cnt), _ptr(_need_cuda(flag, torch.int32, "flag")),
                                                    _stream()))
        else:
            check(_lib.lib().hx_search_sparse(self._h, _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, limit,
                                              _ptr(keys), _ptr(cnt), _stream()))
        return keys, cnt

    def h1_local(self, q: torch.Tensor, q_indptr: torch.Tensor, q_idx: torch.Tensor, q_val: torch.Tensor,
                 dense_limit: int, sparse_limit: int) -> torch.Tensor:
        """This shard's dense and sparse lists side by side, [B, dense_limit + sparse_limit] (hx_h1_local)."""
        q = _need_cuda(q, torch.float32, "q")
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        B = q.shape[0]
        keys = torch.empty((B, dense_limit + sparse_limit), dtype=torch.int64, device=q.device)
        check(_lib.lib().hx_h1_local(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, dense_limit,
                                     sparse_limit, _ptr(keys), _stream()))
        return keys

    def h1_local_async(self, q: torch.Tensor, q_indptr: torch.Tensor, q_idx: torch.Tensor, q_val: torch.Tensor,
                       dense_limit: int, sparse_limit: int) -> torch.Tensor:
        """h1_local without the host round trip (hx_h1_local_async): [B + 1, dense_limit + sparse_limit]; row B,
        element 0 = queries whose lists are not final (then the batch must be redone through h1_local)."""
        q = _need_cuda(q, torch.float32, "q")
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        B = q.shape[0]
        keys = torch.empty((B + 1, dense_limit + sparse_limit), dtype=torch.int64, device=q.device)
        check(_lib.lib().hx_h1_local_async(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, dense_limit,
                                           sparse_limit, _ptr(keys), _stream()))
        return keys

    # -- candidates-first sharded H1 (hx.h: hx_h1_nominate_async / hx_h1_rescore_async; distributed.H1Pipeline) ----------
    def sparse_wmax(self):
        """(largest document weight of this shard, whether it holds a non-positive weight)"""
        w, npos = C.c_float(), C.c_int32()
        check(_lib.lib().hx_sparse_wmax(self._h, C.byref(w), C.byref(npos)))
        return float(w.value), bool(npos.value)

    def set_sparse_wmax(self, wmax: float):
        """The largest document weight of ANY shard: one scale for the integer BM25 scores of all shards."""
        check(_lib.lib().hx_set_sparse_wmax(self._h, float(wmax)))

    def h1_nominate_async(self, q, q_indptr, q_idx, q_val, dense_limit: int, sparse_limit: int, k1: int, k2: int,
                          lout: Optional[int] = None):
        """This shard's nominations (hx_h1_nominate_async): flat int64, the first B * (k1 + k2 + 2) words are what the
        shards exchange, the B * (lout + 1) words behind them stay with the batch on this rank (h1_rescore_async)."""
        q = _need_cuda(q, torch.float32, "q")
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        B = q.shape[0]
        if lout is None:
            lout = h1_plan(dense_limit, sparse_limit, 1)[4]
        nom = torch.empty((B * (k1 + k2 + 2) + B * (lout + 1),), dtype=torch.int64, device=q.device)
        check(_lib.lib().hx_h1_nominate_async(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, dense_limit,
                                              sparse_limit, k1, k2, _ptr(nom), _stream()))
        return nom

    def h1_rescore_async(self, q, q_indptr, q_idx, q_val, nom: torch.Tensor, gathered: torch.Tensor, world: int, rank: int,
                         dense_limit: int, sparse_limit: int, k1: int, k2: int, lp: int, k3: int):
        """Exact scores of this shard's rows among the global candidates, flat [B * (lp + world * k3 + world + 4)]
        (hx_h1_rescore_async).  `nom`: this rank's own h1_nominate_async result; `gathered`: every rank's public part."""
        q = _need_cuda(q, torch.float32, "q")
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        gathered = _need_cuda(gathered, torch.int64, "gathered")
        nom = _need_cuda(nom, torch.int64, "nom")
        B = q.shape[0]
        if gathered.numel() != world * B * (k1 + k2 + 2):
            raise HxError("gathered nominations have the wrong size")
        res = torch.empty((B * (lp + world * k3 + world + 4),), dtype=torch.int64, device=q.device)
        check(_lib.lib().hx_h1_rescore_async(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B, _ptr(nom),
                                             _ptr(gathered), world, rank, dense_limit, sparse_limit, k1, k2, lp, k3,
                                             _ptr(res), _stream()))
        return res

    def rescore(self, q: torch.Tensor, cand_keys: torch.Tensor, cand_counts: Optional[torch.Tensor],
                limit: int, prefix: int = 0):
        q = _need_cuda(q, torch.float32, "q")
        cand_keys = _need_cuda(cand_keys, torch.int64, "cand_keys")
        if cand_counts is not None:
            cand_counts = _need_cuda(cand_counts, torch.int32, "cand_counts")
        B = q.shape[0]
        keys, cnt = self._out(B, limit)
        check(_lib.lib().hx_rescore(self._h, _ptr(q), B, prefix, _ptr(cand_keys), cand_keys.shape[1],
                                    _ptr(cand_counts), limit, _ptr(keys), _ptr(cnt), _stream()))
        return keys, cnt

    def hybrid_query(self, q: torch.Tensor, q_indptr: torch.Tensor, q_idx: torch.Tensor,
                     q_val: torch.Tensor, params: HxParams):
        q = _need_cuda(q, torch.float32, "q")
        q_indptr = _need_cuda(q_indptr, torch.int64, "q_indptr")
        q_idx = _need_cuda(q_idx, torch.int32, "q_idx")
        q_val = _need_cuda(q_val, torch.float32, "q_val")
        B = q.shape[0]
        keys, cnt = self._out(B, params.final_limit)
        check(_lib.lib().hx_hybrid_query_dev(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B,
                                             C.byref(params), _ptr(keys), _ptr(cnt), _stream()))
        return keys, cnt

    def hybrid_query_host(self, q: np.ndarray, q_indptr: np.ndarray, q_idx: np.ndarray, q_val: np.ndarray,
                          params: HxParams):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.dim)
        q_indptr = np.ascontiguousarray(q_indptr, dtype=np.int64)
        q_idx = np.ascontiguousarray(q_idx, dtype=np.int32)
        q_val = np.ascontiguousarray(q_val, dtype=np.float32)
        B, L = q.shape[0], params.final_limit
        scores = np.empty((B, L), dtype=np.float32)
        ids = np.empty((B, L), dtype=np.int64)
        counts = np.empty((B,), dtype=np.int32)
        check(_lib.lib().hx_hybrid_query_host(self._h, _ptr(q), _ptr(q_indptr), _ptr(q_idx), _ptr(q_val), B,
                                              C.byref(params), _ptr(scores), _ptr(ids), _ptr(counts)))
        return scores, ids, counts


# -- index-free stages -------------------------------------------------------------------
def rrf(a_keys: torch.Tensor, a_cnt: torch.Tensor, b_keys: torch.Tensor, b_cnt: torch.Tensor,
        limit: int = 10, k: float = 2.0, rank_base: int = 0):
    dev = a_keys.device
    B = a_keys.shape[0]
    keys = torch.empty((B, limit), dtype=torch.int64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    check(_lib.lib().hx_rrf(dev.index or 0, _ptr(a_keys.contiguous()), a_keys.shape[1], _ptr(a_cnt),
                            _ptr(b_keys.contiguous()), b_keys.shape[1], _ptr(b_cnt), B, k, rank_base, limit,
                            _ptr(keys), _ptr(cnt), _stream()))
    return keys, cnt


def h1_fuse(gathered: torch.Tensor, world: int, dense_limit: int, sparse_limit: int, limit: int = 10,
            k: float = 2.0, rank_base: int = 0):
    """gathered: [world * B, dense_limit + sparse_limit], rank-major (all_gather_into_tensor of h1_local):
    global dense and sparse lists, then RRF (hx_h1_fuse)."""
    dev = gathered.device
    gathered = gathered.contiguous()
    B = gathered.shape[0] // world
    keys = torch.empty((B, limit), dtype=torch.int64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    check(_lib.lib().hx_h1_fuse(dev.index or 0, _ptr(gathered), world, B, dense_limit, sparse_limit, limit, k,
                                rank_base, _ptr(keys), _ptr(cnt), _stream()))
    return keys, cnt


def h1_plan(dense_limit: int, sparse_limit: int, world: int):
    """(k1, k2, lp, k3, lout) of the candidates-first exchange for `world` shards (hx_h1_plan)."""
    v = [C.c_int32() for _ in range(5)]
    check(_lib.lib().hx_h1_plan(dense_limit, sparse_limit, world, *[C.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)


def h1_finish(reduced: torch.Tensor, world: int, B: int, lp: int, k3: int, dense_limit: int, sparse_limit: int,
              limit: int = 10, k: float = 2.0, rank_base: int = 0, nfail: Optional[torch.Tensor] = None):
    """reduced: the all-reduced (integer sum over the `world` ranks) result of h1_rescore_async.  Returns (keys [B, limit],
    counts [B], nfail [1]): nfail = queries whose lists are not final (hx_h1_finish adds to it)."""
    dev = reduced.device
    reduced = reduced.contiguous()
    keys = torch.empty((B, limit), dtype=torch.int64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    if nfail is None:
        nfail = torch.zeros((1,), dtype=torch.int32, device=dev)
    check(_lib.lib().hx_h1_finish(dev.index or 0, _ptr(reduced), world, B, lp, k3, dense_limit, sparse_limit, limit, k,
                                  rank_base, _ptr(keys), _ptr(cnt), _ptr(nfail), _stream()))
    return keys, cnt, nfail


def merge(keys_in: torch.Tensor, counts_in: Optional[torch.Tensor], limit: int, dedupe: bool = False):
    dev = keys_in.device
    keys_in = keys_in.contiguous()
    B = keys_in.shape[0]
    keys = torch.empty((B, limit), dtype=torch.int64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    check(_lib.lib().hx_merge(dev.index or 0, _ptr(keys_in), keys_in.shape[1], _ptr(counts_in), B, limit,
                              1 if dedupe else 0, _ptr(keys), _ptr(cnt), _stream()))
    return keys, cnt


def unpack(keys: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    keys = keys.contiguous()
    scores = torch.empty(keys.shape, dtype=torch.float32, device=keys.device)
    ids = torch.empty(keys.shape, dtype=torch.int64, device=keys.device)
    check(_lib.lib().hx_unpack(keys.device.index or 0, _ptr(keys), keys.numel(), _ptr(scores), _ptr(ids),
                               _stream()))
    return scores, ids


def synth_queries_dense(dim: int, q0: int, B: int, seed: int, device: int = 0) -> torch.Tensor:
    q = torch.empty((B, dim), dtype=torch.float32, device=torch.device("cuda", device))
    check(_lib.lib().hx_synth_queries_dense(dim, q0, B, seed, _ptr(q), _stream()))
    return q
