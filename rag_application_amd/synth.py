"""Synthetic workload of SURVEY.md §8(d): lookup tables and query batches (host side).

The corpus itself is generated on the device by `hx_synth_fill`; this module only
builds what the host hands over: the Zipf CDF / length tables and the sparse query
batch.  Everything is integer-exact so that the oracle (oracle/oracle.py, which keeps
its own independent copy of these rules) regenerates identical data."""
from __future__ import annotations

import numpy as np

SEED_CORPUS, SEED_QUERY, SEED_SPDOC, SEED_SPQUERY = 0x5EED0001, 0x5EED0002, 0x5EED0003, 0x5EED0004
V = 1 << 20
ZIPF_S = 1.07
STOP = 128
_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash2(seed, a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    h = _fmix32((np.uint64(seed) + a * np.uint64(0x9E3779B1)) & _M32)
    return _fmix32(h ^ ((b * np.uint64(0x85EBCA77)) & _M32))


def tables():
    """(cdf_u32[V], len_u16[256]): Zipf(1.07) CDF scaled to 2^32, and 256 quantiles of
    lognormal(ln 120, 0.4) document lengths clipped to [8, 512]."""
    from scipy.special import ndtri
    p = np.arange(1, V + 1, dtype=np.float64) ** (-ZIPF_S)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    cdf_u32 = np.minimum(np.floor(cdf * 4294967296.0), 4294967295.0).astype(np.uint32)
    cdf_u32[-1] = 0xFFFFFFFF
    qs = (np.arange(256, dtype=np.float64) + 0.5) / 256.0
    ln = np.exp(np.log(120.0) + 0.4 * ndtri(qs))
    return cdf_u32, np.clip(np.rint(ln), 8, 512).astype(np.uint16)


def bm25_weight(tf, doc_len, k=1.2, b=0.75, avg_len=256.0):
    tf = np.asarray(tf, dtype=np.float64)
    doc_len = np.asarray(doc_len, dtype=np.float64)
    return (tf * (k + 1.0) / (tf + k * (1.0 - b + b * doc_len / avg_len))).astype(np.float32)


def sparse_queries(seed: int, q0: int, n: int, tabs=None):
    """CSR batch (indptr int64, idx int32 ascending per query, val f32): query q has
    T = 3 + hash % 10 tokens drawn from the Zipf CDF above rank 128, duplicates merged
    into tf, weighted with the document formula (the reference calls .embed() for
    queries too, app/core/embedding/embedding_handler.py:123)."""
    cdf, _ = tabs or tables()
    c0 = np.uint64(cdf[STOP - 1])
    indptr = [0]
    idx, val = [], []
    for q in range(q0, q0 + n):
        T = 3 + int(hash2(seed, q, 0xFFFFFFFF)) % 10
        h = hash2(seed, np.uint64(q), np.arange(T, dtype=np.uint64))
        u = c0 + ((h * (np.uint64(1 << 32) - c0)) >> np.uint64(32))
        rank = np.minimum(np.searchsorted(cdf, u.astype(np.uint32), side="right"), V - 1)
        r, tf = np.unique(rank, return_counts=True)
        t = ((r.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0x7FFFFFFF)).astype(np.int64)
        o = np.argsort(t, kind="stable")
        idx.append(t[o].astype(np.int32))
        val.append(bm25_weight(tf, T)[o])
        indptr.append(indptr[-1] + len(r))
    return (np.asarray(indptr, dtype=np.int64), np.concatenate(idx), np.concatenate(val))


def ingest_batch(seed: int, n: int, dim: int, tabs=None):
    """A host-side ingest batch of the bench's shape for timing `store_document_vectors`' hand-off (bench.py, config
    5): n fp32 rows uniform in [-1, 1) and a sparse CSR drawn like the corpus (document lengths from the lognormal
    table, tokens from the Zipf CDF, equal tokens merged into tf, BM25 tf weights, term id = rank * 0x9E3779B1 mod
    2^31) -- the same distributions as `hx_synth_fill`, vectorised, NOT the same random stream (nothing is compared
    against it).  Returns (dense [n x dim] f32, indptr int64, idx int32, val f32)."""
    cdf, lens = tabs or tables()
    rng = np.random.default_rng(seed)
    dense = rng.random((n, dim), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    L = lens[rng.integers(0, 256, n)].astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), L)
    u = rng.integers(0, 1 << 32, rows.size, dtype=np.uint64).astype(np.uint32)
    rank = np.minimum(np.searchsorted(cdf, u, side="right"), V - 1).astype(np.int64)
    key, tf = np.unique((rows << np.int64(20)) | rank, return_counts=True)      # sorted by row, then rank
    r = key >> np.int64(20)
    idx = (((key & np.int64((1 << 20) - 1)).astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0x7FFFFFFF)).astype(np.int32)
    val = bm25_weight(tf, L[r])
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=indptr[1:])
    return dense, indptr, idx, val
