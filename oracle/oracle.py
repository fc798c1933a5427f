"""CPU oracle (numpy) for the hybrid-retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``rag_application_amd``) never does.

PARITY STATUS: **parity unpinned**.  The reference (VivekMalipatel/RAG_Application)
does not implement this arithmetic itself: it describes the query as a tree of
``Prefetch`` objects (app/core/vector_store/qdrant/qdrant_handler.py:296-372) and
ships it to an un-pinned ``qdrant/qdrant:latest`` server (docker-compose.yml:23-24);
BM25 weights come from un-pinned ``fastembed`` (app/core/embedding/
embedding_handler.py:41,123).  Neither package exists in the build container and
the reference's tests hold no golden vectors for this path.  What
IS pinned to the reference: the int8 quantisation expression (qdrant_handler.py:
144-146, 300-302 -- plain numpy, evaluated literally below), the stage order /
limits / ``using`` names of the query tree (qdrant_handler.py:305-372), the
schema (qdrant_handler.py:58-86) and the 8-key ``search_params`` contract
(app/services/agents/hybrid_search_workflow.py:8-19).  Every semantic inherited
from Qdrant / fastembed is a NAMED SWITCH below whose default is "assumed
upstream behaviour, unverifiable offline".

Arithmetic contract (shared bit-for-bit with oracle/hx_oracle.c and the HIP
engine; all fp32 operations are IEEE round-to-nearest, UNFUSED mul then add):

sparse score        : SPARSE_FIX_BITS = None (default, upstream order): the query's terms in
                      ascending term id, acc = acc + q_t*d_t from +0 (fp32 mul, fp32 add) over the
                      terms the document holds.  SPARSE_FIX_BITS = 40 (round-1 engine arithmetic,
                      kept as a switch): sum of rint(f64(q_t)*f64(d_t)*2^40) as int64, then
                      f32(sum) * 2^-40 -- order-independent, a few fp32 ulps from the former;
                      tests/test_oracle.py measures how many lists differ between the two.
``spec_dot(x, q)``  : zero-pad to a multiple of 64; lane l (0..63) accumulates
                      p_l = p_l + x[64j+l]*q[64j+l] for j ascending from +0;
                      then p_l += p_{l+off} for off = 32,16,8,4,2,1 (l < off);
                      result p_0 + 0.0f.
total order         : (score descending, id ascending) for every ranking.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------
# Named switches -- "assumed upstream behaviour, unverifiable offline"
# ----------------------------------------------------------------------------
RRF_K = 2.0                    # Qdrant: 1/(rank + 2.0), 0-based rank        (a-9)
RRF_RANK_BASE = 0
PREFETCH_DEFAULT_LIMIT = 10    # Qdrant prefetch without ``limit``           (a-9)
NORM_SKIP_IF_UNIT = True       # Qdrant cosine_preprocess: keep the vector if
                               # |len^2 - 1| <= 1e-6 or len^2 < FLT_EPSILON   (a-2)
BM25_K = 1.2                   # fastembed Qdrant/bm25 defaults              (a-4)
BM25_B = 0.75
BM25_AVG_LEN = 256.0
SPARSE_IDF = False             # collection has no sparse modifier           (a-1/a-6)
SPARSE_FIX_BITS = None         # sparse scores: None = fp32 running sum in ascending term id (upstream's
                               # order, the engine's arithmetic); 40 = the order-independent 2^40
                               # fixed-point sum of round 1 (see sparse_scores)

F32 = np.float32
FLT_EPSILON = F32(1.1920929e-07)


# ----------------------------------------------------------------------------
# spec arithmetic
# ----------------------------------------------------------------------------
def _pad64(a: np.ndarray) -> np.ndarray:
    d = a.shape[-1]
    dp = (d + 63) // 64 * 64
    if dp == d:
        return a
    out = np.zeros(a.shape[:-1] + (dp,), dtype=a.dtype)
    out[..., :d] = a
    return out


def spec_dot(X: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Row-wise spec dot product.  X [n, D] f32, q [D] or [n, D] f32 -> [n] f32."""
    X = _pad64(np.ascontiguousarray(X, dtype=F32))
    q = _pad64(np.ascontiguousarray(q, dtype=F32))
    n, dp = X.shape
    Xr = X.reshape(n, dp // 64, 64)
    qr = q.reshape((-1, dp // 64, 64))
    p = np.zeros((n, 64), dtype=F32)
    for j in range(dp // 64):
        p = p + Xr[:, j, :] * qr[:, j, :]          # fp32 mul, fp32 add (unfused)
    for off in (32, 16, 8, 4, 2, 1):
        p = p[:, :off] + p[:, off:2 * off]
    return (p[:, 0] + F32(0.0)).astype(F32)


def spec_dot_matrix(X: np.ndarray, Q: np.ndarray) -> np.ndarray:
    """All-pairs spec dot: X [n, D], Q [B, D] -> S [B, n] f32 (row b = query b)."""
    return np.stack([spec_dot(X, Q[b]) for b in range(Q.shape[0])], axis=0)


def cosine_preprocess(X: np.ndarray) -> np.ndarray:
    """Per-vector L2 normalisation applied by a COSINE collection at upsert and to
    the query at search time (qdrant_handler.py:59-77; Qdrant ``cosine_preprocess``,
    recalled).  len2 = spec_dot(x, x); x / sqrtf(len2) with IEEE sqrt and divide."""
    X = np.ascontiguousarray(X, dtype=F32)
    if X.ndim == 1:
        return cosine_preprocess(X[None, :])[0]
    len2 = spec_dot(X, X)
    keep = len2 < FLT_EPSILON
    if NORM_SKIP_IF_UNIT:
        keep = keep | (np.abs(len2 - F32(1.0)) <= F32(1.0e-6))
    ln = np.sqrt(len2, dtype=F32)
    ln = np.where(keep, F32(1.0), ln).astype(F32)
    out = (X / ln[:, None]).astype(F32)
    out[keep] = X[keep]
    return out


def quantize_i8(X: np.ndarray) -> np.ndarray:
    """The reference's int8 copy, evaluated literally
    (qdrant_handler.py:144-146 for documents, :300-302 for the query):
    ``np.clip((np.array(x) * 127).astype(np.int8), -128, 127)``.
    The ABI hands the engine float32 data, so the float64 product x*127 is exact;
    the cast truncates toward zero and wraps through int32 (x86 numpy behaviour,
    pinned by tests/golden/i8_kat.json); |x*127| >= 2^31, NaN and Inf give 0."""
    with np.errstate(invalid="ignore", over="ignore"):
        x64 = np.asarray(X, dtype=F32).astype(np.float64)
        return np.clip((x64 * 127).astype(np.int8), -128, 127)


def i8_norm_inv(Xi8: np.ndarray) -> np.ndarray:
    """1/||x|| of an int8-valued vector (the "quantized" named vector is a COSINE
    vector too, qdrant_handler.py:64-68): exact integer len2, then
    f32(1/sqrt(f64(len2))); 0 for the zero vector."""
    n2 = (Xi8.astype(np.int64) ** 2).sum(axis=-1)
    with np.errstate(divide="ignore"):
        r = np.where(n2 > 0, 1.0 / np.sqrt(n2.astype(np.float64)), 0.0)
    return r.astype(F32)


def i8_scores(Xi8: np.ndarray, qi8: np.ndarray) -> np.ndarray:
    """Cosine between int8-valued vectors: exact int32 dot, then
    (f32(dot) * rinv_x) * rinv_q in fp32."""
    dot = Xi8.astype(np.int32) @ qi8.astype(np.int32)
    rx = i8_norm_inv(Xi8)
    rq = i8_norm_inv(qi8[None, :])[0]
    return ((dot.astype(F32) * rx) * rq).astype(F32)


def order_key(scores: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """64-bit key whose DESCENDING order is (score desc, id asc)."""
    u = np.ascontiguousarray(scores, dtype=F32).view(np.uint32).astype(np.uint64)
    u = np.where(u & np.uint64(0x80000000), (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (u << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - ids.astype(np.uint64))


def topk(scores: np.ndarray, ids: np.ndarray, limit: int):
    """Top ``limit`` of (score, id) pairs under the total order; returns (scores, ids)."""
    ids = np.asarray(ids, dtype=np.int64)
    scores = np.asarray(scores, dtype=F32)
    key = order_key(scores, ids)
    order = np.argsort(key, kind="stable")[::-1][: max(int(limit), 0)]
    return scores[order], ids[order]


# ----------------------------------------------------------------------------
# BM25 (fastembed Qdrant/bm25, recalled) -- sparse data contract (a-4)
# ----------------------------------------------------------------------------
def bm25_weight(tf, doc_len, k=None, b=None, avg_len=None):
    """value = tf*(k+1) / (tf + k*(1 - b + b*len/avg_len)), computed in float64
    and rounded to float32 (embedding_handler.py:123 -> fastembed Bm25, recalled)."""
    k = BM25_K if k is None else k
    b = BM25_B if b is None else b
    avg_len = BM25_AVG_LEN if avg_len is None else avg_len
    tf = np.asarray(tf, dtype=np.float64)
    doc_len = np.asarray(doc_len, dtype=np.float64)
    return (tf * (k + 1.0) / (tf + k * (1.0 - b + b * doc_len / avg_len))).astype(F32)


def murmur3_x86_32(data: bytes, seed: int = 0) -> int:
    """Unsigned murmur3_x86_32 (published algorithm; fastembed hashes stemmed tokens
    with mmh3.hash(token) and takes abs() of the signed result, recalled)."""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed & 0xFFFFFFFF
    n = len(data)
    for i in range(0, n - n % 4, 4):
        k = int.from_bytes(data[i:i + 4], "little")
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    tail = data[n - n % 4:]
    k = 0
    for i, byte in enumerate(tail):
        k |= byte << (8 * i)
    if tail:
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
    h ^= n
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def bm25_term_id(token: str) -> int:
    """abs(int32(murmur3_x86_32(token utf-8, seed 0)))."""
    h = murmur3_x86_32(token.encode("utf-8"), 0)
    if h >= 0x80000000:
        h -= 1 << 32
    return abs(h)


# ----------------------------------------------------------------------------
# Index (the six named vectors of qdrant_handler.py:58-86, one row per point)
# ----------------------------------------------------------------------------
class OracleIndex:
    """Exhaustive-search restatement of one user collection."""

    def __init__(self, dim=768, matryoshka_sizes=(64, 128, 256)):
        self.dim = int(dim)
        self.msizes = tuple(int(m) for m in matryoshka_sizes)
        self.raw = np.zeros((0, self.dim), dtype=F32)
        self.sp_indptr = np.zeros(1, dtype=np.int64)
        self.sp_idx = np.zeros(0, dtype=np.int64)
        self.sp_val = np.zeros(0, dtype=F32)
        self._final = False

    # store_document_vectors (qdrant_handler.py:120-198): dense + sparse per chunk
    def add(self, dense: np.ndarray, sp_indptr=None, sp_idx=None, sp_val=None):
        dense = np.ascontiguousarray(dense, dtype=F32).reshape(-1, self.dim)
        n = dense.shape[0]
        self.raw = np.concatenate([self.raw, dense], axis=0)
        if sp_indptr is None:
            sp_indptr = np.zeros(n + 1, dtype=np.int64)
            sp_idx = np.zeros(0, dtype=np.int64)
            sp_val = np.zeros(0, dtype=F32)
        sp_indptr = np.asarray(sp_indptr, dtype=np.int64)
        assert sp_indptr.shape[0] == n + 1
        base = self.sp_indptr[-1]
        self.sp_indptr = np.concatenate([self.sp_indptr, base + sp_indptr[1:]])
        self.sp_idx = np.concatenate([self.sp_idx, np.asarray(sp_idx, dtype=np.int64)])
        self.sp_val = np.concatenate([self.sp_val, np.asarray(sp_val, dtype=F32)])
        self._final = False

    @property
    def n(self):
        return self.raw.shape[0]

    def finalize(self):
        if self._final:
            return
        self.dense = cosine_preprocess(self.raw)                      # "dense"
        self.prefix = {m: cosine_preprocess(self.raw[:, :m]) for m in self.msizes}
        self.q8 = quantize_i8(self.raw)                               # "quantized"
        self.q8_rinv = i8_norm_inv(self.q8)
        # inverted view of the sparse matrix: doc-major CSR is enough for the oracle
        self._final = True

    # ---- whole-collection stages -----------------------------------------
    def search_dense(self, q_raw: np.ndarray, limit: int, prefix: int = 0):
        """Prefetch(query=dense_vector[:d], using="matryoshka_d"|"dense", limit)
        (qdrant_handler.py:311-315, 327-329).  prefix=0 means the full vector."""
        self.finalize()
        if prefix:
            qn = cosine_preprocess(np.asarray(q_raw, dtype=F32)[:prefix])
            s = spec_dot(self.prefix[prefix], qn)
        else:
            qn = cosine_preprocess(np.asarray(q_raw, dtype=F32))
            s = spec_dot(self.dense, qn)
        return topk(s, np.arange(self.n), limit)

    def search_i8(self, q_raw: np.ndarray, limit: int):
        """Prefetch(query=quantized_query, using="quantized", limit)
        (qdrant_handler.py:299-302, 335-339)."""
        self.finalize()
        q8 = quantize_i8(np.asarray(q_raw, dtype=F32))
        rq = i8_norm_inv(q8[None, :])[0]
        dot = self.q8.astype(np.int32) @ q8.astype(np.int32)
        s = ((dot.astype(F32) * self.q8_rinv) * rq).astype(F32)
        return topk(s, np.arange(self.n), limit)

    def sparse_scores(self, q_idx, q_val, fix_bits="default"):
        """score(d) = sum over the query's terms of q_t * d_t; only docs sharing >= 1
        term are candidates; IDF-free (a-6).
        fix_bits None: the terms in ascending term id, acc = f32(acc + f32(q_t * d_t)) from +0 -- the
        running sum of a term-at-a-time inverted-index search whose query indices are sorted (assumed
        upstream behaviour).  fix_bits k: each product exact in fp64, scaled by 2^k, rounded (ties to
        even) to int64; the integer sum is converted once to fp32 (order-independent).
        Returns (touched_doc_ids, scores)."""
        fix_bits = SPARSE_FIX_BITS if fix_bits == "default" else fix_bits
        q_idx = np.asarray(q_idx, dtype=np.int64)
        q_val = np.asarray(q_val, dtype=F32)
        order = np.argsort(q_idx, kind="stable")
        q_idx, q_val = q_idx[order], q_val[order]
        touched = np.zeros(self.n, dtype=bool)
        doc_of = np.repeat(np.arange(self.n), np.diff(self.sp_indptr))
        if fix_bits is None:
            acc = np.zeros(self.n, dtype=F32)
            for t, w in zip(q_idx, q_val):
                m = self.sp_idx == t
                d = doc_of[m]                    # a term occurs once per document: no duplicate d
                acc[d] = (acc[d] + (F32(w) * self.sp_val[m]).astype(F32)).astype(F32)
                touched[d] = True
            ids = np.nonzero(touched)[0]
            return ids, acc[ids]
        acc = np.zeros(self.n, dtype=np.int64)
        scale = float(1 << fix_bits)
        for t, w in zip(q_idx, q_val):
            m = self.sp_idx == t
            d = doc_of[m]
            fx = np.rint((np.float64(w) * self.sp_val[m].astype(np.float64)) * scale).astype(np.int64)
            np.add.at(acc, d, fx)
            touched[d] = True
        ids = np.nonzero(touched)[0]
        return ids, (acc[ids].astype(F32) * F32(2.0 ** -fix_bits)).astype(F32)

    def search_sparse(self, q_idx, q_val, limit: int, fix_bits="default"):
        """Prefetch(query=SparseVector, using="sparse", limit) (qdrant_handler.py:347-354)."""
        ids, s = self.sparse_scores(q_idx, q_val, fix_bits)
        return topk(s, ids, limit)

    # ---- candidate re-scoring (outer levels of a nested Prefetch) ----------
    def rescore(self, q_raw: np.ndarray, cand_ids: np.ndarray, limit: int, prefix: int = 0):
        """Re-score only ``cand_ids`` (deduplicated) with one named vector and keep
        ``limit`` (qdrant_handler.py:307-330, 333-344, 363-372)."""
        self.finalize()
        cand = np.unique(np.asarray(cand_ids, dtype=np.int64))
        if prefix:
            qn = cosine_preprocess(np.asarray(q_raw, dtype=F32)[:prefix])
            s = spec_dot(self.prefix[prefix][cand], qn)
        else:
            qn = cosine_preprocess(np.asarray(q_raw, dtype=F32))
            s = spec_dot(self.dense[cand], qn)
        return topk(s, cand, limit)


def rrf(lists, limit=None, k=None, rank_base=None):
    """Reciprocal-rank fusion (qdrant_handler.py:357-360):
    score(d) = sum over lists (in list order) of 1/(k + rank) in fp32, rank
    0-based; dedupe by id; (score desc, id asc); un-limited prefetch => 10."""
    k = RRF_K if k is None else k
    rank_base = RRF_RANK_BASE if rank_base is None else rank_base
    limit = PREFETCH_DEFAULT_LIMIT if limit is None else limit
    acc = {}
    for ids in lists:
        for r, d in enumerate(np.asarray(ids, dtype=np.int64).tolist()):
            c = F32(1.0) / (F32(r + rank_base) + F32(k))
            acc[d] = F32(acc.get(d, F32(0.0)) + c)
    if not acc:
        return np.zeros(0, F32), np.zeros(0, np.int64)
    ids = np.fromiter(acc.keys(), dtype=np.int64)
    sc = np.array([acc[d] for d in ids.tolist()], dtype=F32)
    return topk(sc, ids, limit)


SEARCH_PARAM_KEYS = ("matryoshka_64_limit", "matryoshka_128_limit", "matryoshka_256_limit",
                     "dense_limit", "quantized_limit", "sparse_limit", "final_limit", "hnsw_ef")


def hybrid_tree(ix: OracleIndex, q_raw, q_sp_idx, q_sp_val, params: dict):
    """The reference query (qdrant_handler.py:305-372), every stage exhaustive
    ("exact" mode).  Returns (scores, ids): <= final_limit points scored by the
    root stage's full-D dense cosine."""
    m = sorted(ix.msizes)
    limits = [params[f"matryoshka_{d}_limit"] for d in m]
    # matryoshka cascade :305-330
    _, c = ix.search_dense(q_raw, limits[0], prefix=m[0])
    for d, lim in zip(m[1:], limits[1:]):
        _, c = ix.rescore(q_raw, c, lim, prefix=d)
    _, cand_a = ix.rescore(q_raw, c, params["dense_limit"])
    # quantized -> dense refinement :333-344
    _, cq = ix.search_i8(q_raw, params["quantized_limit"])
    _, cand_q = ix.rescore(q_raw, cq, params["dense_limit"])
    # sparse :347-354
    _, cand_s = ix.search_sparse(q_sp_idx, q_sp_val, params["sparse_limit"])
    # RRF :357-360 (no limit => PREFETCH_DEFAULT_LIMIT)
    _, cand_r = rrf([cand_q, cand_s])
    # root :363-372 -- union re-scored by dense cosine
    return ix.rescore(q_raw, np.concatenate([cand_a, cand_r]), params["final_limit"])


def hybrid_h1(ix: OracleIndex, q_raw, q_sp_idx, q_sp_val, dense_limit=100, sparse_limit=100, limit=10):
    """H1 "simple hybrid" (SURVEY.md §8d): dense top-L (+) sparse top-L -> RRF -> top-10.
    Returned scores are RRF scores."""
    _, cd = ix.search_dense(q_raw, dense_limit)
    _, cs = ix.search_sparse(q_sp_idx, q_sp_val, sparse_limit)
    return rrf([cd, cs], limit=limit)


# ----------------------------------------------------------------------------
# Synthetic data -- integer-exact, shared with C oracle and HIP
# ----------------------------------------------------------------------------
SEED_CORPUS, SEED_QUERY, SEED_SPDOC, SEED_SPQUERY = 0x5EED0001, 0x5EED0002, 0x5EED0003, 0x5EED0004
SYNTH_V = 1 << 20
SYNTH_ZIPF_S = 1.07
SYNTH_STOP = 128
TERM_MULT = 0x9E3779B1
M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    h = np.asarray(h, dtype=np.uint64) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def hash2(seed, a, b):
    """hash32(seed, a, b) = fmix32(fmix32(seed + a*0x9E3779B1) ^ (b*0x85EBCA77))."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    h = fmix32((np.uint64(seed) + a * np.uint64(0x9E3779B1)) & M32)
    return fmix32(h ^ ((b * np.uint64(0x85EBCA77)) & M32))


def synth_dense(seed: int, row0: int, n: int, dim: int) -> np.ndarray:
    """Row r, column c: x = (int32(hash32(seed, r, c)) >> 8) * 2^-23 in [-1, 1)."""
    r = np.arange(row0, row0 + n, dtype=np.uint64)[:, None]
    c = np.arange(dim, dtype=np.uint64)[None, :]
    h = hash2(seed, r, c).astype(np.uint32).view(np.int32)
    return ((h >> 8).astype(F32) * F32(2.0 ** -23)).astype(F32)


def synth_tables():
    """(cdf_u32[V], len_u16[256]) shared lookup tables: Zipf(s) CDF scaled to 2^32
    and 256 quantiles of lognormal(ln 120, 0.4) clipped to [8, 512]."""
    from scipy.special import ndtri
    p = np.arange(1, SYNTH_V + 1, dtype=np.float64) ** (-SYNTH_ZIPF_S)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    cdf_u32 = np.minimum(np.floor(cdf * 4294967296.0), 4294967295.0).astype(np.uint32)
    cdf_u32[-1] = 0xFFFFFFFF
    qs = (np.arange(256, dtype=np.float64) + 0.5) / 256.0
    ln = np.exp(np.log(120.0) + 0.4 * ndtri(qs))
    len_u16 = np.clip(np.rint(ln), 8, 512).astype(np.uint16)
    return cdf_u32, len_u16


def rank_to_term(rank):
    return ((np.asarray(rank, dtype=np.uint64) * np.uint64(TERM_MULT)) & np.uint64(0x7FFFFFFF)).astype(np.int64)


def synth_sparse_docs(seed: int, doc0: int, n: int, tables=None):
    """Doc d: L = len[hash(seed,d,0xFFFFFFFF) & 255] tokens; token i has
    u = ((i << 32) + hash(seed, d, i)) // L, rank = #{r : cdf[r] <= u}; equal
    ranks are runs -> tf; term = rank*0x9E3779B1 mod 2^31; value = BM25 tf weight
    (k=1.2, b=0.75, avg_len=256, doc_len = L).  Returns CSR (indptr, idx, val)."""
    cdf, lens = tables or synth_tables()
    indptr = [0]
    idx, val = [], []
    for d in range(doc0, doc0 + n):
        L = int(lens[int(hash2(seed, d, 0xFFFFFFFF)) & 255])
        i = np.arange(L, dtype=np.uint64)
        u = ((i << np.uint64(32)) + hash2(seed, np.uint64(d), i)) // np.uint64(L)
        rank = np.minimum(np.searchsorted(cdf, u.astype(np.uint32), side="right"), SYNTH_V - 1)
        r, tf = np.unique(rank, return_counts=True)
        idx.append(rank_to_term(r))
        val.append(bm25_weight(tf, L))
        indptr.append(indptr[-1] + len(r))
    return (np.asarray(indptr, dtype=np.int64),
            np.concatenate(idx) if idx else np.zeros(0, np.int64),
            np.concatenate(val) if val else np.zeros(0, F32))


def synth_sparse_queries(seed: int, q0: int, n: int, tables=None):
    """Query q: T = 3 + hash(seed,q,0xFFFFFFFF) % 10 tokens drawn from the Zipf
    CDF restricted to rank >= 128 (stop-word removal): u = c0 + ((h*(2^32-c0))>>32),
    c0 = cdf[127]; duplicates merge into tf; weights by the document formula with
    doc_len = T (the reference uses .embed() for queries, embedding_handler.py:123);
    terms sorted by term id."""
    cdf, _ = tables or synth_tables()
    c0 = np.uint64(cdf[SYNTH_STOP - 1])
    indptr = [0]
    idx, val = [], []
    for q in range(q0, q0 + n):
        T = 3 + int(hash2(seed, q, 0xFFFFFFFF)) % 10
        h = hash2(seed, np.uint64(q), np.arange(T, dtype=np.uint64))
        u = c0 + ((h * (np.uint64(1 << 32) - c0)) >> np.uint64(32))
        rank = np.minimum(np.searchsorted(cdf, u.astype(np.uint32), side="right"), SYNTH_V - 1)
        r, tf = np.unique(rank, return_counts=True)
        t = rank_to_term(r)
        o = np.argsort(t, kind="stable")
        idx.append(t[o])
        val.append(bm25_weight(tf, T)[o])
        indptr.append(indptr[-1] + len(r))
    return (np.asarray(indptr, dtype=np.int64), np.concatenate(idx), np.concatenate(val))


# ---- IndexerAPI search_across_spaces ------------------------------------------------
# IndexerAPI/src/core/storage/neo4j_handler.py:809-1047.  Assumed upstream behaviour, unverifiable
# offline: a cosine vector index scores (1 + cos) / 2 and the top-k is exact.
SCOUT_SPACES = ("page", "entity", "column", "relationship")
SCOUT_SCORE_AFFINE = True      # switch: Neo4j's (1 + cos) / 2; False = raw cosine


def scout_score(cos: np.ndarray) -> np.ndarray:
    cos = np.asarray(cos, F32)
    if not SCOUT_SCORE_AFFINE:
        return cos
    return ((F32(1.0) + cos) * F32(0.5)).astype(F32)


def scout_search(spaces: dict, q_raw: np.ndarray, top_k: int, user_id: str, org_id: str):
    """spaces: name -> (X raw [n, d], user_ids, org_ids).  Returns [(space, row, score)] as
    neo4j_handler.py:809-827: per space exact cosine top-k (score desc, row asc), tenant filter AFTER
    the top-k, concatenation in SCOUT_SPACES order, stable sort by score descending, [:limit]."""
    limit = max(1, int(top_k))
    agg = []
    for name in SCOUT_SPACES:
        if name not in spaces or len(spaces[name][1]) == 0:
            continue
        X, users, orgs = spaces[name]
        Xn = cosine_preprocess(np.asarray(X, F32))
        qn = cosine_preprocess(np.asarray(q_raw, F32)[None, :])[0]
        sc = spec_dot(Xn, qn)
        s, i = topk(sc, np.arange(Xn.shape[0], dtype=np.int64), limit)
        t = scout_score(s)
        for r in range(len(i)):
            row = int(i[r])
            if users[row] == user_id and orgs[row] == org_id:
                agg.append((name, row, float(t[r])))
    agg.sort(key=lambda it: it[2], reverse=True)
    return agg[:limit]
