"""ctypes face of oracle/libhx_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhx_oracle.so")
_lib = None
F32 = np.float32


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "hx_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.run(["make", "-C", HERE, "-B", "libhx_oracle.so"], check=True, capture_output=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        l = C.CDLL(LIB)
        l.ho_spec_dot.restype = C.c_float
        l.ho_inv_build.restype = C.c_void_p
        l.ho_num_threads.restype = C.c_int
        _lib = l
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def num_threads() -> int:
    return lib().ho_num_threads()


def set_num_threads(n: int) -> None:
    lib().ho_set_num_threads(int(n))


def spec_dot(x, q) -> float:
    x = np.ascontiguousarray(x, F32)
    q = np.ascontiguousarray(q, F32)
    return float(lib().ho_spec_dot(_p(x), _p(q), C.c_int(x.shape[0])))


def cosine_preprocess(X, d=None, skip_if_unit=True):
    X = np.ascontiguousarray(X, F32)
    n, stride = X.shape
    d = stride if d is None else d
    out = np.empty((n, d), F32)
    lib().ho_cosine_preprocess(_p(X), C.c_int64(n), C.c_int(stride), C.c_int(d), _p(out), C.c_int(int(skip_if_unit)))
    return out


def quantize_i8(X):
    X = np.ascontiguousarray(X, F32)
    n, dim = X.shape
    out = np.empty((n, dim), np.int8)
    rinv = np.empty(n, F32)
    lib().ho_quantize_i8(_p(X), C.c_int64(n), C.c_int(dim), _p(out), _p(rinv))
    return out, rinv


def search_dense(Xn, Qn, L, id_base=0):
    Xn = np.ascontiguousarray(Xn, F32)
    Qn = np.ascontiguousarray(Qn, F32)
    n, d = Xn.shape
    B = Qn.shape[0]
    s = np.empty((B, L), F32)
    i = np.empty((B, L), np.int64)
    c = np.empty(B, np.int32)
    lib().ho_search_dense(_p(Xn), C.c_int64(n), C.c_int(d), _p(Qn), C.c_int(B), C.c_int(L), C.c_int64(id_base),
                          _p(s), _p(i), _p(c))
    return s, i, c


def search_i8(X8, rinv_x, Q8, rinv_q, L, id_base=0):
    X8 = np.ascontiguousarray(X8, np.int8)
    Q8 = np.ascontiguousarray(Q8, np.int8)
    n, dim = X8.shape
    B = Q8.shape[0]
    s = np.empty((B, L), F32)
    i = np.empty((B, L), np.int64)
    c = np.empty(B, np.int32)
    lib().ho_search_i8(_p(X8), _p(np.ascontiguousarray(rinv_x, F32)), C.c_int64(n), C.c_int(dim), _p(Q8),
                       _p(np.ascontiguousarray(rinv_q, F32)), C.c_int(B), C.c_int(L), C.c_int64(id_base),
                       _p(s), _p(i), _p(c))
    return s, i, c


def rescore(Xn, qn, cand, L, id_base=0):
    Xn = np.ascontiguousarray(Xn, F32)
    qn = np.ascontiguousarray(qn, F32)
    cand = np.ascontiguousarray(cand, np.int64)
    n, d = Xn.shape
    s = np.empty(L, F32)
    i = np.empty(L, np.int64)
    c = C.c_int(0)
    lib().ho_rescore(_p(Xn), C.c_int64(n), C.c_int(d), _p(qn), _p(cand), C.c_int(cand.shape[0]), C.c_int(L),
                     C.c_int64(id_base), _p(s), _p(i), C.byref(c))
    return s[:c.value], i[:c.value]


class InvIndex:
    def __init__(self, indptr, idx, val):
        self.indptr = np.ascontiguousarray(indptr, np.int64)
        self.idx = np.ascontiguousarray(idx, np.int32)
        self.val = np.ascontiguousarray(val, F32)
        self.n_docs = self.indptr.shape[0] - 1
        self._h = C.c_void_p(lib().ho_inv_build(_p(self.indptr), _p(self.idx), _p(self.val), C.c_int64(self.n_docs)))

    def search(self, qip, qix, qv, L, id_base=0):
        qip = np.ascontiguousarray(qip, np.int64)
        qix = np.ascontiguousarray(qix, np.int32)
        qv = np.ascontiguousarray(qv, F32)
        B = qip.shape[0] - 1
        s = np.empty((B, L), F32)
        i = np.empty((B, L), np.int64)
        c = np.empty(B, np.int32)
        lib().ho_search_sparse(self._h, _p(qip), _p(qix), _p(qv), C.c_int(B), C.c_int(L), C.c_int64(id_base),
                               _p(s), _p(i), _p(c))
        return s, i, c

    def __del__(self):
        try:
            lib().ho_inv_free(self._h)
        except Exception:
            pass


def set_sparse_fix_bits(bits) -> None:
    """oracle.SPARSE_FIX_BITS for the C restatement: None = fp32 running sum in ascending term id."""
    lib().ho_set_sparse_fix_bits(C.c_int(-1 if bits is None else int(bits)))


def sparse_brute(indptr, idx, val, qip, qix, qv, L, id_base=0):
    """Document-at-a-time sparse top-L straight from the doc-major CSR (ho_sparse_brute)."""
    indptr = np.ascontiguousarray(indptr, np.int64)
    idx = np.ascontiguousarray(idx, np.int32)
    val = np.ascontiguousarray(val, F32)
    qip = np.ascontiguousarray(qip, np.int64)
    qix = np.ascontiguousarray(qix, np.int32)
    qv = np.ascontiguousarray(qv, F32)
    B = qip.shape[0] - 1
    s = np.empty((B, L), F32)
    i = np.empty((B, L), np.int64)
    c = np.empty(B, np.int32)
    lib().ho_sparse_brute(_p(indptr), _p(idx), _p(val), C.c_int64(indptr.shape[0] - 1), _p(qip), _p(qix), _p(qv),
                          C.c_int(B), C.c_int(L), C.c_int64(id_base), _p(s), _p(i), _p(c))
    return s, i, c


def rrf(a, b, k=2.0, rank_base=0, limit=10):
    a = np.ascontiguousarray(a, np.int64)
    b = np.ascontiguousarray(b, np.int64)
    s = np.empty(limit, F32)
    i = np.empty(limit, np.int64)
    c = C.c_int(0)
    lib().ho_rrf(_p(a), C.c_int(a.shape[0]), _p(b), C.c_int(b.shape[0]), C.c_float(k), C.c_int(rank_base),
                 C.c_int(limit), _p(s), _p(i), C.byref(c))
    return s[:c.value], i[:c.value]


def synth_dense(seed, row0, n, dim):
    out = np.empty((n, dim), F32)
    lib().ho_synth_dense(C.c_uint32(seed), C.c_int64(row0), C.c_int64(n), C.c_int(dim), _p(out))
    return out


def synth_sparse_docs(seed, doc0, n, tables):
    cdf = np.ascontiguousarray(tables[0], np.uint32)
    lens = np.ascontiguousarray(tables[1], np.uint16)
    per = np.zeros(n + 1, np.int64)
    lib().ho_synth_sparse_docs(C.c_uint32(seed), C.c_int64(doc0), C.c_int64(n), _p(cdf), C.c_int(cdf.shape[0]),
                               _p(lens), C.c_int(0), _p(per), None, None, None)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(per[:n], out=indptr[1:])
    idx = np.empty(indptr[-1], np.int32)
    val = np.empty(indptr[-1], F32)
    lib().ho_synth_sparse_docs(C.c_uint32(seed), C.c_int64(doc0), C.c_int64(n), _p(cdf), C.c_int(cdf.shape[0]),
                               _p(lens), C.c_int(1), None, _p(indptr), _p(idx), _p(val))
    return indptr, idx, val
