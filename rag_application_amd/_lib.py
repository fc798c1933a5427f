"""ctypes binding of include/hx.h.  The product has NO CPU path: a missing or
unloadable libhx.so raises, and every entry point needs a HIP device."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhx.so")

HX_MODE_TREE = 0
HX_MODE_H1 = 1


class HxError(RuntimeError):
    pass


class HxParams(C.Structure):
    _fields_ = [
        ("matryoshka_64_limit", C.c_int32),
        ("matryoshka_128_limit", C.c_int32),
        ("matryoshka_256_limit", C.c_int32),
        ("dense_limit", C.c_int32),
        ("quantized_limit", C.c_int32),
        ("sparse_limit", C.c_int32),
        ("final_limit", C.c_int32),
        ("hnsw_ef", C.c_int32),
        ("rrf_k", C.c_float),
        ("rrf_rank_base", C.c_int32),
        ("rrf_limit", C.c_int32),
        ("mode", C.c_int32),
    ]


class HxStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_rows", "nnz", "n_segments", "n_groups", "hash_capacity", "bytes_dense_f32",
        "bytes_dense_f16", "bytes_i8", "bytes_prefix", "bytes_sparse",
        "dense_fallback_queries", "i8_fallback_queries", "retry_queries", "sparse_fallback_queries",
        "bytes_i8_cand", "cand8_queries", "cand8_uncertified_queries")] + [("cand8_row_error_max", C.c_double),
                                                                              ("tree_batches_redone", C.c_int64),
                                                                              ("cand8_switched_off", C.c_int64),
                                                                              ("tree_deferral_switched_off", C.c_int64)]


class HxProf(C.Structure):
    _fields_ = [("launches", C.c_int64 * 6), ("ms", C.c_double * 6), ("flops", C.c_double * 6),
                ("bytes", C.c_double * 6)]


_P = C.c_void_p
_SIGS = {
    "hx_create": [C.c_int32, _P, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_P)],
    "hx_destroy": [_P],
    "hx_abi_version": [],
    "hx_reserve": [_P, C.c_int64, C.c_int64],
    "hx_add_dense": [_P, _P, C.c_int64],
    "hx_add_dense_dev": [_P, _P, C.c_int64, _P],
    "hx_add_sparse": [_P, _P, _P, _P, C.c_int64],
    "hx_add_rows": [_P, _P, _P, _P, _P, C.c_int64],
    "hx_add_rows_dev": [_P, _P, _P, _P, _P, C.c_int64, _P],
    "hx_set_next_id": [_P, C.c_int64],
    "hx_truncate": [_P, C.c_int64],
    "hx_finalize": [_P],
    "hx_count": [_P, C.POINTER(C.c_int64)],
    "hx_nnz": [_P, C.POINTER(C.c_int64)],
    "hx_synth_fill": [_P, C.c_int64, C.c_uint32, C.c_uint32, _P, C.c_int32, _P, C.c_int32],
    "hx_synth_queries_dense": [C.c_int32, C.c_int64, C.c_int32, C.c_uint32, _P, _P],
    "hx_search_dense": [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P],
    "hx_search_i8": [_P, _P, C.c_int32, C.c_int32, _P, _P, _P],
    "hx_search_sparse": [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P],
    "hx_search_dense_async": [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P],
    "hx_search_i8_async": [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P],
    "hx_search_sparse_async": [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P],
    "hx_rescore": [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, _P],
    "hx_rrf": [C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, C.c_float, C.c_int32,
               C.c_int32, _P, _P, _P],
    "hx_merge": [C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P],
    "hx_h1_local": [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P],
    "hx_h1_local_async": [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P],
    "hx_h1_fuse": [C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                   _P, _P, _P],
    "hx_h1_plan": [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "hx_sparse_wmax": [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)],
    "hx_set_sparse_wmax": [_P, C.c_float],
    "hx_h1_nominate_async": [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P],
    "hx_h1_rescore_async": [_P, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                            C.c_int32, C.c_int32, C.c_int32, _P, _P],
    "hx_h1_finish": [C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                     C.c_int32, _P, _P, _P, _P],
    "hx_unpack": [C.c_int32, _P, C.c_int64, _P, _P, _P],
    "hx_hybrid_query_host": [_P, _P, _P, _P, _P, C.c_int32, C.POINTER(HxParams), _P, _P, _P],
    "hx_hybrid_query_dev": [_P, _P, _P, _P, _P, C.c_int32, C.POINTER(HxParams), _P, _P, _P],
    "hx_bm25_embed_batch": [_P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int32, _P, _P, _P, C.c_int64, _P],
    "hx_save": [_P, C.c_char_p],
    "hx_load": [C.c_char_p, C.c_int32, C.POINTER(_P)],
    "hx_get_stats": [_P, C.POINTER(HxStats)],
    "hx_debug_row": [_P, C.c_int32, C.c_int64, _P],
    "hx_set_dense_candidates": [_P, C.c_int32],
    "hx_set_stream_overlap": [_P, C.c_int32],
    "hx_rebuild_sparse": [_P],
    "hx_profile": [_P, C.c_int32],
    "hx_profile_read": [_P, C.POINTER(HxProf)],
}
EXPORTS = tuple(_SIGS) + ("hx_last_error",)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("HX_LIB_PATH") or LIB_PATH   # override: diagnostic builds (scripts/) only
        if not os.path.exists(path):
            raise HxError(f"{path} is missing: run `python -m rag_application_amd.build` "
                          "(the engine has no CPU fallback)")
        try:   # PyTorch's HIP runtime first: loaded the other way round, libhx's calls see no device
            import torch
            torch.cuda.is_available()
        except Exception:
            pass
        l = C.CDLL(path)
        for name, args in _SIGS.items():
            f = getattr(l, name)
            f.argtypes = args
            f.restype = C.c_int
        l.hx_last_error.argtypes = []
        l.hx_last_error.restype = C.c_char_p
        if l.hx_abi_version() != 3:
            raise HxError("libhx ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise HxError(lib().hx_last_error().decode("utf-8", "replace"))
