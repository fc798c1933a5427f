#!/usr/bin/env python3
"""Regenerates the committed golden fixtures from the numpy oracle.

The reference holds NO golden vectors for this path ("parity
unpinned"), so these fixtures are authored by this repo: hand-computed known-answer
values (kat.json), the output of the reference's own numpy quantisation expression
evaluated here (i8_kat.json), and seeded-corpus results of oracle/oracle.py
(corpus_*.npz).  Run from the repo root:  python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

P_MCP = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=40,
             quantized_limit=40, sparse_limit=50, final_limit=30, hnsw_ef=128)


def p_fallback(n):
    return dict(matryoshka_64_limit=min(500, n // 10), matryoshka_128_limit=min(400, n // 15),
                matryoshka_256_limit=min(300, n // 20), dense_limit=min(200, n // 25),
                quantized_limit=min(300, n // 30), sparse_limit=min(100, n // 50), hnsw_ef=256, final_limit=10)


def pad(lists, L):
    ids = np.full((len(lists), L), -1, np.int64)
    bits = np.zeros((len(lists), L), np.uint32)
    cnt = np.zeros(len(lists), np.int32)
    for b, (s, i) in enumerate(lists):
        ids[b, :len(i)] = i
        bits[b, :len(i)] = np.asarray(s, np.float32).view(np.uint32)
        cnt[b] = len(i)
    return ids, bits, cnt


def main():
    # ---- known-answer values (SURVEY.md §8c ii-iv), hand-computed ----------------------
    kat = {
        "murmur3_x86_32": [["", 0], ["hello", 613153351], ["foo", 4138058784]],
        "bm25_term_id": [["foo", 156908512], ["hello", 613153351]],
        "bm25_weight": [[1, 256, 1.0], [2, 256, 1.375], [1, 128, 1.2571428571428571]],
        "rrf": {"lists": [[1, 2, 3], [2, 4]], "ids": [2, 1, 4, 3],
                "scores": [1 / 2 + 1 / 3, 1 / 2, 1 / 3, 1 / 4]},
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # ---- the reference's quantisation line, evaluated by numpy itself --------------------
    xs = [1.0, 0.999, -0.999, 0.0078, -0.0078, 0.0079, -0.0079, 1.6, -1.6, 2.5, 3.0, 100.0,
          0.5, -0.5, 0.0, 1e10, -1e10, 16909320.5, -16909320.5]
    xs32 = np.asarray(xs, np.float32)
    with np.errstate(all="ignore"):
        # qdrant_handler.py:144-146 -- np.array(list of python floats) is float64
        expect = np.clip((np.array(xs32.astype(np.float64).tolist()) * 127).astype(np.int8), -128, 127)
    json.dump({"x_f32_bits": xs32.view(np.uint32).tolist(), "expected_i8": expect.tolist()},
              open(os.path.join(HERE, "i8_kat.json"), "w"), indent=1)

    tabs = O.synth_tables()
    # ---- corpus A: 2048 x 768 hybrid, 32 queries, every stage + both parameter sets ----------
    n, dim, B = 2048, 768, 32
    ora = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
    ora.add(O.synth_dense(O.SEED_CORPUS, 0, n, dim), ip, si, sv)
    ora.finalize()
    Q = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    sp = lambda b: (qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]])  # noqa: E731
    out = {}
    for name, fn, L in (
        ("dense", lambda b: ora.search_dense(Q[b], 10), 10),
        ("m64", lambda b: ora.search_dense(Q[b], 10, 64), 10),
        ("i8", lambda b: ora.search_i8(Q[b], 10), 10),
        ("sparse", lambda b: ora.search_sparse(*sp(b), 10), 10),
        ("tree_mcp", lambda b: O.hybrid_tree(ora, Q[b], *sp(b), P_MCP), 30),
        ("tree_fallback", lambda b: O.hybrid_tree(ora, Q[b], *sp(b), p_fallback(n)), 10),
        ("h1", lambda b: O.hybrid_h1(ora, Q[b], *sp(b), 100, 100, 10), 10),
    ):
        ids, bits, cnt = pad([fn(b) for b in range(B)], L)
        out[name + "_ids"], out[name + "_bits"], out[name + "_cnt"] = ids, bits, cnt
    np.savez_compressed(os.path.join(HERE, "corpus_a_2048x768.npz"), **out)

    # ---- corpus B: 4096 x 64 dense only ------------------------------------------------------
    n, dim = 4096, 64
    orb = O.OracleIndex(dim, ())
    orb.add(O.synth_dense(O.SEED_CORPUS, 0, n, dim))
    orb.finalize()
    Qb = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    ids, bits, cnt = pad([orb.search_dense(Qb[b], 10) for b in range(B)], 10)
    np.savez_compressed(os.path.join(HERE, "corpus_b_4096x64.npz"), dense_ids=ids, dense_bits=bits, dense_cnt=cnt)

    # ---- ties: duplicate rows and equal sparse scores pin (score desc, id asc) -------------------
    base = O.synth_dense(77, 0, 16, 128)
    X = base[np.arange(512) % 16]
    ort = O.OracleIndex(128, ())
    sp_ip = np.arange(513, dtype=np.int64)           # every doc holds term 5 with weight 1.0
    ort.add(X, sp_ip, np.full(512, 5, np.int64), np.ones(512, np.float32))
    ort.finalize()
    Qt = O.synth_dense(78, 0, 4, 128)
    ids, bits, cnt = pad([ort.search_dense(Qt[b], 40) for b in range(4)], 40)
    sids, sbits, scnt = pad([ort.search_sparse([5], [2.0], 20) for _ in range(1)], 20)
    np.savez_compressed(os.path.join(HERE, "ties_512x128.npz"), dense_ids=ids, dense_bits=bits, dense_cnt=cnt,
                        sparse_ids=sids, sparse_bits=sbits, sparse_cnt=scnt)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
