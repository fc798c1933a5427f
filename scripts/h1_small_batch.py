"""Diagnostic: the H1 step (dense top-100 (+) sparse top-100 -> RRF -> top-10) of hx_hybrid_query_dev at small batches on a
10M x 768 corpus, with the sparse stage started beside the dense stage's TAIL (every batch size; round 4) or beside its SCAN
(HX_DEBUG_FORK_EARLY_MAX: batches up to that size).  The lists must not depend on it.   argv: rows"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
BS = (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024)
tabs = synth.tables()
P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
         quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128)
hp = eng.make_params(P, mode=eng.HX_MODE_H1)
res, lists, lat, tree = {}, {}, {}, {}
hpt = eng.make_params(P, mode=eng.HX_MODE_TREE)
for setting in ("0", "4096"):
    os.environ["HX_DEBUG_FORK_EARLY_MAX"] = setting
    ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
    for B in BS:
        Q = eng.synth_queries_dense(768, 0, B, synth.SEED_QUERY)
        t = [torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)]
        for _ in range(3):
            k, c = ix.hybrid_query(Q, *t, hp)
        torch.cuda.synchronize()
        n = 30 if B <= 256 else 10
        t0 = time.perf_counter()
        for _ in range(n):
            k, c = ix.hybrid_query(Q, *t, hp)
        torch.cuda.synchronize()
        res[(setting, B)] = (time.perf_counter() - t0) / n * 1e3
        ts = []
        for _ in range(n):                      # ... and one call at a time (the latency of a call on an idle device)
            t0 = time.perf_counter()
            ix.hybrid_query(Q, *t, hp)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        lat[(setting, B)] = float(np.median(ts))
        for _ in range(3):
            ix.hybrid_query(Q, *t, hpt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ix.hybrid_query(Q, *t, hpt)
        torch.cuda.synchronize()
        tree[(setting, B)] = (time.perf_counter() - t0) / n * 1e3
        if setting == "0":
            lists[B] = (k.clone(), c.clone())
        else:
            assert torch.equal(k, lists[B][0]) and torch.equal(c, lists[B][1]), f"lists differ at B = {B}"
    ix.close()
print("ms per call (inputs resident); sparse stage beside the dense TAIL / beside the SCAN: H1 back to back | H1 one call at a time | tree back to back")
for B in BS:
    print(f"  B {B:5d}   {res[('0', B)]:8.3f} {res[('4096', B)]:8.3f}  |  {lat[('0', B)]:8.3f} {lat[('4096', B)]:8.3f}  |  {tree[('0', B)]:8.3f} {tree[('4096', B)]:8.3f}")
