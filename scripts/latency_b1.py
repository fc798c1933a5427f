"""Diagnostic: latency of ONE query (B = 1) through the reference tree and through H1 on a 10M x 768 corpus --
the shape of the reference's own use (hybrid_search is called per user query).  argv: rows"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tabs = synth.tables()
ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
         quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128)
out = {"rows": N}
for B in (1, 8):
    Q = eng.synth_queries_dense(768, 0, B, synth.SEED_QUERY)
    qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
    t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
    for name, mode in (("tree", eng.HX_MODE_TREE), ("h1", eng.HX_MODE_H1)):
        hp = eng.make_params(P, mode=mode)
        for _ in range(3):
            ix.hybrid_query(Q, *t, hp)
        torch.cuda.synchronize(); ts = []
        for _ in range(20):
            t0 = time.perf_counter(); ix.hybrid_query(Q, *t, hp); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out[f"{name}_B{B}_ms_p50"] = round(float(np.median(ts)), 3)
        out[f"{name}_B{B}_ms_max"] = round(float(np.max(ts)), 3)
print(json.dumps(out))
