"""Diagnostic: the reference tree at B = 1024 on 10M x 768, a few calls, meant for `rocprofv3 --kernel-trace`: the per-kernel
timeline of one call.   argv: rows"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = 1024
tabs = synth.tables()
P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=40,
         quantized_limit=40, sparse_limit=50, final_limit=30, hnsw_ef=128)
hp = eng.make_params(P, mode=eng.HX_MODE_TREE)
ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
Q = eng.synth_queries_dense(768, 0, B, synth.SEED_QUERY)
t = [torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)]
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ix.hybrid_query(Q, *t, hp); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("tree ms", [round(x, 3) for x in ts])
