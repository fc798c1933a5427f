"""Diagnostic: a few small H1 batches, six calls each, meant to run under `rocprofv3 --kernel-trace`: the per-kernel timeline of a
fast and a slow call of the same batch (profiles/r04_h1_small_batch.txt, item 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_application_amd import engine as eng, synth
N = 10_000_000
tabs = synth.tables()
P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
         quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128)
hp = eng.make_params(P, mode=eng.HX_MODE_H1)
ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
for B, q0 in ((4, 5), (4, 9), (2, 5), (2, 40), (1, 0)):
    Q = eng.synth_queries_dense(768, q0, B, synth.SEED_QUERY)
    t = [torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, q0, B, tabs)]
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); ix.hybrid_query(Q, *t, hp); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(B, q0, "ms", [round(x, 3) for x in ts], flush=True)
    time.sleep(0.05)
