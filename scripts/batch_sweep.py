"""Diagnostic: H1 step time against the batch size on the 10M x 768 hybrid index (looking for cliffs).  argv: rows"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = 768
tabs = synth.tables()
ix = eng.HxIndex(D, (64, 128, 256)); ix.reserve(N)
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
hp = eng.make_params(dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
                          quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128), mode=eng.HX_MODE_H1)
for B in (1, 8, 64, 128, 129, 256, 512, 1024, 1500, 2048, 4096, 5000):
    Q = eng.synth_queries_dense(D, 0, B, synth.SEED_QUERY)
    qip, qix, qv = (torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs))
    f = lambda: ix.hybrid_query(Q, qip, qix, qv, hp)
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    n = 3
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    st = ix.stats()
    print(f"B {B:5d}  {dt * 1e3:8.2f} ms/step  {B / dt:9.0f} q/s   fallbacks dense {st['dense_fallback_queries']} sparse {st['sparse_fallback_queries']}", flush=True)
