"""Diagnostic: the sparse stage alone on the synthetic corpus.  argv: rows [batch] [limit]"""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
tabs = synth.tables()
ix = eng.HxIndex(64, ())
ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = (torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs))
for _ in range(2): ix.search_sparse(qip, qix, qv, L)
torch.cuda.synchronize(); ix.profile(True); ix.profile_read(); t0 = time.perf_counter()
for _ in range(5): k, c = ix.search_sparse(qip, qix, qv, L)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
p = ix.profile_read()["sparse"]
print("sparse stage ms", dt * 1e3, "select kernel ms", p["ms"] / max(p["launches"], 1), "alg GB", p["bytes"] / max(p["launches"], 1) / 1e9,
      "GB/s", p["bytes"] / p["ms"] / 1e6 if p["ms"] else 0, "stats", {k: v for k, v in ix.stats().items() if k in ("n_segments", "sparse_fallback_queries", "nnz")})
