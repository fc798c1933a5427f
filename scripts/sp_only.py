"""Runs only the sparse stage (for rocprofv3 --pmc on k_sparse_score)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
B = 1024
tabs = synth.tables()
ix = eng.HxIndex(768, (64,))
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
for _ in range(3): ix.search_sparse(*t, 100)
torch.cuda.synchronize()
t0 = time.time(); ix.search_sparse(*t, 100); torch.cuda.synchronize(); print("sparse ms", (time.time() - t0) * 1e3)
