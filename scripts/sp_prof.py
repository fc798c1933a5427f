"""Diagnostic: sparse stage time, postings visited, algorithmic GB/s.  argv: rows [B]"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tabs = synth.tables()
ix = eng.HxIndex(768, (64,))
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
for _ in range(2): ix.search_sparse(*t, 100)
torch.cuda.synchronize()
ix.profile(True); ix.profile_read()
for _ in range(3): ix.search_sparse(*t, 100)
torch.cuda.synchronize()
p = ix.profile_read()["sparse"]; ix.profile(False)
st = ix.stats()
print(f"rows {N} nnz {st['nnz']} B {B} terms/query {len(qix)/B:.2f}: {p['ms']/3:.3f} ms/launch, bytes/launch {p['bytes']/3/1e9:.3f} GB "
      f"-> {p['bytes']/p['ms']/1e6:.1f} GB/s; postings/query {p['bytes']/3/8/B:.0f}; segments {st['n_segments']}")
