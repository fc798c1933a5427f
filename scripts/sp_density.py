"""Diagnostic: how the sparse select kernel's time depends on the postings per visit at a FIXED number of visits: the same
10M-document index and 1024 queries, but query terms more frequent than a rank cut are dropped.  argv: rows"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B, L = 1024, 100
tabs = synth.tables()
ix = eng.HxIndex(64, ())
ix.reserve(N); ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
inv = pow(0x9E3779B1, -1, 1 << 31)
rank = (qix.astype(np.int64) * inv) & 0x7FFFFFFF
for cut in (0, 300, 1000, 3000, 10000, 100000):
    keep = rank >= cut
    ip = np.zeros(B + 1, np.int64)
    for b in range(B):
        ip[b + 1] = ip[b] + int(keep[qip[b]:qip[b + 1]].sum())
    t = (torch.from_numpy(ip).cuda(), torch.from_numpy(qix[keep].astype(np.int32)).cuda(), torch.from_numpy(qv[keep]).cuda())
    for _ in range(2): ix.search_sparse(*t, L)
    torch.cuda.synchronize(); ix.profile(True); ix.profile_read()
    for _ in range(5): ix.search_sparse(*t, L)
    torch.cuda.synchronize()
    p = ix.profile_read()["sparse"]; ix.profile(False)
    print(f"rank >= {cut:6d}: terms/query {keep.sum() / B:5.2f}  postings/visit {p['bytes'] / p['launches'] / 8 / (B * 153):8.0f}  "
          f"select {p['ms'] / p['launches']:6.3f} ms  {p['bytes'] / p['ms'] / 1e6:7.0f} GB/s", flush=True)
