"""Parity at the edges of the parameter space (against the C restatement): dim 4096, batch 4100
(beyond the LDS threshold table of scan8: atomic-append kernel), limit 2048, one row, dim 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O, c_oracle as CO
from rag_application_amd import engine as eng

def check(n, dim, B, L, seed=5, i8=True):
    X = O.synth_dense(seed, 0, n, dim); Q = O.synth_dense(seed + 1, 0, B, dim)
    ix = eng.HxIndex(dim, ()); ix.add(X)
    es, ei, ec = CO.search_dense(CO.cosine_preprocess(X), CO.cosine_preprocess(Q), L)
    s, i = eng.unpack(ix.search_dense(torch.from_numpy(Q).cuda(), L)[0]); s, i = s.cpu().numpy(), i.cpu().numpy()
    for b in range(B):
        m = int(ec[b])
        assert np.array_equal(i[b, :m], ei[b, :m]) and np.array_equal(s[b, :m].view(np.uint32), es[b, :m].view(np.uint32)), ("dense", n, dim, B, L, b)
    if i8:
        Xu, Qu = CO.cosine_preprocess(X), CO.cosine_preprocess(Q)
        ix8 = eng.HxIndex(dim, ()); ix8.add(Xu)
        X8, rx = CO.quantize_i8(Xu); Q8, rq = CO.quantize_i8(Qu)
        es, ei, ec = CO.search_i8(X8, rx, Q8, rq, L)
        s, i = eng.unpack(ix8.search_i8(torch.from_numpy(Qu).cuda(), L)[0]); s, i = s.cpu().numpy(), i.cpu().numpy()
        for b in range(B):
            m = int(ec[b])
            assert np.array_equal(i[b, :m], ei[b, :m]) and np.array_equal(s[b, :m].view(np.uint32), es[b, :m].view(np.uint32)), ("i8", n, dim, B, L, b)
        ix8.close()
    print("ok", n, dim, B, L, ix.stats()["retry_queries"], ix.stats()["dense_fallback_queries"], flush=True)
    ix.close()

check(20000, 4096, 300, 10)
check(30000, 128, 4100, 10)
check(30000, 256, 200, 2048)
check(1, 768, 140, 10)
check(300, 1, 200, 5, i8=False)
check(260, 64, 257, 300)
print("extremes: all bit-exact")
