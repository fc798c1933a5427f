"""Diagnostic: sparse-stage time vs number of queries (occupancy / contention)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rag_application_amd import engine as eng, synth
N = 2_000_000
tabs = synth.tables()
ix = eng.HxIndex(768, (64,))
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
nseg = ix.stats()["n_segments"]
for B in (32, 64, 128, 256, 512, 768, 1024, 2048):
    qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
    t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
    for _ in range(2): ix.search_sparse(*t, 100)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): ix.search_sparse(*t, 100)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 3 * 1e3
    parts = max(1, min((512 + B - 1) // B, nseg, 8192 // 100))
    print(f"B={B:5d} parts={parts:3d} blocks={B*parts:5d} ms={ms:7.3f}  us per (block-visit)={ms*1e3/ (nseg/parts) / max(1,(B*parts+511)//512):6.2f}", flush=True)
