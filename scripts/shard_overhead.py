"""Diagnostic: per-rank step time of the row-sharded H1 path at N-GPU shard size through the PER-SHARD exchange (rounds 2-3),
with the exchange replaced by a local stand-in (the lists repeated `world` times), i.e. everything but the wire.
The candidates-first exchange of round 4 is measured by scripts/shard_cf.py (real shards)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
from rag_application_amd.distributed import ShardedIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, dim = (int(sys.argv[3]) if len(sys.argv) > 3 else 1024), 768      # argv: rows world [batch]
tabs = synth.tables()
ix = eng.HxIndex(dim, (64, 128, 256)); ix.reserve(rows); ix.synth_fill(rows, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
Q = eng.synth_queries_dense(dim, 0, B, synth.SEED_QUERY)
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
qip, qix, qv = (torch.from_numpy(a).cuda() for a in (qip, qix, qv))
sh = ShardedIndex(ix)
sh.world = world
def fake_gather(keys):
    Bq, L = keys.shape
    out = keys.contiguous().repeat(world, 1)           # stands in for all_gather_into_tensor
    return out.view(world, Bq, L).permute(1, 0, 2).reshape(Bq, -1)
sh.gather = fake_gather
sh.gather_raw = lambda keys: keys.contiguous().repeat(world, 1)
hp = eng.make_params(dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
                          quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128), mode=eng.HX_MODE_H1)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("rows", rows, "world", world, "batch", B)
print("one ABI call (N=1 path)     ms", round(timeit(lambda: ix.hybrid_query(Q, qip, qix, qv, hp)), 3))
print("sharded path, no wire       ms", round(timeit(lambda: sh.hybrid_h1(Q, qip, qix, qv, 100, 100, 10)), 3))
# the pipeline: local stage without a flag read (h1_local_async), exchange stand-in + fusion on a side stream,
# flags verified two batches later -- what bench.py runs at N > 1
from rag_application_amd.distributed import H1Pipeline
sh.gather_raw = lambda keys: keys.contiguous().repeat(world, 1)
# (this stand-in REPEATS rank 0's lists for the other ranks: fine for the per-shard exchange, meaningless for the
# candidates-first one, whose ranks must own disjoint rows -- scripts/shard_cf.py builds real shards for that)
pipe = H1Pipeline(sh, 100, 100, 10, candidates_first=False)
def piped(n=10):
    for _ in range(n): pipe.submit(Q, qip, qix, qv)
    pipe.wait()
piped(3); torch.cuda.synchronize(); t = time.perf_counter(); piped(20); torch.cuda.synchronize()
print("pipelined, flags deferred   ms", round((time.perf_counter() - t) / 20 * 1e3, 3), "redone", pipe.redone)
print("  local dense               ms", round(timeit(lambda: ix.search_dense(Q, 100)), 3))
print("  local sparse              ms", round(timeit(lambda: ix.search_sparse(qip, qix, qv, 100)), 3))
dk, dc = ix.search_dense(Q, 100); sk, sc = ix.search_sparse(qip, qix, qv, 100)
print("  cat+gather stand-in       ms", round(timeit(lambda: fake_gather(torch.cat([dk, sk], dim=1))), 3))
allk = fake_gather(torch.cat([dk, sk], dim=1)).reshape(B, world, -1)
print("  2 merges                  ms", round(timeit(lambda: (eng.merge(allk[:, :, :100].reshape(B, -1), None, 100, False), eng.merge(allk[:, :, 100:].reshape(B, -1), None, 100, False))), 3))
print("  rrf                       ms", round(timeit(lambda: eng.rrf(dk, dc, sk, sc, 10, 2.0, 0)), 3))
