"""Diagnostic: per-rank step of the row-sharded H1 path at N-GPU shard size on ONE GPU, candidates-first exchange
against the per-shard exchange.  All `world` shards are REAL (built side by side on this GPU: 8 x 1.25M rows fit);
rank 0's work is what the timed loop runs, the other ranks' nominations / exact keys are computed once up front (the
batch is the same every step) and stand in for the collectives as device copies.  The lists are checked against the
per-shard exchange's.     python scripts/shard_cf.py [rows_per_shard] [world] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth  # noqa: E402
from rag_application_amd.distributed import ShardedIndex, H1Pipeline  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, dim, dl, sl, limit = (int(sys.argv[3]) if len(sys.argv) > 3 else 1024), 768, 100, 100, 10
tabs = synth.tables()
shards = []
for r in range(world):
    ix = eng.HxIndex(dim, (64, 128, 256), id_base=r * rows)
    ix.reserve(rows)
    ix.synth_fill(rows, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
    ix.finalize()
    shards.append(ix)
wmax = max(s.sparse_wmax()[0] for s in shards)
for s in shards:
    s.set_sparse_wmax(wmax)
Q = eng.synth_queries_dense(dim, 0, B, synth.SEED_QUERY)
qip, qix, qv = (torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs))
k1, k2, lp, k3, lout = eng.h1_plan(dl, sl, world)
if os.environ.get("K1"):
    k1 = int(os.environ["K1"])
if os.environ.get("K2"):
    k2 = int(os.environ["K2"])
if os.environ.get("K3"):
    k3 = int(os.environ["K3"])
PUB = B * (k1 + k2 + 2)
print(f"rows/shard {rows} world {world} batch {B}: k1 {k1} k2 {k2} lp {lp} k3 {k3} lout {lout}")

# ---- the other ranks' parts, once
noms = [s.h1_nominate_async(Q, qip, qix, qv, dl, sl, k1, k2, lout) for s in shards]
g_all = torch.cat([x[:PUB] for x in noms])
res = [s.h1_rescore_async(Q, qip, qix, qv, noms[r], g_all, world, r, dl, sl, k1, k2, lp, k3) for r, s in enumerate(shards)]
others_res = torch.stack(res[1:]).sum(dim=0) if world > 1 else torch.zeros_like(res[0])
red_all = torch.stack(res).sum(dim=0)
k_cf, c_cf, nf = eng.h1_finish(red_all, world, B, lp, k3, dl, sl, limit)
print("candidates first: failed queries", int(nf.item()))
if int(nf.item()):
    meta = red_all[B * (lp + world * k3 + world):].view(B, 4)
    fl = (meta[:, 2] // world).cpu()
    print("  flag words of the failed queries (1 dense ovf, 2 dense cut, 4 sparse flag, 8 sparse cut, 16 sparse list, 32 scale):",
          sorted(set(int(x) for x in fl if x)), "queries with flags:", int((fl != 0).sum()))
# the per-shard exchange on the same shards
allk = torch.cat([s.h1_local(Q, qip, qix, qv, dl, sl) for s in shards], dim=0)
k_ps, c_ps = eng.h1_fuse(allk, world, dl, sl, limit)
print("lists equal the per-shard exchange's:", bool(torch.equal(k_cf, k_ps) and torch.equal(c_cf, c_ps)))
others_nom = g_all[PUB:].clone()
others_loc = allk[B:].clone()


class Rank0(ShardedIndex):
    """rank 0 of `world`: the collectives are copies of what the other ranks were computed to send"""

    def __init__(self, local):
        super().__init__(local)
        self.world = world

    def reduce_sum(self, t):
        t.add_(others_res)
        return t

    def sync_sparse_scale(self):
        pass


def run(pipe, n):
    for _ in range(n):
        pipe.submit(Q, qip, qix, qv)
    pipe.wait()


def timeit(f, n=10):
    f()
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


sh = Rank0(shards[0])
MODES = (("per-shard exchange (h1_local_async + fuse)", False), ("candidates first", True))
if os.environ.get("CF_ONLY"):           # (profiling: only the candidates-first pipeline runs)
    MODES = MODES[1:]
for name, cf in MODES:
    pipe = H1Pipeline(sh, dl, sl, limit, force_side_stream=True, candidates_first=cf)
    if cf:
        pipe.k1, pipe.k2, pipe.k3 = k1, k2, k3
    if not cf:
        # the flag row travels with the lists: rebuild the gather stand-in for [B + 1] rows per rank
        sh.gather_raw = lambda keys: torch.cat([keys] + [torch.cat([others_loc[i * B:(i + 1) * B], keys[B:]]) for i in range(world - 1)])
    else:
        sh.gather_raw = lambda keys: torch.cat([keys.view(-1), others_nom]).view(world, -1)
    run(pipe, 3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(pipe, 20)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    out = pipe.submit(Q, qip, qix, qv)
    pipe.wait()
    print(f"pipelined step, {name:44s} ms {ms:.3f}  redone {pipe.redone}  lists ok "
          f"{bool(torch.equal(out[0], k_ps) and torch.equal(out[1], c_ps))}")
if os.environ.get("CF_ONLY"):
    sys.exit(0)
ix = shards[0]
print("  nominate                  ms", round(timeit(lambda: ix.h1_nominate_async(Q, qip, qix, qv, dl, sl, k1, k2, lout)), 3))
print("  rescore                   ms", round(timeit(lambda: ix.h1_rescore_async(Q, qip, qix, qv, noms[0], g_all, world, 0, dl, sl, k1, k2, lp, k3)), 3))
print("  finish                    ms", round(timeit(lambda: eng.h1_finish(red_all, world, B, lp, k3, dl, sl, limit)), 3))
print("  h1_local_async (old)      ms", round(timeit(lambda: ix.h1_local_async(Q, qip, qix, qv, dl, sl)), 3))
print("  local dense (old)         ms", round(timeit(lambda: ix.search_dense(Q, dl)), 3))
print("  local sparse (old)        ms", round(timeit(lambda: ix.search_sparse(qip, qix, qv, sl)), 3))
