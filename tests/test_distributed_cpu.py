"""CPU, world_size 2 over gloo: the row-sharded exchange logic of
rag_application_amd.distributed (shard ranges, id bases, gather layout, one exchange per
cascade level, RRF after the gather).  The per-shard stage kernels are replaced by an
oracle-backed stand-in here (the HIP ones need a GPU and are covered by -m gpu); the
result on every rank must equal the oracle on the UNSHARDED corpus, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P = dict(matryoshka_64_limit=60, matryoshka_128_limit=50, matryoshka_256_limit=40, dense_limit=30,
         quantized_limit=30, sparse_limit=25, final_limit=12, hnsw_ef=128)


def keys_of(lists, L):
    from oracle import oracle as O
    out = np.zeros((len(lists), L), np.uint64)
    cnt = np.zeros(len(lists), np.int32)
    for b, (s, i) in enumerate(lists):
        out[b, :len(i)] = O.order_key(s, i)
        cnt[b] = len(i)
    return torch.from_numpy(out.view(np.int64)), torch.from_numpy(cnt)


def unkey(k):
    k = np.asarray(k).view(np.uint64)
    ids = (np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64)
    u = (k >> np.uint64(32)).astype(np.uint32)
    u = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
    return u.view(np.float32), ids


class FakeShard:
    """Oracle-backed stand-in for engine.HxIndex over rows [r0, r1) with global ids."""

    def __init__(self, ora, r0):
        self.ora, self.r0 = ora, r0

    def _shift(self, s, i):
        return s, i + self.r0

    def search_dense(self, q, limit, prefix=0):
        return keys_of([self._shift(*self.ora.search_dense(x, limit, prefix)) for x in q.numpy()], limit)

    def search_i8(self, q, limit):
        return keys_of([self._shift(*self.ora.search_i8(x, limit)) for x in q.numpy()], limit)

    def search_sparse(self, qip, qix, qv, limit):
        qip, qix, qv = qip.numpy(), qix.numpy(), qv.numpy()
        return keys_of([self._shift(*self.ora.search_sparse(qix[qip[b]:qip[b + 1]], qv[qip[b]:qip[b + 1]], limit))
                        for b in range(len(qip) - 1)], limit)

    def rescore(self, q, cand_keys, cand_counts, limit, prefix=0):
        out = []
        for b, x in enumerate(q.numpy()):
            row = cand_keys[b].numpy()
            if cand_counts is not None:
                row = row[: int(cand_counts[b])]
            ids = unkey(row[row != 0])[1] - self.r0
            ids = ids[(ids >= 0) & (ids < self.ora.n)]
            out.append(self._shift(*self.ora.rescore(x, ids, limit, prefix)) if len(ids) else
                       (np.zeros(0, np.float32), np.zeros(0, np.int64)))
        return keys_of(out, limit)


class FakeShardDeferred(FakeShard):
    """... whose whole-collection stages take `flag` (engine.HxIndex with hx_*_async): the stage adds to flag[0] the
    number of queries whose lists are not final -- `bad_calls`: the search_i8 calls (by number) on which this
    stand-in pretends so, and hands out garbage"""
    deferred_stages = True
    bad_calls = ()
    n_i8 = 0

    def search_dense(self, q, limit, prefix=0, flag=None):
        return super().search_dense(q, limit, prefix)

    def search_sparse(self, qip, qix, qv, limit, flag=None):
        return super().search_sparse(qip, qix, qv, limit)

    def search_i8(self, q, limit, flag=None):
        k, c = super().search_i8(q, limit)
        if flag is not None:
            if self.n_i8 in self.bad_calls:
                flag[0] += 2
                k = torch.zeros_like(k)
            self.n_i8 += 1
        return k, c


class CpuOps:
    @staticmethod
    def merge(keys, counts, limit, dedupe):
        k = keys.numpy().view(np.uint64)
        out = np.zeros((k.shape[0], limit), np.uint64)
        cnt = np.zeros(k.shape[0], np.int32)
        for b in range(k.shape[0]):
            row = k[b] if counts is None else k[b, : int(counts[b])]
            row = row[row != 0]
            row = np.unique(row)[::-1] if dedupe else np.sort(row)[::-1]
            row = row[:limit]
            out[b, :len(row)] = row
            cnt[b] = len(row)
        return torch.from_numpy(out.view(np.int64)), torch.from_numpy(cnt)

    @staticmethod
    def rrf(a, ac, b, bc, limit, k, rank_base):
        from oracle import oracle as O
        lists = []
        for q in range(a.shape[0]):
            ia = unkey(a[q, : int(ac[q])].numpy())[1]
            ib = unkey(b[q, : int(bc[q])].numpy())[1]
            lists.append(O.rrf([ia, ib], limit=limit, k=k, rank_base=rank_base))
        return keys_of(lists, limit)


class FakeShardH1(FakeShard):
    """... with the packed local H1 stage (engine.HxIndex.h1_local)"""

    def h1_local(self, q, qip, qix, qv, dense_limit, sparse_limit):
        return torch.cat([self.search_dense(q, dense_limit)[0], self.search_sparse(qip, qix, qv, sparse_limit)[0]],
                         dim=1)


    # the call that does not read the flags: one more row, element 0 = the shard's flag word.  `bad_calls`: the
    # calls (by number) on which this stand-in pretends a stage flagged queries -- and hands out garbage lists,
    # as a shard whose lists are not final may
    bad_calls = ()
    n_async = 0

    def h1_local_async(self, q, qip, qix, qv, dense_limit, sparse_limit):
        keys = self.h1_local(q, qip, qix, qv, dense_limit, sparse_limit)
        row = torch.zeros((1, keys.shape[1]), dtype=keys.dtype)
        if self.n_async in self.bad_calls:
            keys = torch.zeros_like(keys)
            row[0, 0] = 3
        self.n_async += 1
        return torch.cat([keys, row], dim=0)


class CpuOpsH1(CpuOps):
    @staticmethod
    def h1_fuse(gathered, world, dense_limit, sparse_limit, limit, k, rank_base):
        """[world * B, dl + sl] rank-major -> global lists -> RRF (engine.h1_fuse / hx_h1_fuse)"""
        B = gathered.shape[0] // world
        g = gathered.view(world, B, dense_limit + sparse_limit)
        d = g[:, :, :dense_limit].permute(1, 0, 2).reshape(B, -1)
        sp = g[:, :, dense_limit:].permute(1, 0, 2).reshape(B, -1)
        dk, dc = CpuOps.merge(d, None, dense_limit, False)
        sk, sc = CpuOps.merge(sp, None, sparse_limit, False)
        return CpuOps.rrf(dk, dc, sk, sc, limit, k, rank_base)


def worker(rank, world, port, n, dim, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rag_application_amd.distributed import ShardedIndex
    tabs = O.synth_tables()
    r0, r1 = n * rank // world, n * (rank + 1) // world
    ora = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, r0, r1 - r0, tabs)
    ora.add(O.synth_dense(O.SEED_CORPUS, r0, r1 - r0, dim), ip, si, sv)
    ora.finalize()
    sh = ShardedIndex(FakeShard(ora, r0), ops=CpuOps)
    assert sh.world == world and sh.rank == rank
    Q = torch.from_numpy(O.synth_dense(O.SEED_QUERY, 0, B, dim))
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    tq = (torch.from_numpy(qip), torch.from_numpy(qsi.astype(np.int32)), torch.from_numpy(qsv))
    tk, tc = sh.hybrid_tree(Q, *tq, P)
    hk, hc = sh.hybrid_h1(Q, *tq, 40, 30, 10)
    dk, dc = sh.search_dense(Q, 15)
    # the tree with deferred flags (hx_*_async): rank 1 flags its SECOND batch, every rank redoes that batch and only it
    fd = FakeShardDeferred(ora, r0)
    if rank == 1:
        fd.bad_calls = (1,)
    shd = ShardedIndex(fd, ops=CpuOps)
    for _ in range(3):
        t2k, t2c = shd.hybrid_tree(Q, *tq, P)
        assert torch.equal(t2k, tk) and torch.equal(t2c, tc)
    assert shd.redone == 1
    # the same H1 query through the two calls that bracket the single exchange of the step
    h2k, h2c = ShardedIndex(FakeShardH1(ora, r0), ops=CpuOpsH1).hybrid_h1(Q, *tq, 40, 30, 10)
    assert torch.equal(h2k, hk) and torch.equal(h2c, hc)
    # ... and through the batch pipeline (no side stream without a HIP device: it degenerates to the call above)
    from rag_application_amd.distributed import H1Pipeline
    fake = FakeShardH1(ora, r0)
    if rank == 1:
        fake.bad_calls = (1,)              # ONE rank flags its second batch: every rank must redo that batch
    pipe = H1Pipeline(ShardedIndex(fake, ops=CpuOpsH1), 40, 30, 10)
    assert pipe.deferred and pipe.side is None
    ps = [pipe.submit(Q, *tq) for _ in range(4)]
    pipe.wait()
    assert pipe.redone == 1 and not pipe.pending
    for pk, pc in ps:
        assert torch.equal(pk, hk) and torch.equal(pc, hc)
    ret[rank] = (tk.numpy(), tc.numpy(), hk.numpy(), hc.numpy(), dk.numpy(), dc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_tree_equals_unsharded_oracle():
    from oracle import oracle as O
    n, dim, B, world = 1500, 256, 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, n, dim, B, ret), nprocs=world, join=True)
    tabs = O.synth_tables()
    full = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
    full.add(O.synth_dense(O.SEED_CORPUS, 0, n, dim), ip, si, sv)
    full.finalize()
    Q = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    for rank in range(world):
        tk, tc, hk, hc, dk, dc = ret[rank]
        for b in range(B):
            sp = (qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]])
            for (k, c), (es, ei) in (((tk, tc), O.hybrid_tree(full, Q[b], *sp, P)),
                                     ((hk, hc), O.hybrid_h1(full, Q[b], *sp, 40, 30, 10)),
                                     ((dk, dc), full.search_dense(Q[b], 15))):
                m = len(ei)
                assert int(c[b]) == m
                s, i = unkey(k[b, :m])
                np.testing.assert_array_equal(i, ei)
                np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))


# ---------------------------------------------------------------------------- the candidates-first exchange
class FakeShardCF(FakeShardH1):
    """... with the two calls that bracket the candidates-first exchange (engine.HxIndex.h1_nominate_async /
    h1_rescore_async).  The packing is this stand-in's own -- H1Pipeline moves the tensors, it never looks inside:
        nomination  [B * k1 dense keys | B * k2 sparse keys | B * 2 words: how many keys the shard HAS above its share]
        exact keys  [B * dl | B * sl | B flag words]: slot j of a list = the j-th best key of the gathered nominations if
                    this shard owns its row, else 0 (the integer sum over the ranks fills every slot exactly once)
    A query is flagged when a shard's share was full and its last key still lies above the global cut: rows of that shard
    beyond its share could belong to the list (engine: shardx.hip k_h1x_cuts, flags 2 / 8)."""
    n_nom = 0

    def h1_nominate_async(self, q, qip, qix, qv, dl, sl, k1, k2, lout):
        B = q.shape[0]
        d = self.search_dense(q, min(k1, dl))[0]
        sp = self.search_sparse(qip, qix, qv, min(k2, sl))[0]
        pad = lambda t, k: torch.cat([t, torch.zeros((B, k - t.shape[1]), dtype=t.dtype)], dim=1)
        more = torch.zeros((B, 2), dtype=torch.int64)
        more[:, 0] = (self.search_dense(q, dl)[1] > k1).long()
        more[:, 1] = (self.search_sparse(qip, qix, qv, sl)[1] > k2).long()
        self.n_nom += 1
        return torch.cat([pad(d, k1).reshape(-1), pad(sp, k2).reshape(-1), more.reshape(-1)])

    def h1_rescore_async(self, q, qip, qix, qv, nom, gathered, world, rank, dl, sl, k1, k2, lp, k3):
        B = q.shape[0]
        g = gathered.view(world, -1)
        gd = g[:, :B * k1].reshape(world, B, k1).numpy().view(np.uint64)
        gs = g[:, B * k1:B * (k1 + k2)].reshape(world, B, k2).numpy().view(np.uint64)
        more = g[:, B * (k1 + k2):].reshape(world, B, 2).numpy()
        res = np.zeros((B, dl + sl + 1), np.uint64)
        n_loc = self.ora.n
        for b in range(B):
            for col, (gk, L, kk, off) in enumerate(((gd, dl, k1, 0), (gs, sl, k2, dl))):
                allk = np.sort(gk[:, b, :].reshape(-1))[::-1]
                allk = allk[allk != 0][:L]
                cut = allk[-1] if len(allk) == L else np.uint64(0)
                ids = unkey(allk)[1] - self.r0
                own = (ids >= 0) & (ids < n_loc)
                res[b, off:off + len(allk)][own] = allk[own]
                # a shard that HAS more keys than its share, and whose last sent key still makes the list: cut short
                for r in range(world):
                    if more[r, b, col] and (gk[r, b, kk - 1] >= cut):
                        res[b, dl + sl] = 1
        return torch.from_numpy(res.view(np.int64).reshape(-1).copy())


class CpuOpsCF(CpuOpsH1):
    plan = (8, 8)

    @classmethod
    def h1_plan(cls, dl, sl, world):
        return cls.plan[0], cls.plan[1], dl, sl, 0          # k1, k2, lp, k3, lout

    @staticmethod
    def h1_finish(res, world, B, lp, k3, dl, sl, limit, k, rank_base):
        r = res.view(B, dl + sl + 1)
        flags = r[:, dl + sl] // world                      # every rank adds the same word
        d, sp = r[:, :dl].contiguous(), r[:, dl:dl + sl].contiguous()
        dc = (d != 0).sum(dim=1).to(torch.int32)
        sc = (sp != 0).sum(dim=1).to(torch.int32)
        keys, cnt = CpuOps.rrf(d, dc, sp, sc, limit, k, rank_base)
        return keys, cnt, torch.tensor([int((flags != 0).sum())])


def cf_worker(rank, world, port, n, dim, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rag_application_amd.distributed import ShardedIndex, H1Pipeline
    tabs = O.synth_tables()
    r0, r1 = n * rank // world, n * (rank + 1) // world
    ora = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, r0, r1 - r0, tabs)
    ora.add(O.synth_dense(O.SEED_CORPUS, r0, r1 - r0, dim), ip, si, sv)
    ora.finalize()
    Q = torch.from_numpy(O.synth_dense(O.SEED_QUERY, 0, B, dim))
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    tq = (torch.from_numpy(qip), torch.from_numpy(qsi.astype(np.int32)), torch.from_numpy(qsv))
    fake = FakeShardCF(ora, r0)
    pipe = H1Pipeline(ShardedIndex(fake, ops=CpuOpsCF), 40, 30, 10)
    assert pipe.cf and pipe.deferred and pipe.side is None
    assert (pipe.k1, pipe.k2, pipe.k1max, pipe.k2max) == (8, 8, 64, 32)
    # shares of 8 cannot hold a top-40 / top-30 list spread over two shards: the first batches are flagged (by EVERY rank,
    # from the same gathered words), redone through the per-shard exchange and the shares doubled -- until they hold.
    # Batches are verified `depth` submits late: one that went out with shares since widened must not widen them again
    outs, ks = [], []
    NB = 12
    for _ in range(NB):
        outs.append(pipe.submit(Q, *tq))
        ks.append((pipe.k1, pipe.k2))
    pipe.wait()
    assert pipe.cf, "the candidates-first exchange must survive widened shares"
    assert 1 <= pipe.redone < NB, pipe.redone
    assert 8 < pipe.k1 <= 32 and 8 < pipe.k2 <= 32, (pipe.k1, pipe.k2)   # (two doublings suffice: late verdicts did not add more)
    assert fake.n_nom == NB and fake.n_async == 0            # every batch went out candidates first; redone ones via h1_local
    ret[rank] = ([(k.numpy(), c.numpy()) for k, c in outs], pipe.redone, ks)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_candidates_first_pipeline_equals_unsharded_oracle():
    """H1Pipeline's candidates-first exchange (all-gather of the nominations, integer-sum all-reduce of the exact keys)
    over gloo with two ranks: every batch -- served by the exchange or flagged and redone -- equals the oracle's H1 list
    on the unsharded corpus, and both ranks take the same decisions."""
    from oracle import oracle as O
    n, dim, B, world = 1500, 256, 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(cf_worker, args=(world, port, n, dim, B, ret), nprocs=world, join=True)
    tabs = O.synth_tables()
    full = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
    full.add(O.synth_dense(O.SEED_CORPUS, 0, n, dim), ip, si, sv)
    full.finalize()
    Q = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    want = [O.hybrid_h1(full, Q[b], qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]], 40, 30, 10) for b in range(B)]
    assert ret[0][1] == ret[1][1] and ret[0][2] == ret[1][2]       # the same batches redone, the same shares after each
    for rank in range(world):
        for k, c in ret[rank][0]:
            for b, (es, ei) in enumerate(want):
                m = len(ei)
                assert int(c[b]) == m
                s, i = unkey(k[b, :m])
                np.testing.assert_array_equal(i, ei)
                np.testing.assert_array_equal(s.view(np.uint32), es.view(np.uint32))


# ---------------------------------------------------------------------------- the sharded front end (C2 + ingest)
class GrowingShard(FakeShard):
    """engine.HxIndex stand-in that also ingests: an oracle index over this rank's rows; the global id of a local
    row is what `set_next_id` named when its block arrived (hx_set_next_id), `truncate` rolls a block back
    (hx_truncate).  `fail_adds` / `fail_searches`: the add / search_dense calls (by number) that raise, `hang`:
    seconds the next search_dense sleeps -- a shard in trouble."""
    fail_adds = ()
    fail_searches = ()
    hang = 0.0

    def __init__(self, dim, msizes, id_base):
        from oracle import oracle as O
        super().__init__(O.OracleIndex(dim, msizes), id_base)
        self.gids = np.zeros(0, np.int64)
        self.next = None
        self.n_add = self.n_search = 0

    def set_next_id(self, first):
        assert self.gids.size == 0 or first > self.gids[-1]
        self.next = int(first)

    def add(self, dense, ip=None, ix=None, v=None):
        k, self.n_add = self.n_add, self.n_add + 1
        first, self.next = self.next, None
        if k in self.fail_adds:
            raise RuntimeError("shard refuses the block")
        n = len(dense)
        first = first if first is not None else (int(self.gids[-1]) + 1 if self.gids.size else self.r0)
        self.ora.add(dense, ip, ix, v)
        self.gids = np.concatenate([self.gids, first + np.arange(n, dtype=np.int64)])

    def truncate(self, n):
        o = self.ora
        o.raw = o.raw[:n]
        o.sp_idx, o.sp_val = o.sp_idx[:o.sp_indptr[n]], o.sp_val[:o.sp_indptr[n]]
        o.sp_indptr = o.sp_indptr[:n + 1]
        o._final = False
        self.gids = self.gids[:n]

    def count(self):
        return self.ora.n

    def save(self, path):
        o = self.ora
        np.savez(path + ".npz", raw=o.raw, ip=o.sp_indptr, ix=o.sp_idx, v=o.sp_val, gids=self.gids,
                 msizes=np.asarray(o.msizes))

    @classmethod
    def load(cls, path):
        z = np.load(path + ".npz")
        self = cls(z["raw"].shape[1], tuple(int(m) for m in z["msizes"]), 0)
        self.ora.add(z["raw"], z["ip"], z["ix"], z["v"])
        self.gids = z["gids"]
        return self

    def _shift(self, s, i):
        return s, self.gids[i]

    def search_dense(self, q, limit, prefix=0):
        k, self.n_search = self.n_search, self.n_search + 1
        if k in self.fail_searches:
            raise RuntimeError("shard cannot search")
        if self.hang:
            import time
            time.sleep(self.hang)
        return super().search_dense(q, limit, prefix)

    def rescore(self, q, cand_keys, cand_counts, limit, prefix=0):
        out = []
        for b, x in enumerate(q.numpy()):
            row = cand_keys[b].numpy()
            if cand_counts is not None:
                row = row[: int(cand_counts[b])]
            g = unkey(row[row != 0])[1]
            loc = np.searchsorted(self.gids, g)
            loc = loc[(loc < self.gids.size) & (self.gids[np.minimum(loc, max(self.gids.size - 1, 0))] == g)] \
                if self.gids.size else loc[:0]
            out.append(self._shift(*self.ora.rescore(x, loc, limit, prefix)) if len(loc) else
                       (np.zeros(0, np.float32), np.zeros(0, np.int64)))
        return keys_of(out, limit)


def _chunks(O, tabs, n, dim):
    X = O.synth_dense(O.SEED_CORPUS, 0, n, dim)
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
    return [{"content": f"chunk {r}", "dense_embedding": X[r].tolist(),
             "sparse_embedding": {"indices": si[ip[r]:ip[r + 1]].tolist(), "values": sv[ip[r]:ip[r + 1]].tolist()},
             "chunk_metadata": {"document_id": f"d{r % 5}", "user_id": "u", "file_name": f"f{r % 3}.txt", "mime_type": "text/plain",
                                "file_size": 1, "description": "", "file_path": "/p", "context_version": 1,
                                "chunk_number": r, "doc_summary": "s"}} for r in range(n)]


def front_worker(rank, world, port, n, dim, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import asyncio
    import datetime
    from oracle import oracle as O
    from rag_application_amd.sharded import ShardedCollection, ShardedHandler, ShardError, bcast_queries
    tabs = O.synth_tables()
    # -- C2 alone: the packed broadcast delivers the front rank's batch bit for bit
    Q = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    got = bcast_queries(*((Q, qip, qsi.astype(np.int32), qsv) if rank == 0 else (None,) * 4))
    # ... and with the header on a host-side group of its own (what bench.py does beside RCCL)
    got2 = bcast_queries(*((Q, qip, qsi.astype(np.int32), qsv) if rank == 0 else (None,) * 4),
                         header_group=dist.new_group(backend="gloo"))
    assert all(torch.equal(a, b) for a, b in zip(got, got2))
    assert np.array_equal(got[0].numpy(), Q) and np.array_equal(got[1].numpy(), qip)
    assert np.array_equal(got[2].numpy(), qsi.astype(np.int32)) and np.array_equal(got[3].numpy(), qsv)
    # -- the handler: rank 0 is the application, the others serve
    shards = []

    def factory(d, ms, base):
        shards.append(GrowingShard(d, ms, base))
        return shards[-1]

    class Rerank:                                   # rerank hook (qdrant_handler.py:380): reverses the list
        def rerank_documents(self, query, documents, max_tokens):
            return list(range(len(documents)))[::-1]

    import tempfile
    pdir = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(pdir, 0)
    h = ShardedHandler(index_factory=factory, ops=CpuOpsH1, dense_vector_size=dim, reranker=Rerank(), timeout=60,
                       persist_dir=pdir[0], index_loader=GrowingShard.load)
    if rank != 0:
        h.serve()
    else:
        chunks = _chunks(O, tabs, n, dim)
        run = asyncio.run
        try:
            run(h.store_document_vectors([dict(chunks[0], dense_embedding=[0.0] * 7)], "u"))
            raise AssertionError("dimension mismatch must raise")
        except ValueError:
            pass
        bad = dict(chunks[0], sparse_embedding={"indices": [5, 5], "values": [1.0, 2.0]})
        try:                                      # refused on the front rank: the ranks never hear of the batch
            run(h.store_document_vectors([bad], "u"))
            raise AssertionError("duplicate sparse index must raise")
        except ValueError:
            pass
        c1, c2 = n // 2 + 1, n // 2 + n // 7 + 2  # three uneven batches: blocks of different sizes on every rank
        run(h.store_document_vectors(chunks[:c1], "u"))
        run(h.store_document_vectors(chunks[c1:c2], "u"))
        run(h.store_document_vectors(chunks[c2:], "u"))
        assert run(h.get_collection_chunk_count("u")) == n and run(h.get_all_containers()) == ["u"]
        assert run(h.get_collection_chunk_count("u", filters={"must": [{"key": "file_name", "match": {"value": "f1.txt"}}]})) \
            == len([r for r in range(n) if r % 3 == 1])
        sp = [{"indices": qsi[qip[b]:qip[b + 1]].tolist()[::-1], "values": qsv[qip[b]:qip[b + 1]].tolist()[::-1]}
              for b in range(B)]                  # reversed term order: the front end sorts by term id
        tree = run(h.hybrid_search_batch("u", Q.tolist(), sp, top_k=12, search_params=P))
        h1 = run(h.hybrid_search_batch("u", Q.tolist(), sp, top_k=10, search_params=P, mode="h1"))
        one = run(h.hybrid_search("u", "text", Q[0].tolist(), sp[0], top_k=3, search_params=P))   # rerank hook: reversed
        assert run(h.hybrid_search("u", "text", Q[0].tolist(), sp[0], search_params=None)) == []
        flt = {"must": [{"key": "file_name", "match": {"value": "f1.txt"}}]}
        filt = run(h.hybrid_search_batch("u", Q.tolist(), sp, top_k=12, search_params=P, filters=flt))
        ret["tree"] = [[(p.payload["chunk_number"], p.score, p.payload["content"], sorted(p.payload)) for p in row] for row in tree]
        ret["h1"] = [[(p.payload["chunk_number"], p.score) for p in row] for row in h1]
        ret["one"] = [(p.payload["chunk_number"], p.score) for p in one]
        ret["filt"] = [[(p.payload["chunk_number"], p.score, p.payload["file_name"]) for p in row] for row in filt]
        # persist_dir: every rank writes its shard, the front rank the ids + payloads; a dropped collection comes back
        run(h.save_collection("u"))
        run(h.delete_collection("u"))
        assert run(h.get_collection_chunk_count("u")) == 0
        run(h.create_collection("u"))
        assert run(h.get_collection_chunk_count("u")) == n
        again = run(h.hybrid_search_batch("u", Q.tolist(), sp, top_k=12, search_params=P))
        assert [[(p.id, p.score, p.payload["content"]) for p in row] for row in again] == \
            [[(p.id, p.score, p.payload["content"]) for p in row] for row in tree]
        # chat vectors (qdrant_handler.py:200-267) into a collection of their own
        chats = [{"dense_embedding": c["dense_embedding"], "sparse_embedding": c["sparse_embedding"], "chat_id": f"c{r}",
                  "message_type": "user", "timestamp": "2025-01-01T00:00:00", "entities": [], "relationships": [],
                  "chat_summary": "", "message": f"msg {r}"} for r, c in enumerate(chunks[:40])]
        run(h.store_chat_vectors(chats[:13], "chat"))
        run(h.store_chat_vectors(chats[13:], "chat"))
        ch = run(h.hybrid_search_batch("chat", Q[:2].tolist(), sp[:2], top_k=5, search_params=P))
        ret["chat"] = [[(p.payload["content"], p.payload["is_chat"], p.score) for p in row] for row in ch]
        # -- a shard that fails: store rolls back everywhere and raises; search returns []
        run(h.create_collection("w", dense_vector_size=dim))
        run(h.store_document_vectors(chunks[:50], "w"))
        try:
            run(h.store_document_vectors(chunks[50:90], "w"))      # the worker's shard refuses its block (its 2nd add)
            raise AssertionError("a failed shard must make the store raise")
        except ShardError as e:
            assert "rank 1" in str(e)
        assert run(h.get_collection_chunk_count("w")) == 50 and shards[-1].count() == 25   # this rank rolled back
        run(h.store_document_vectors(chunks[50:90], "w"))          # ... and the same batch goes in afterwards
        hw = run(h.hybrid_search_batch("w", Q.tolist(), sp, top_k=10, search_params=P, mode="h1"))
        ret["h1w"] = [[(p.payload["chunk_number"], p.score) for p in row] for row in hw]
        assert run(h.hybrid_search("w", "text", Q[0].tolist(), sp[0], search_params=P)) == []   # worker's search raises
        assert len(run(h.hybrid_search("w", "text", Q[0].tolist(), sp[0], search_params=P))) == 10   # ... once
        try:
            run(h.delete_collection("nobody"))
            raise AssertionError("unknown collection must raise")
        except KeyError:
            pass
        run(h.delete_collection("w"))
        run(h.delete_collection("chat"))
        run(h.delete_collection("u"))
        assert run(h.get_collection_chunk_count("u")) == 0
        h.shutdown()
    if rank == 1:                                   # the failures the front rank met above were planned here
        pass
    # -- ingest of TEXT chunks with a per-rank encoder replica (BASELINE config 5's shape): every rank encodes its block
    class Enc:
        def encode(self, texts):
            return np.stack([O.synth_dense(900 + rank * 0, int(t.split()[1]), 1, dim)[0] for t in texts]) if texts else \
                np.zeros((0, dim), np.float32)
    col = ShardedCollection(dim, (64, 128, 256), index_factory=GrowingShard, ops=CpuOpsH1)
    texts = [f"row {r}" for r in range(101)]
    seq = col.store(texts=texts[:40] if rank == 0 else None, encoder=Enc())
    seq2 = col.store(texts=texts[40:] if rank == 0 else None, encoder=Enc())
    assert col.count() == 101 and col.counts.tolist() == [50, 51]
    k, c = col.search(*((Q[:2], np.zeros(3, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32),
                         dict(P, sparse_limit=5), "h1") if rank == 0 else (None,) * 6))
    if rank == 0:
        assert seq.tolist() == list(range(40)) and seq2.tolist() == list(range(40, 101))
        ret["texts"] = col.resolve(k, c)
    dist.barrier()
    dist.destroy_process_group()


# the worker rank's planned failures: collection "w" is the fourth shard it makes ("u", "u" loaded again, "chat", "w")
_orig_init = GrowingShard.__init__


def _planned_init(self, dim, msizes, id_base):
    _orig_init(self, dim, msizes, id_base)
    GrowingShard._made = getattr(GrowingShard, "_made", 0) + 1
    if dist.is_initialized() and dist.get_rank() == 1 and GrowingShard._made == 4:
        self.fail_adds = (1,)          # its second block
        self.fail_searches = (1,)      # one search_dense call per query batch: the batch after the H1 batch


GrowingShard.__init__ = _planned_init


@pytest.mark.timeout(600)
def test_sharded_front_end_from_rank0_equals_unsharded_oracle():
    """ShardedHandler on two ranks (gloo): rank 0 ingests THREE uneven batches (dealt in contiguous blocks, ids named
    per block) and queries (ONE packed broadcast per batch); rank 1 serves.  Payloads, ids and score bits of the
    reference tree and of H1 equal the oracle on the unsharded corpus in insertion order -- ties included.  Chat
    vectors, a payload filter, the rerank hook, and a shard that fails (store rolls back + raises, search -> [])."""
    from oracle import oracle as O
    from rag_application_amd import filters as F
    n, dim, B, world = 900, 256, 4, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(front_worker, args=(world, port, n, dim, B, ret), nprocs=world, join=True)
    tabs = O.synth_tables()
    full = O.OracleIndex(dim, (64, 128, 256))
    ip, si, sv = O.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
    X = O.synth_dense(O.SEED_CORPUS, 0, n, dim)
    full.add(X, ip, si, sv)
    full.finalize()
    Q = O.synth_dense(O.SEED_QUERY, 0, B, dim)
    qip, qsi, qsv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, B, tabs)
    REF_PAYLOAD = sorted(["document_id", "user_id", "file_name", "mime_type", "file_size", "file_description", "file_path",
                          "context_version", "chunk_number", "entities", "relationships", "context", "document_summary",
                          "content", "page_number", "languages", "element_id", "is_continuation", "category"])
    for b in range(B):
        sp = (qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]])
        es, ei = O.hybrid_tree(full, Q[b], *sp, P)
        assert [t[0] for t in ret["tree"][b]] == ei.tolist() and [t[2] for t in ret["tree"][b]] == [f"chunk {i}" for i in ei]
        assert all(t[3] == REF_PAYLOAD for t in ret["tree"][b])         # the reference's 19 fields (qdrant_handler.py:165-185)
        np.testing.assert_array_equal(np.array([t[1] for t in ret["tree"][b]], np.float32).view(np.uint32), es.view(np.uint32))
        # H1 over three batches: ids AND score bits, no carve-out for ties
        es, ei = O.hybrid_h1(full, Q[b], *sp, P["dense_limit"], P["sparse_limit"], P["final_limit"])
        assert [t[0] for t in ret["h1"][b]] == ei[:10].tolist()
        np.testing.assert_array_equal(np.array([t[1] for t in ret["h1"][b]], np.float32).view(np.uint32), es[:10].view(np.uint32))
        # the filter belongs to the root query (:297, :371): the re-scored union, filtered, cut to final_limit
        us, ui = O.hybrid_tree(full, Q[b], *sp, dict(P, final_limit=P["dense_limit"] + 10))
        keep = [(int(i), s) for s, i in zip(us, ui) if i % 3 == 1][:P["final_limit"]]
        assert [t[0] for t in ret["filt"][b]] == [k[0] for k in keep] and all(t[2] == "f1.txt" for t in ret["filt"][b])
    es, ei = O.hybrid_tree(full, Q[0], qsi[qip[0]:qip[1]], qsv[qip[0]:qip[1]], P)
    assert [t[0] for t in ret["one"]] == ei[::-1][:3].tolist()          # the rerank hook reversed the list, then [:top_k]
    # chat vectors: 40 rows in two batches, the chat payload
    chat = O.OracleIndex(dim, (64, 128, 256))
    chat.add(X[:40], ip[:41], si[:ip[40]], sv[:ip[40]])
    for b in range(2):
        es, ei = O.hybrid_tree(chat, Q[b], qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]], P)
        assert [t[0] for t in ret["chat"][b]] == [f"msg {i}" for i in ei[:5]] and all(t[1] is True for t in ret["chat"][b])
    # collection "w": the batch that failed once is in exactly once
    w = O.OracleIndex(dim, (64, 128, 256))
    w.add(X[:90], ip[:91], si[:ip[90]], sv[:ip[90]])
    for b in range(B):
        es, ei = O.hybrid_h1(w, Q[b], qsi[qip[b]:qip[b + 1]], qsv[qip[b]:qip[b + 1]], P["dense_limit"], P["sparse_limit"], P["final_limit"])
        assert [t[0] for t in ret["h1w"][b]] == ei[:10].tolist()
    # the text ingest: row r was encoded (on whichever rank got it) as synth_dense(900, r): dense-only H1
    enc = O.OracleIndex(dim, (64, 128, 256))
    enc.add(np.stack([O.synth_dense(900, r, 1, dim)[0] for r in range(101)]))
    for b in range(2):
        es, ei = O.rrf([enc.search_dense(Q[b], P["dense_limit"])[1], np.zeros(0, np.int64)], limit=P["final_limit"])
        assert [t[0] for t in ret["texts"][b]] == ei.tolist()


def hang_worker(rank, world, port, dim, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import asyncio
    import time
    from oracle import oracle as O
    from rag_application_amd.sharded import ShardedHandler, ShardError
    tabs = O.synth_tables()
    shards = []

    def factory(d, ms, base):
        shards.append(GrowingShard(d, ms, base))
        if rank == 1:
            shards[-1].hang = 12.0             # the worker's shard stops answering
        return shards[-1]

    h = ShardedHandler(index_factory=factory, ops=CpuOpsH1, dense_vector_size=dim, timeout=3)
    if rank != 0:
        h.serve()                              # ends when the group is found broken
    else:
        run = asyncio.run
        chunks = _chunks(O, tabs, 60, dim)
        run(h.store_document_vectors(chunks, "u"))
        Q = O.synth_dense(O.SEED_QUERY, 0, 1, dim)
        t0 = time.time()
        out = run(h.hybrid_search("u", "t", Q[0].tolist(), {"indices": [1], "values": [1.0]}, search_params=P))
        ret["search"] = (out, time.time() - t0)
        t0 = time.time()
        try:
            run(h.store_document_vectors(chunks, "u"))
            ret["store"] = "stored"
        except ShardError as e:
            ret["store"] = ("raised", time.time() - t0)
    # no barrier: the group is broken by design; each rank leaves on its own
    time.sleep(1.0)


@pytest.mark.timeout(300)
def test_sharded_handler_times_out_on_a_hanging_rank():
    """A worker whose shard stops answering: the front rank's search returns [] and its next mutation raises, both
    within the handler's timeout -- it is never left inside a collective (qdrant_handler.py:384-386, :196-198)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(hang_worker, args=(2, port, 128, ret), nprocs=2, join=True)
    out, dt = ret["search"]
    assert out == [] and dt < 10.0
    assert ret["store"][0] == "raised" and ret["store"][1] < 10.0


def deal_fail_worker(rank, world, port, dim, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import asyncio
    import time
    from oracle import oracle as O
    from rag_application_amd import sharded as SH
    tabs = O.synth_tables()
    h = SH.ShardedHandler(index_factory=lambda d, ms, base: GrowingShard(d, ms, base), ops=CpuOpsH1, dense_vector_size=dim,
                          timeout=3)
    if rank != 0:
        real = SH.ShardedCollection._deal_arrays
        calls = {"n": 0}

        def failing(self, *a, **kw):               # the worker dies inside the deal of the SECOND batch
            calls["n"] += 1
            if calls["n"] == 2:
                raise RuntimeError("worker lost inside the deal")
            return real(self, *a, **kw)
        SH.ShardedCollection._deal_arrays = failing
        h.serve()                                  # must END: an out-of-step worker leaves the loop
        ret["worker_broken"] = bool(h.broken)
    else:
        run = asyncio.run
        chunks = _chunks(O, tabs, 60, dim)
        run(h.store_document_vectors(chunks, "u"))
        t0 = time.time()
        try:
            run(h.store_document_vectors(chunks, "u"))
            ret["store"] = "stored"
        except SH.ShardError:
            ret["store"] = ("raised", time.time() - t0)
        ret["front_broken"] = bool(h.broken)
        Q = O.synth_dense(O.SEED_QUERY, 0, 1, dim)
        t0 = time.time()
        out = run(h.hybrid_search("u", "t", Q[0].tolist(), {"indices": [1], "values": [1.0]}, search_params=P))
        ret["search"] = (out, time.time() - t0)
    time.sleep(1.0)


@pytest.mark.timeout(300)
def test_sharded_handler_closes_when_a_rank_fails_inside_the_deal():
    """A rank that fails INSIDE the point-to-point deal of a batch leaves the ranks out of step (round 3: the front
    rank kept issuing commands on the control group).  Now the store raises within the timeout, the handler is closed
    on the front rank (`broken`), the worker leaves its serve() loop, and a later search returns [] at once instead of
    waiting on a group nobody answers on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(deal_fail_worker, args=(2, port, 128, ret), nprocs=2, join=True)
    assert ret["store"][0] == "raised" and ret["store"][1] < 10.0
    assert ret["front_broken"] and ret["worker_broken"]
    out, dt = ret["search"]
    assert out == [] and dt < 1.0
