"""Row-sharded search across the GPUs of one node (one process per GPU, RCCL over xGMI).

Every ranking stage of the reference query (qdrant_handler.py:305-372) is "top-L of a
score over all documents", and top-L(union of shards) is a subset of the union of the
shards' top-L lists.  So each stage runs on the local shard, the per-shard lists (64-bit
keys: score + GLOBAL row id) are all-gathered, and every rank merges them into the same
global list.  Nested stages then re-score only their OWN rows among the global survivors
and exchange again; RRF runs after the gather because its ranks are global.  Payloads
are tiny (B x L x 8 bytes per rank), i.e. latency-bound: one all-gather per stage, no
ring of its own.  The reference has no multi-device path at all."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class ShardedIndex:
    """`local` is this rank's shard: anything with the stage methods of
    `engine.HxIndex` (search_dense / search_i8 / search_sparse / rescore).  `ops`
    provides merge(keys, counts, limit, dedupe) and rrf(a, ac, b, bc, limit, k,
    rank_base): the HIP implementations of `engine` by default (tests inject a CPU
    checker to exercise the exchange logic over gloo)."""

    def __init__(self, local, group: Optional[dist.ProcessGroup] = None, ops=None):
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if ops is None:
            from . import engine as ops  # HIP kernels
        self.ops = ops
        # A local stage that raises must not leave the other ranks alone inside the stage's all-gather: the error is
        # kept, this rank goes on with EMPTY lists of the right shape (so every collective of the query still
        # happens, on every rank, in the same order), and the caller exchanges `take_error()` at the end of the
        # call -- a failed rank makes the whole call fail on every rank (sharded.ShardedCollection.search).
        self.err: Optional[BaseException] = None
        self.redone = 0              # tree batches run again because a rank's deferred flag word was set

    def take_error(self) -> Optional[BaseException]:
        e, self.err = self.err, None
        return e

    def _local(self, name: str, B: int, width: int, like: torch.Tensor, *args, counts: bool = True, **kw):
        if self.err is None:
            try:
                return getattr(self.local, name)(*args, **kw)
            except Exception as e:            # kept for the status exchange; the collectives go on
                self.err = e
        k = torch.zeros((B, width), dtype=torch.int64, device=like.device)
        return (k, torch.zeros((B,), dtype=torch.int32, device=like.device)) if counts else k

    # -- exchange ---------------------------------------------------------------------
    def gather(self, keys: torch.Tensor) -> torch.Tensor:
        """[B, L] per rank -> [B, world*L] (rank-major inside a row), same on every rank."""
        if self.world == 1:
            return keys
        B, L = keys.shape
        return self.gather_raw(keys).view(self.world, B, L).permute(1, 0, 2).reshape(B, -1)

    def gather_raw(self, keys: torch.Tensor) -> torch.Tensor:
        """[B, L] per rank -> [world*B, L], rank-major (what the all-gather leaves)."""
        B, L = keys.shape
        out = torch.empty((self.world * B, L), dtype=keys.dtype, device=keys.device)   # rank-major concat
        if dist.get_backend(self.group) == "gloo":   # CPU tests / rehearsals: gloo has no flat all-gather on devices
            parts = list(out.view(self.world, B, L).unbind(0))
            host = [torch.empty((B, L), dtype=keys.dtype) for _ in parts]
            dist.all_gather(host, keys.contiguous().cpu(), group=self.group)
            for p, h in zip(parts, host):
                p.copy_(h)
        else:
            dist.all_gather_into_tensor(out, keys.contiguous(), group=self.group)
        return out

    def reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place integer sum over the ranks (the exchange of exact keys: one owner per slot, zeros elsewhere)."""
        if self.world == 1:
            return t
        if dist.get_backend(self.group) == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def sync_sparse_scale(self) -> None:
        """Collective: every shard learns the largest document weight of ANY shard, so that the integer BM25 scores of
        the select pass mean the same on all of them (the candidates-first exchange compares them across shards).
        Call after rows were added; a shard whose own maximum has outgrown the shared one is caught at search time
        (the batch is flagged and redone per shard)."""
        if not hasattr(self.local, "sparse_wmax"):
            return
        w, _ = self.local.sparse_wmax()
        t = torch.tensor([w], dtype=torch.float32)
        if self.world > 1:
            if dist.get_backend(self.group) == "gloo":
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            else:
                d = t.cuda()
                dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
                t = d.cpu()
        self.local.set_sparse_wmax(float(t.item()))

    def _global(self, keys, limit, dedupe=False):
        if self.world == 1:
            return keys, None
        return self.ops.merge(self.gather(keys), None, limit, dedupe)

    # -- stages: global lists, replicated on every rank ----------------------------------
    # `flag` (engine.HxIndex: a zeroed int32 device word): the local stage is enqueued without its host round trip and
    # adds its failure count to the word (hx_*_async); hybrid_tree reads it once behind the whole tree
    def search_dense(self, q, limit, prefix=0, flag=None):
        kw = {} if flag is None else {"flag": flag}
        k, c = self._local("search_dense", q.shape[0], limit, q, q, limit, prefix, **kw)
        return (k, c) if self.world == 1 else self._global(k, limit)

    def search_i8(self, q, limit, flag=None):
        kw = {} if flag is None else {"flag": flag}
        k, c = self._local("search_i8", q.shape[0], limit, q, q, limit, **kw)
        return (k, c) if self.world == 1 else self._global(k, limit)

    def search_sparse(self, q_indptr, q_idx, q_val, limit, flag=None):
        kw = {} if flag is None else {"flag": flag}
        k, c = self._local("search_sparse", q_indptr.shape[0] - 1, limit, q_indptr, q_indptr, q_idx, q_val, limit, **kw)
        return (k, c) if self.world == 1 else self._global(k, limit)

    def rescore(self, q, cand_keys, cand_counts, limit, prefix=0):
        # each rank scores the candidates that live in its shard (others are skipped)
        k, c = self._local("rescore", q.shape[0], limit, q, q, cand_keys, cand_counts, limit, prefix)
        return (k, c) if self.world == 1 else self._global(k, limit)

    # -- whole queries -----------------------------------------------------------------------
    def hybrid_h1(self, q, q_indptr, q_idx, q_val, dense_limit=100, sparse_limit=100, limit=10,
                  rrf_k=2.0, rank_base=0):
        if hasattr(self.local, "h1_local") and hasattr(self.ops, "h1_fuse"):
            # two ABI calls around the one exchange of the step: no per-stage host work in between
            mine = self._local("h1_local", q.shape[0], dense_limit + sparse_limit, q, q, q_indptr, q_idx, q_val,
                               dense_limit, sparse_limit, counts=False)
            allk = mine if self.world == 1 else self.gather_raw(mine)
            return self.ops.h1_fuse(allk, self.world, dense_limit, sparse_limit, limit, rrf_k, rank_base)
        # sparse first: the dense stage ends with a host read of its failure flags, and the device
        # should not sit idle behind that read with the sparse stage still to be enqueued
        sk, sc = self._local("search_sparse", q.shape[0], sparse_limit, q, q_indptr, q_idx, q_val, sparse_limit)
        dk, dc = self._local("search_dense", q.shape[0], dense_limit, q, q, dense_limit)
        if self.world > 1:  # one exchange carries both lists
            allk = self.gather(torch.cat([dk, sk], dim=1)).reshape(q.shape[0], self.world, -1)
            dk, dc = self.ops.merge(allk[:, :, :dense_limit].reshape(q.shape[0], -1), None, dense_limit, False)
            sk, sc = self.ops.merge(allk[:, :, dense_limit:].reshape(q.shape[0], -1), None, sparse_limit, False)
        return self.ops.rrf(dk, dc, sk, sc, limit, rrf_k, rank_base)

    def hybrid_tree(self, q, q_indptr, q_idx, q_val, p: dict, msizes=(64, 128, 256), rrf_k=2.0, rank_base=0,
                    rrf_limit=10, deferred: Optional[bool] = None):
        """The reference tree (qdrant_handler.py:305-372) with one exchange per cascade level.
        `deferred` (default: whenever the shard offers it): every level is enqueue -> all-gather -> enqueue, the three
        whole-collection stages keep their failure flags on the device (one word per rank), and the word is looked at
        ONCE behind the whole tree -- summed over the ranks, so that all of them take the same decision.  A batch with
        a flagged query anywhere (rare) is run again with every stage resolving its own flags (`self.redone` counts
        them).  Without it every one of the three stages parks the host -- and with it the device -- on its flags."""
        if deferred is None:
            deferred = bool(getattr(self.local, "deferred_stages", False))
        if not deferred:
            return self._tree(q, q_indptr, q_idx, q_val, p, msizes, rrf_k, rank_base, rrf_limit, None)
        flag = torch.zeros(1, dtype=torch.int32, device=q.device)
        out = self._tree(q, q_indptr, q_idx, q_val, p, msizes, rrf_k, rank_base, rrf_limit, flag)
        if self.world > 1:
            if dist.get_backend(self.group) == "gloo":
                host = flag.cpu()
                dist.all_reduce(host, group=self.group)
                flag = host
            else:
                dist.all_reduce(flag, group=self.group)
        if int(flag.item()) != 0:           # the one host look at the flags of the batch
            self.redone = getattr(self, "redone", 0) + 1
            out = self._tree(q, q_indptr, q_idx, q_val, p, msizes, rrf_k, rank_base, rrf_limit, None)
        return out

    def _tree(self, q, q_indptr, q_idx, q_val, p, msizes, rrf_k, rank_base, rrf_limit, flag):
        lim = [p[f"matryoshka_{m}_limit"] for m in msizes]
        if msizes:
            ck, cc = self.search_dense(q, lim[0], msizes[0], flag=flag)
            for m, l in zip(msizes[1:], lim[1:]):
                ck, cc = self.rescore(q, ck, cc, l, m)
            ak, ac = self.rescore(q, ck, cc, p["dense_limit"], 0)
        else:
            ak, ac = self.search_dense(q, p["dense_limit"], 0, flag=flag)
        qk, qc = self.search_i8(q, p["quantized_limit"], flag=flag)
        dk, dc = self.rescore(q, qk, qc, p["dense_limit"], 0)
        sk, sc = self.search_sparse(q_indptr, q_idx, q_val, p["sparse_limit"], flag=flag)
        if dc is None:
            dc = torch.full((q.shape[0],), dk.shape[1], dtype=torch.int32, device=dk.device)
        if sc is None:
            sc = torch.full((q.shape[0],), sk.shape[1], dtype=torch.int32, device=sk.device)
        rk, rc = self.ops.rrf(dk, dc, sk, sc, rrf_limit, rrf_k, rank_base)
        uk = torch.cat([ak, rk], dim=1)     # empty slots are 0 and are ignored downstream
        return self.rescore(q, uk, None, p["final_limit"], 0)


class H1Pipeline:
    """Consecutive H1 batches in flight ("overlap the gather of batch i with K3 of batch
    i+1 on a second stream").  `submit` enqueues the local stage of a batch on the current stream WITHOUT
    reading its failure flags (`h1_local_async`) and hands its exchange + fusion to a side stream, so they
    run beside the local stage of the next batch and the device never waits for the host.  The flags travel
    through the exchange as one extra row per rank, so every rank sees every rank's word; a batch any rank
    flagged (a retry or the exact path was needed: rare) is redone through the synchronous path, by all ranks
    alike, when it is verified -- at the latest in `wait()`.  The returned tensors are valid once `wait()`
    has passed.  With one rank, or without a HIP device (the gloo tests), there is no side stream and the
    same steps run in sequence; without `h1_local_async` it degenerates to `ShardedIndex.hybrid_h1`.

    The input tensors of a batch are kept (not copied) until its flags have been looked at -- up to `depth` + 1
    submits later, or `wait()`: a flagged batch is redone from them, so the caller must not refill them in place
    before that.  The returned tensors are allocated on the side stream and handed to the submitting stream
    (`record_stream`): the caching allocator will not reuse them for a later gather while a consumer on the
    submitting stream may still read them.
    Hardware status: over RCCL this pipeline has run with ONE rank per box only (scripts/nccl_one_rank.py); with more
    ranks it has run over gloo (CPU tests, 2-4 ranks sharing one GPU).  DESIGN.md section 7 says so."""

    def __init__(self, sh: ShardedIndex, dense_limit=100, sparse_limit=100, limit=10, rrf_k=2.0, rank_base=0,
                 depth: int = 2, force_side_stream: bool = False, candidates_first: Optional[bool] = None):
        self.sh = sh
        self.args = (dense_limit, sparse_limit, limit, rrf_k, rank_base)
        fast = hasattr(sh.local, "h1_local") and hasattr(sh.ops, "h1_fuse")
        self.deferred = fast and hasattr(sh.local, "h1_local_async") and (sh.world > 1 or force_side_stream)
        self.side = torch.cuda.Stream() if (self.deferred and torch.cuda.is_available()) else None
        self.depth = max(1, depth)
        self.pending = []          # batches whose flags have not been looked at yet, oldest first
        self.redone = 0            # batches that went through the synchronous path after all
        self._pin, self._n = None, 0   # ring of pinned host buffers for the ranks' flag words
        # Candidates first (hx.h: hx_h1_nominate_async ...): a shard sends its SHARE of the global candidate lists --
        # int8-scored rows, integer-scored documents -- the exact scores are computed once per candidate by the rank
        # that owns it, and the certificate is evaluated once on the global list.  Two collectives per batch (an
        # all-gather of the nominations, an integer-sum all-reduce of the exact keys) instead of one, both beside the
        # next batch's local stage; what a shard pays per query whatever its row count falls by about the number of
        # shards.  A batch that a check flags -- a shard's list cut above the global cut (topically clustered rows), an
        # overflow, a certificate that does not hold -- is redone through the per-shard path, and the lists are widened.
        cf_ok = (self.deferred and hasattr(sh.local, "h1_nominate_async") and hasattr(sh.ops, "h1_finish")
                 and hasattr(sh.ops, "h1_plan"))
        self.cf = cf_ok if candidates_first is None else bool(candidates_first and cf_ok)
        if self.cf:
            try:
                self.k1, self.k2, self.lp, self.k3, self.lout = sh.ops.h1_plan(dense_limit, sparse_limit, max(sh.world, 1))
                cap = 8192 // max(sh.world, 1) // 32 * 32       # world x k keys are merged in one 8192-key buffer
                L32 = (sparse_limit + 31) // 32 * 32
                self.k1max = min(max((self.lp + 31) // 32 * 32, self.k1), cap)
                self.k2max, self.k3max = min(max(L32, self.k2), cap), min(max(L32, self.k3), cap, 256)
                sh.sync_sparse_scale()
            except Exception:          # limits the plan does not take: the per-shard exchange serves them
                self.cf = False

    def _exchange_and_fuse(self, mine, B):
        sh = self.sh
        dl, sl, limit, rrf_k, rank_base = self.args
        g = mine if sh.world == 1 else sh.gather_raw(mine)              # [world * (B + 1), dl + sl], rank-major
        g3 = g.view(sh.world, B + 1, dl + sl)
        host = self._pinned(g3[:, B, 0])
        out = sh.ops.h1_fuse(g3[:, :B, :].reshape(sh.world * B, dl + sl), sh.world, dl, sl, limit, rrf_k, rank_base)
        return out, host

    def _pinned(self, flags):
        if flags.is_cuda:
            n = max(self.sh.world, 1)
            if self._pin is None:
                self._pin = [torch.zeros((n,), dtype=torch.int64, pin_memory=True) for _ in range(self.depth + 2)]
            host = self._pin[self._n % len(self._pin)][:flags.numel()]     # (more buffers than batches ever pending)
            self._n += 1
            host.copy_(flags, non_blocking=True)
            return host
        return flags.clone()

    def _exchange_candidates_first(self, nom, inputs, B):
        sh = self.sh
        dl, sl, limit, rrf_k, rank_base = self.args
        k1, k2, lp, k3 = self._cfk
        pub = nom[:B * (k1 + k2 + 2)]                                               # (the rest is this rank's own)
        g = pub if sh.world == 1 else sh.gather_raw(pub.view(1, -1)).view(-1)       # [world * B * (k1 + k2 + 2)]
        res = sh.local.h1_rescore_async(*inputs, nom, g, sh.world, sh.rank, dl, sl, k1, k2, lp, k3)
        sh.reduce_sum(res)
        keys, cnt, nfail = sh.ops.h1_finish(res, sh.world, B, lp, k3, dl, sl, limit, rrf_k, rank_base)
        return (keys, cnt), self._pinned(nfail.to(torch.int64))

    def submit(self, q, q_indptr, q_idx, q_val):
        sh = self.sh
        dl, sl, limit, rrf_k, rank_base = self.args
        if not self.deferred:
            return sh.hybrid_h1(q, q_indptr, q_idx, q_val, dl, sl, limit, rrf_k, rank_base)
        B = q.shape[0]
        cf = self.cf
        if cf:
            self._cfk = (self.k1, self.k2, self.lp, self.k3)
            mine = sh.local.h1_nominate_async(q, q_indptr, q_idx, q_val, dl, sl, self.k1, self.k2, self.lout)
        else:
            mine = sh.local.h1_local_async(q, q_indptr, q_idx, q_val, dl, sl)      # enqueued; no host round trip
        done = None
        if self.side is not None:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                if cf:
                    out, host = self._exchange_candidates_first(mine, (q, q_indptr, q_idx, q_val), B)
                else:
                    out, host = self._exchange_and_fuse(mine, B)
                done = torch.cuda.Event()
                done.record()
            mine.record_stream(self.side)
            main = torch.cuda.current_stream()
            for t in out:                       # made on the side stream, consumed on the submitting one
                t.record_stream(main)
        elif cf:
            out, host = self._exchange_candidates_first(mine, (q, q_indptr, q_idx, q_val), B)
        else:
            out, host = self._exchange_and_fuse(mine, B)
        self.pending.append((done, host, (q, q_indptr, q_idx, q_val), out, self._cfk if cf else None))
        while len(self.pending) > self.depth:
            self._verify(self.pending.pop(0))
        return out

    def _verify(self, entry):
        done, host, inputs, out, shares = entry
        if done is not None:
            done.synchronize()
        if bool((host != 0).any()):        # the same words on every rank: all ranks redo the batch together
            dl, sl, limit, rrf_k, rank_base = self.args
            if self.side is not None:      # the lists being replaced must not be written by the side stream any more
                torch.cuda.current_stream().wait_stream(self.side)
            k, c = self.sh.hybrid_h1(*inputs, dl, sl, limit, rrf_k, rank_base)
            out[0].copy_(k)
            out[1].copy_(c)
            self.redone += 1
            # (the same decision on every rank: the flag words are the same.)  Only a batch that went out with the shares
            # in force NOW says anything about them: batches verified late had the narrower ones of before
            if self.cf and shares is not None and shares[:2] + shares[3:] == (self.k1, self.k2, self.k3):
                if self.k1 >= self.k1max and self.k2 >= self.k2max and self.k3 >= self.k3max:
                    self.cf = False     # full-length lists were still not enough: the per-shard exchange from here on
                self.k1, self.k2 = min(2 * self.k1, self.k1max), min(2 * self.k2, self.k2max)
                self.k3 = min(2 * self.k3, self.k3max)

    def wait(self):
        while self.pending:
            self._verify(self.pending.pop(0))
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
