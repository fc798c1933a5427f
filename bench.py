#!/usr/bin/env python3
"""Hybrid-retrieval benchmark: queries/sec on the 10M x 768 dense+BM25 workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4]

One "step" = one batch of B queries through the hot path with every input already
resident in HBM: per-shard cosine top-100 (int8 MFMA candidate scan + exact fp32
re-score, certified) and BM25 top-100 over the on-device inverted index, per-stage all-gather of
the shards' lists over RCCL (N > 1), reciprocal-rank fusion, top-10.  The corpus is
row-sharded over the N ranks (strong scaling: total work is fixed).  Rank 0 prints one
JSON line; see DESIGN.md "Measurement" for every field."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows, dim, batch, mode)
    "cfg2": dict(rows=1_000_000, dim=384, batch=256, mode="dense", desc="1M x 384 dense-only cosine top-10, batch 256"),
    "cfg3": dict(rows=10_000_000, dim=768, batch=1024, mode="h1",
                 desc="10M x 768 dense+BM25 hybrid (dense top-100 (+) sparse top-100 -> RRF -> top-10), batch 1024"),
    "cfg4": dict(rows=100_000_000, dim=768, batch=1024, mode="dense",
                 desc="100M x 768 row-sharded dense top-10, batch 1024 (needs 8 GPUs)"),
}
PEAK_FP16_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
PEAK_I8_TOPS = 5000.0       # MI355X dense int8 MFMA: twice the fp16 rate (same guide, MFMA table)
PEAK_HBM_GBS = 8000.0       # HBM3E spec
SPARSE_ARITH = "16-bit integer select pass over the inverted index + exact re-score in upstream order (fp32 running sum, ascending term id)"
DENSE_ARITH = ("int8 MFMA candidate scan over a per-row-scaled int8 copy of the normalised rows + exact fp32 re-score of "
               "the candidates, certified per query (a query the certificate does not cover is re-run on the fp16 copy)")
PMC_PROFILE = "r04_pmc_scan.json"   # committed rocprofv3 --pmc pass the `traffic` figure is read from


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override corpus rows (testing)")
    ap.add_argument("--batch", type=int, default=0, help="override query batch (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=64,
                    help="queries of the timed batch brute-forced on the host over the WHOLE corpus")
    ap.add_argument("--no-secondary", action="store_true", help="skip tree mode / cfg2 / small-batch side measurements")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the extra K steps that time the kernels with nothing beside them (`roofline.alone`): profiler runs, "
                         "whose per-kernel averages should hold the timed region's launches only")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the CPU baseline (0 = host cores / GPUs of the host: the box share of one GPU)")
    return ap.parse_args()


CPU_BUILD = "gcc -O3 -mavx2 -fopenmp -ffp-contract=off (unfused fp32 mul + add: the arithmetic contract)"


def _merge_best(best, s, i, c, L):
    """Keep the best L (score desc, id asc) of the running per-query lists and a chunk's lists."""
    from oracle import oracle as O
    out = []
    for b in range(len(best)):
        ps, pi = best[b]
        ns, ni = np.concatenate([ps, s[b, :c[b]]]), np.concatenate([pi, i[b, :c[b]]])
        out.append(O.topk(ns, ni, L))
    return out


HOST_SHARE = {}


def host_share():
    """(threads, note): the host cores that belong to one GPU of this box = cores the process may run on, divided by
    the GPUs of the host (KFD topology: every GPU of the machine is listed there even when one is handed to us)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gpus = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for nd in os.listdir(root):
            with open(os.path.join(root, nd, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            gpus += 1 if int(props.get("simd_count", "0")) > 0 else 0
    except OSError:
        pass
    # The topology of a box that hands out ONE GPU of its host lists what the container may see (1 or 2 nodes on the
    # boxes of this pool), not what the host carries.  BASELINE.json's machine is "8 x MI355X of one node": a host
    # with the cores of such a node is taken to carry its 8 GPUs, whatever part of them is visible here.
    try:
        import torch
        gpus = max(gpus, torch.cuda.device_count())
    except Exception:
        pass
    visible = gpus
    if cores >= 128:
        gpus = max(gpus, 8)
    gpus = max(gpus, 1)
    HOST_SHARE.update(usable_host_cores=cores, visible_gpus=visible, assumed_gpus_of_host=gpus,
                      rule="cores // max(visible GPUs, 8 if cores >= 128 else 1): a reported baseline, no speed-up claim rests on it")
    return max(1, cores // gpus), f"{cores} usable host cores / {gpus} GPUs of the host (an MI355X node carries 8)"


def _median3(fn):
    """(median seconds of three runs, result of the last)"""
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1], r


def cpu_baseline(wl, sel, dim, tabs, res_keys, chunk_rows=1_000_000, share_note="", threads=0):
    """The C restatement of the path (oracle/hx_oracle.c, kind "port") timed on this box's host cores on a
    bounded sample: the queries `sel` of the timed batch, brute force over the WHOLE corpus (regenerated
    chunk by chunk, never resident at once).  The lists it produces are then compared -- ids and fp32 score
    bits -- with what the TIMED GPU steps returned for those queries."""
    from oracle import c_oracle as CO
    from oracle import oracle as O
    from rag_application_amd import synth
    rows, mode = wl["rows"], wl["mode"]
    bs = len(sel)
    L = 100 if mode == "h1" else 10
    if threads > 0:
        CO.set_num_threads(threads)      # (PyTorch sets its own OpenMP thread count at import: say ours explicitly)
    threads = CO.num_threads()
    Qall = CO.synth_dense(synth.SEED_QUERY, 0, int(sel.max()) + 1, dim)
    Qn = CO.cosine_preprocess(Qall[sel])
    if mode == "h1":
        qip_a, qix_a, qv_a = synth.sparse_queries(synth.SEED_SPQUERY, 0, int(sel.max()) + 1, tabs)
        qip = np.zeros(bs + 1, np.int64)
        qix, qv = [], []
        for k, b in enumerate(sel):
            qix.append(qix_a[qip_a[b]:qip_a[b + 1]])
            qv.append(qv_a[qip_a[b]:qip_a[b + 1]])
            qip[k + 1] = qip[k] + len(qix[-1])
        qix, qv = np.concatenate(qix).astype(np.int32), np.concatenate(qv).astype(np.float32)
    empty = (np.zeros(0, np.float32), np.zeros(0, np.int64))
    dbest, sbest = [empty] * bs, [empty] * bs
    t_cpu = 0.0
    for r0 in range(0, rows, chunk_rows):
        n = min(chunk_rows, rows - r0)
        Xn = CO.cosine_preprocess(CO.synth_dense(synth.SEED_CORPUS, r0, n, dim))
        dt, (s, i, c) = _median3(lambda: CO.search_dense(Xn, Qn, L, id_base=r0))
        t_cpu += dt
        del Xn
        dbest = _merge_best(dbest, s, i, c, L)
        if mode == "h1":
            ip, ix, v = CO.synth_sparse_docs(synth.SEED_SPDOC, r0, n, tabs)
            dt, (s, i, c) = _median3(lambda: CO.sparse_brute(ip, ix, v, qip, qix, qv, L, id_base=r0))
            t_cpu += dt
            del ip, ix, v
            sbest = _merge_best(sbest, s, i, c, L)
    t0 = time.perf_counter()
    if mode == "h1":
        out = [CO.rrf(dbest[k][1], sbest[k][1], 2.0, 0, 10) for k in range(bs)]
    else:
        out = dbest
    t_cpu += time.perf_counter() - t0
    # ---- the TIMED result against the brute force: ids and score bits, and recall@10
    ok, hit, want = True, 0, 0
    gs, gi, gc = res_keys
    for k, b in enumerate(sel):
        es, ei = out[k]
        m = len(ei)
        ok &= int(gc[b]) == m and np.array_equal(gi[b, :m], ei) and \
            np.array_equal(gs[b, :m].view(np.uint32), np.asarray(es, np.float32).view(np.uint32))
        hit += len(np.intersect1d(gi[b, :int(gc[b])], ei))
        want += m
    return dict(value=bs / t_cpu, unit="queries/sec", cores=threads, host_cores=os.cpu_count(), kind="port",
                build=CPU_BUILD,
                sample=f"{bs} of {wl['batch']} queries of the timed batch (every {wl['batch'] // bs}th) x ALL {rows} rows, "
                       f"brute force chunk by chunk ({t_cpu:.2f} s of CPU search work, data generation not counted); "
                       f"document-at-a-time sparse scoring; every chunk's search timed three times, the medians summed; "
                       f"{threads} threads = {share_note}",
                host_share=dict(HOST_SHARE),
                parity_on_sample=bool(ok), recall_at_10=(hit / want if want else None),
                checked="ids and fp32 score bits of the LAST TIMED STEP's lists for the sampled queries")


def ingest_leg(eng, synth, torch, local, t_build, rows, nnz, tabs):
    """BASELINE config 5 on one GPU, chunks/sec (ranks ingest their own deal of the chunks:
    sharded.ShardedCollection.store).  Three legs:
      store_from_host   what store_document_vectors hands over (qdrant_handler.py:120-198): fp32 rows + the sparse CSR
                        in host memory -> hx_add_rows (PCIe-inclusive: pageable host buffers, staged 65536 rows at a
                        time) -> the inverted index over them (K9);
      store_from_device the same rows already in HBM, as an encoder on this GPU leaves them -> hx_add_rows_dev;
      encode_and_append a bge-base-shaped BERT (random init: no checkpoint ships), bf16, the reference's unmasked mean
                        pooling (app/core/models/huggingface/huggingface.py:165-170), output appended where it lies.
    K1/K2 (k_prep_rows) is reported against the HBM roofline from HIP events around its launches (hx_profile slot 4:
    the raw row read once, every derived copy written once)."""
    out = dict(what="config 5 legs on ONE GPU",
               synthetic_index_build=dict(rows=rows, postings=nnz, seconds_incl_generation_on_device=t_build,
                                          note="rows generated ON the device (hx_synth_fill): not an ingest rate"))
    try:
        n_ing, dim = 131072, 768
        X, ip, ix_, v = synth.ingest_batch(7, n_ing, dim, tabs)   # host arrays of the corpus' shape (not its stream)
        for leg in ("store_from_host", "store_from_device"):
            sc = eng.HxIndex(dim, (64, 128, 256), device=local)
            sc.reserve(n_ing, int(ip[-1]))
            Xd = torch.from_numpy(X).to(f"cuda:{local}") if leg == "store_from_device" else None
            sc.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if Xd is None:
                sc.add(X, ip, ix_, v)
            else:
                sc.add_device(Xd, ip, ix_, v)
            t_add = time.perf_counter() - t0
            sc.finalize()
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            pr = sc.profile_read()["prep_rows"]
            out[leg] = dict(chunks=n_ing, postings=int(ip[-1]), chunks_per_sec=n_ing / t_all, seconds_add=t_add,
                            seconds_index_build=t_all - t_add,
                            k_prep_rows=dict(launches=pr["launches"], ms=pr["ms"], alg_gb=pr["bytes"] / 1e9,
                                             gbs=pr["bytes"] / pr["ms"] / 1e6 if pr["ms"] else None,
                                             frac_of_hbm_peak=pr["bytes"] / pr["ms"] / 1e6 / PEAK_HBM_GBS if pr["ms"] else None))
            sc.close()
            del Xd
        del X
        # K1/K2 in steady state: 1M rows already on the device = 16 launches of k_prep_rows (65536 rows each); the two
        # legs above time two cold launches each, which is what an upsert of 131072 chunks costs, not what the kernel does
        n_st = 1 << 20
        sc = eng.HxIndex(dim, (64, 128, 256), device=local)
        sc.reserve(n_st)
        Xd = torch.rand((n_st, dim), device=f"cuda:{local}", dtype=torch.float32) * 2.0 - 1.0
        sc.add_device(Xd[:65536].contiguous())          # first-touch of the buffers
        sc.truncate(0)
        sc.profile(True)
        sc.profile_read()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.add_device(Xd)
        torch.cuda.synchronize()
        t_add = time.perf_counter() - t0
        pr = sc.profile_read()["prep_rows"]
        out["prep_rows_steady"] = dict(rows=n_st, chunks_per_sec=n_st / t_add, launches=pr["launches"], ms=pr["ms"],
                                       alg_gb=pr["bytes"] / 1e9, gbs=pr["bytes"] / pr["ms"] / 1e6 if pr["ms"] else None,
                                       frac_of_hbm_peak=pr["bytes"] / pr["ms"] / 1e6 / PEAK_HBM_GBS if pr["ms"] else None,
                                       note="K1/K2 (k_prep_rows) over 16 consecutive launches, HIP events per launch: D*4 read + "
                                            "every derived copy written once (11,144 B per row at D = 768)")
        sc.close()
        del Xd
    except Exception as e:
        out["store_error"] = repr(e)[:300]
    try:
        from transformers import BertConfig, BertModel
        Bn, S, NB = 256, 128, 12
        torch.manual_seed(0)
        cfg = BertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                         intermediate_size=3072, max_position_embeddings=512)
        model = BertModel(cfg, add_pooling_layer=False).to(f"cuda:{local}").to(torch.bfloat16).eval()
        ids = torch.randint(1000, 30000, (Bn, S), device=f"cuda:{local}")
        mask = torch.ones((Bn, S), dtype=torch.long, device=f"cuda:{local}")
        sc = eng.HxIndex(768, (64, 128, 256), device=local)
        sc.reserve(Bn * (NB + 3))

        def step():
            with torch.no_grad():
                e = model(input_ids=ids, attention_mask=mask).last_hidden_state.mean(dim=1)
            sc.add_device(e.float())

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(NB):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["encode_and_append"] = dict(encoder="BERT-base shape (bge-base), random init, bf16", batch=Bn, seq_len=S,
                                        batches=NB, chunks_per_sec=Bn * NB / dt, rows_in_index=sc.count())
        sc.close()
        del model
        torch.cuda.empty_cache()
    except Exception as e:       # the encoder is PyTorch plumbing around the path, not the path: report and move on
        out["encode_and_append"] = dict(error=repr(e)[:200])
    return out


def secondary(eng, synth, torch, ix, wl, tabs, Q, sp_q, local, hp_tree, hp_h1=None, last_res=None, steps=3):
    """Side measurements on the driver record: the reference tree on the same index, the
    bandwidth-bound dense kNN (B = 1 / 8 / 32) on the same corpus, and BASELINE config 2."""
    out = {}
    B = Q.shape[0]

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, r

    if wl["mode"] == "h1":
        dt, _ = timed(lambda: ix.hybrid_query(Q, *sp_q, hp_tree), steps)
        out["tree_mode"] = dict(what="reference tree (qdrant_handler.py:305-372), P-mcp limits 100/80/60/40/40/50/30, "
                                     "same index and batch", queries_per_sec=B / dt, ms_per_step=dt * 1e3)
        # the same step with the fp16 copy nominating the dense candidates (round 2's path; the lists are the same)
        ix.set_dense_candidates("f16")
        ix.profile(True)
        ix.profile_read()
        dt, r16 = timed(lambda: ix.hybrid_query(Q, *sp_q, hp_h1), steps)
        p16 = ix.profile_read()["scan_f16"]
        ix.profile(False)
        ix.set_dense_candidates("i8")
        out["fp16_candidates"] = dict(
            what="the timed step with the dense candidates nominated by the fp16 scan instead of the int8 scan",
            queries_per_sec=B / dt, ms_per_step=dt * 1e3, same_lists_as_the_timed_step=bool(
                last_res is not None and torch.equal(r16[0], last_res[0]) and torch.equal(r16[1], last_res[1])),
            scan=dict(kernel="k_scan8<fp16, v_mfma_f32_16x16x32_f16>", launches=p16["launches"],
                      avg_launch_ms=p16["ms"] / max(p16["launches"], 1),
                      achieved_tflops=p16["flops"] / p16["ms"] / 1e9 if p16["ms"] else None, peak_tflops=PEAK_FP16_TFLOPS,
                      frac=p16["flops"] / p16["ms"] / 1e9 / PEAK_FP16_TFLOPS if p16["ms"] else None))
    runs = []
    L10 = 10
    flag = torch.zeros(1, dtype=torch.int32, device=Q.device)   # the stages' failure counts (hx_*_async): read once, below
    for b in (1, 8, 32):
        q = Q[:b].contiguous()
        # candidates re-scored per query (engine.hip: cand8_lprime / geometry): their fp32 rows are algorithmic bytes too
        for name, fn, key, ncand in (
                ("int8 candidate scan + exact fp32 re-score (the dense stage)", lambda: ix.search_dense(q, L10, flag=flag),
                 "scan_cand8", max(9 * L10 // 2, L10 + 288)),
                ("fp16 candidate scan + exact fp32 re-score", lambda: ix.search_dense(q, L10, flag=flag), "scan_f16",
                 L10 + max(32, L10 // 2)),
                ("int8 scan of the 'quantized' vector (exact integer scores)", lambda: ix.search_i8(q, L10, flag=flag),
                 "scan_i8", 0)):
            ix.set_dense_candidates("f16" if key == "scan_f16" else "i8")
            fn()
            torch.cuda.synchronize()
            ix.profile(True)
            ix.profile_read()
            dt, _ = timed(fn, 5)
            p = ix.profile_read()[key]
            ix.profile(False)
            if p["ms"] > 0:
                gbs = p["bytes"] / p["ms"] / 1e6
                pass_bytes = p["bytes"] / 6 + b * ncand * wl["dim"] * 4          # (1 + 5 calls were profiled)
                runs.append(dict(batch=b, stage=name, ms_per_pass=dt * 1e3, scan_ms_per_pass=p["ms"] / 6,
                                 scan_gbs=gbs, frac_of_hbm_peak=gbs / PEAK_HBM_GBS,
                                 pass_frac_of_hbm_peak=pass_bytes / dt / 1e9 / PEAK_HBM_GBS, queries_per_sec=b / dt))
    flagged = int(flag.item())
    ix.set_dense_candidates("i8")
    out["dense_knn_small_batch"] = dict(
        what=f"dense kNN top-10 over the same {wl['rows']} x {wl['dim']} corpus, B queries per pass.  frac_of_hbm_peak: "
             "algorithmic bytes of the scanned copy (rows * row_bytes + B * row_bytes: 1 B per element for the int8 copies, "
             "2 B for fp16; SURVEY 8(d)) / HIP-event time of the scan launches.  pass_frac_of_hbm_peak: the same bytes plus "
             "the re-scored candidates' fp32 rows / WALL time of the pass (prep, compactions, re-score, certificate "
             "included; no host round trip: the stages' flags go to one device word, read once behind all passes -- "
             "hx_search_*_async).  North-star target >= 0.70",
        peak_gbs=PEAK_HBM_GBS, runs=runs, queries_not_final_in_any_pass=flagged)
    # BASELINE config 2 on an index of its own (1M x 384, dense only, B = 256), checked against the C oracle
    c2 = WORKLOADS["cfg2"]
    ix2 = eng.HxIndex(c2["dim"], (64, 128, 256), device=local)
    ix2.synth_fill(c2["rows"], synth.SEED_CORPUS)
    Q2 = eng.synth_queries_dense(c2["dim"], 0, c2["batch"], synth.SEED_QUERY, device=local)
    dt, (k2, n2) = timed(lambda: ix2.search_dense(Q2, 10), 10)
    s2, i2 = eng.unpack(k2)
    s2, i2, n2 = s2.cpu().numpy(), i2.cpu().numpy(), n2.cpu().numpy()
    ix2.close()
    from oracle import c_oracle as CO
    sel = np.arange(0, c2["batch"], 4)
    Xn = CO.cosine_preprocess(CO.synth_dense(synth.SEED_CORPUS, 0, c2["rows"], c2["dim"]))
    Qn = CO.cosine_preprocess(CO.synth_dense(synth.SEED_QUERY, 0, c2["batch"], c2["dim"])[sel])
    es, ei, ec = CO.search_dense(Xn, Qn, 10)
    ok = all(int(n2[b]) == int(ec[k]) and np.array_equal(i2[b, :ec[k]], ei[k, :ec[k]]) and
             np.array_equal(s2[b, :ec[k]].view(np.uint32), es[k, :ec[k]].view(np.uint32)) for k, b in enumerate(sel))
    out["cfg2"] = dict(what=c2["desc"], queries_per_sec=c2["batch"] / dt, ms_per_step=dt * 1e3,
                       parity_vs_brute_force=bool(ok), checked=f"{len(sel)} of {c2['batch']} queries x all rows, ids + score bits")
    return out


class _LazyRows:
    """ids / payloads of a bench collection made on demand (10M payload dicts would be 20 GB of Python objects): the
    handler's code path -- id and payload looked up per returned row, ScoredPoint built -- is what is timed."""

    def __init__(self, make):
        self.make = make

    def __getitem__(self, r):
        return self.make(int(r))


def boundary_legs(eng, torch, ix, Q, sp_np, P, local):
    """The reference's own call shape on the driver record (qdrant_handler.py:269-279 one query per hybrid_search call,
    Python lists in; :363-372 the query): `latency_b1` = ONE query per call through hx_hybrid_query_host (pageable host
    buffers in, host buffers out, PCIe-inclusive), tree and H1, median of 50 calls; `host_boundary` = the whole batch
    through the same entry from numpy buffers, and through QdrantHandler.hybrid_search_batch from Python lists with
    payloads attached.  None of this is `value` (which starts with the inputs resident in HBM)."""
    import asyncio
    from rag_application_amd import handler as H
    out = {}
    Qh = Q.cpu().numpy()
    qip, qix, qv = sp_np
    B = Qh.shape[0]
    hp = {"tree": eng.make_params(P, mode=eng.HX_MODE_TREE), "h1": eng.make_params(P, mode=eng.HX_MODE_H1)}
    lat = {}
    for mode in ("tree", "h1"):
        ts = []
        for i in range(55):
            b = (i * 37) % B
            a, e = int(qip[b]), int(qip[b + 1])
            one = (Qh[b:b + 1], np.array([0, e - a], np.int64), qix[a:e], qv[a:e])
            t0 = time.perf_counter()
            ix.hybrid_query_host(*one, hp[mode])
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[5:])
        lat[mode] = dict(median_ms=ts[len(ts) // 2] * 1e3, p10_ms=ts[len(ts) // 10] * 1e3, p90_ms=ts[len(ts) * 9 // 10] * 1e3,
                         calls=len(ts))
    out["latency_b1"] = dict(what="ONE query per call through hx_hybrid_query_host (the reference's hybrid_search call shape, "
                                  "qdrant_handler.py:269-279, 363-372): host buffers in and out, PCIe and the host round trips "
                                  "included; 10M x 768 index; tree = P-mcp limits 100/80/60/100/40/100/10, h1 = 100 (+) 100 -> 10",
                             **lat)
    hb = {}
    for mode in ("h1", "tree"):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ix.hybrid_query_host(Qh, qip, qix, qv, hp[mode])
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        hb[f"abi_{mode}"] = dict(ms_per_batch=t * 1e3, queries_per_sec=B / t)
    # the handler: Python lists in (what the reference's callers hold), ScoredPoint objects with payloads out
    h = H.QdrantHandler(device=local)
    col = H._Collection(ix.dim, ix.msizes, local, index=ix)
    # ids / payloads of 10M rows would be 20 GB of Python objects: a pool of 65536 of each, looked up by row -- what the
    # handler pays per returned row (a list index, a ScoredPoint) is what a real collection's lists cost it
    pool_ids = [f"00000000-0000-4000-8000-{r:012x}" for r in range(65536)]
    pool_pay = [{"content": f"chunk {r}", "file_name": f"doc{r >> 6}.txt", "page_number": r & 63, "chunk_number": r & 63,
                 "document_summary": "", "context": None} for r in range(65536)]
    col.ids = _LazyRows(lambda r: pool_ids[r & 65535])
    col.payloads = _LazyRows(lambda r: pool_pay[r & 65535])
    h._collections["bench"] = col
    dense_lists = Qh.tolist()
    sparse_dicts = [{"indices": qix[qip[b]:qip[b + 1]].tolist(), "values": qv[qip[b]:qip[b + 1]].tolist()} for b in range(B)]
    for name, dv in (("handler_from_python_lists", dense_lists), ("handler_from_ndarray", Qh)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = asyncio.run(h.hybrid_search_batch("bench", dv, sparse_dicts, top_k=10, search_params=P, mode="h1"))
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[1]
        hb[name] = dict(ms_per_batch=t * 1e3, queries_per_sec=B / t, results=sum(len(r) for r in res))
    t0 = time.perf_counter()
    np.asarray(dense_lists, dtype=np.float32)
    hb["list_to_ndarray_ms"] = (time.perf_counter() - t0) * 1e3
    hb["what"] = (f"B = {B} through hx_hybrid_query_host from pageable numpy buffers (abi_*), and through "
                  "QdrantHandler.hybrid_search_batch (mode h1) with ScoredPoint + payload per returned row; ids / payloads come from "
                  "a pool of 65536 looked up by row (10M dicts do not fit a bench); `list_to_ndarray_ms` = the cost of packing "
                  "B x 768 Python floats alone (20-28 ns per float in CPython: the reference's call shape hands over lists)")
    h._collections.pop("bench")          # (the index belongs to the caller)
    out["host_boundary"] = hb
    return out


def cfg4_shard_leg(eng, synth, torch, local, tabs, Q, sp_q, hp, B, sel, dim, check, share_note, share=0):
    """BASELINE config 4 = 100M x 768 row-sharded over 8 GPUs = 12.5M rows per GPU.  One such shard (rank 0's: rows
    [0, 12.5M) of the same generator, with its postings) on this one GPU: the H1 step every rank would run before the
    exchange, its lists brute-forced on the host like the main workload's."""
    rows = 12_500_000
    out = dict(what="one rank's shard of BASELINE config 4 (100M x 768 over 8 GPUs): 12.5M x 768 + postings on ONE GPU, "
                    "H1 step at B = 1024 -- the per-rank step before the exchange", rows=rows, batch=B)
    try:
        t0 = time.perf_counter()
        ix = eng.HxIndex(dim, (64, 128, 256), device=local)
        ix.reserve(rows)
        ix.synth_fill(rows, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
        ix.finalize()
        torch.cuda.synchronize()
        out["build_s"] = round(time.perf_counter() - t0, 2)
        for _ in range(2):
            res = ix.hybrid_query(Q, *sp_q, hp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            res = ix.hybrid_query(Q, *sp_q, hp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        st = ix.stats()
        out.update(ms_per_step=dt * 1e3, queries_per_sec_per_rank=B / dt,
                   x8_before_the_exchange=f"{8 * rows} rows at {B / dt:.0f} queries/s if the all-gather of 8 x {B} x 200 keys "
                                          "hides behind the next batch (distributed.H1Pipeline)",
                   nnz=st["nnz"], uncertified_queries=st["cand8_uncertified_queries"],
                   exact_fallback_queries=st["dense_fallback_queries"], sparse_fallback_queries=st["sparse_fallback_queries"],
                   hbm_gb=(st["bytes_dense_f32"] + st["bytes_dense_f16"] + st["bytes_i8"] + st["bytes_i8_cand"] +
                           st["bytes_prefix"] + st["bytes_sparse"]) / 1e9)
        if check:
            gs, gi = eng.unpack(res[0])
            res_np = (gs.cpu().numpy(), gi.cpu().numpy(), res[1].cpu().numpy())
            wl = dict(rows=rows, mode="h1", batch=B)
            cb = cpu_baseline(wl, sel, dim, tabs, res_np, share_note=share_note, threads=share)
            out.update(parity_on_sample=cb["parity_on_sample"], recall_at_10=cb["recall_at_10"], checked=cb["checked"],
                       cpu_queries_per_sec=cb["value"], sample=cb["sample"])
        ix.close()
    except Exception as e:
        out["error"] = repr(e)[:300]
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process starts N fresh copies
    of itself, one rank per GPU, relays rank 0's JSON line and exits with the first non-zero code of a child.  It
    never imports torch and never touches HIP itself (a parent that had initialised the GPU could not hand it to
    children safely); the device count comes from a short-lived child.  Fewer visible devices than N is an error line,
    not a silent rehearsal -- ranks share GPUs only under the explicit HX_DIST_BACKEND=gloo."""
    import subprocess
    n = args.gpus
    backend = os.environ.get("HX_DIST_BACKEND", "nccl")
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        ndev = int(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        ndev = -1
        print(f"[bench] could not count devices: {e!r}", file=sys.stderr)
    if backend == "nccl" and ndev < n:
        print(json.dumps({"metric": "queries/sec, 10M x 768 hybrid dense+BM25 (RRF top-10)", "value": None,
                          "unit": "queries/sec", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "error": f"--gpus {n} but {ndev} HIP device(s) visible; one rank per GPU over RCCL needs {n} "
                                   "(rehearsal on fewer GPUs: HX_DIST_BACKEND=gloo)"}), flush=True)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    limit = float(os.environ.get("HX_BENCH_TIMEOUT", "3000"))
    t0, rc, line = time.time(), 0, None
    import threading
    got = []
    rd = threading.Thread(target=lambda: got.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c
                    print(f"[bench] rank {r} exited with code {c}", file=sys.stderr)
        if time.time() - t0 > limit:
            rc = 124
            print(f"[bench] ranks still running after {limit:.0f} s: killed", file=sys.stderr)
        if live and rc == 0:
            time.sleep(0.2)
    for r in live:               # a failed or timed-out job: the exact children started here, by pid
        procs[r].kill()
    for p_ in procs:
        p_.wait()
    rd.join(timeout=5)
    for ln in got:
        if ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    if line:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("[bench] rank 0 printed no result line", file=sys.stderr)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    share, share_note = host_share()
    if args.cpu_threads > 0:
        share, share_note = args.cpu_threads, f"--cpu-threads {args.cpu_threads}"
    os.environ["OMP_NUM_THREADS"] = str(share)
    import torch
    import torch.distributed as dist
    from rag_application_amd import engine as eng, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("HX_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on fewer GPUs
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        wl["rows"] = args.rows
    if args.batch:
        wl["batch"] = args.batch
    rows, dim, B, mode = wl["rows"], wl["dim"], wl["batch"], wl["mode"]
    tabs = synth.tables() if mode == "h1" else None

    # ---- build this rank's shard (contiguous rows) --------------------------------
    r0 = rows * rank // world
    r1 = rows * (rank + 1) // world
    t_build = time.perf_counter()
    ix = eng.HxIndex(dim, (64, 128, 256), device=local, id_base=r0)
    ix.reserve(r1 - r0)
    ix.synth_fill(r1 - r0, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
    torch.cuda.synchronize()
    # K9 (the on-device inverted-index build) on its own: HIP events on the stream it runs on, generation excluded
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_a.record()
    ix.finalize()
    ev_b.record()
    torch.cuda.synchronize()
    k9_cold_ms = ev_a.elapsed_time(ev_b)
    t_build = time.perf_counter() - t_build
    # ... and once more (hx_rebuild_sparse): the first build of a fresh process also pays for 32 GB of temporary device
    # allocations -- 45 ms or 1.2 s on two boxes of the pool for the same kernels; the second is the kernels' time
    k9_ms = k9_cold_ms
    if rank == 0 and world == 1 and mode == "h1" and not args.no_secondary:
        ev_a.record()
        ix.rebuild_sparse()
        ev_b.record()
        torch.cuda.synchronize()
        k9_ms = ev_a.elapsed_time(ev_b)

    # ---- queries, resident in HBM ------------------------------------------------------
    Q = eng.synth_queries_dense(dim, 0, B, synth.SEED_QUERY, device=local)
    if mode == "h1":
        qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
        qip_d, qix_d, qv_d = (torch.from_numpy(a).to(dev) for a in (qip, qix, qv))
    P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
             quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128)
    hp = eng.make_params(P, mode=eng.HX_MODE_H1)

    from rag_application_amd.distributed import ShardedIndex, H1Pipeline
    from rag_application_amd.sharded import bcast_queries
    sh = ShardedIndex(ix)      # one process per GPU; exchange = one RCCL all-gather per stage
    if world > 1 and rank != 0:
        # N > 1: the batch exists on the front rank only; every step starts with its broadcast (C2, one packed
        # buffer) -- the other ranks drop their copies so that nothing but the broadcast can feed them
        Q = None
        if mode == "h1":
            qip_d = qix_d = qv_d = None
    # N > 1: the exchange + fusion of a batch run on a side stream beside the local stage of the next
    pipe = H1Pipeline(sh, 100, 100, 10) if (world > 1 and (backend == "nccl" or os.environ.get("HX_BENCH_PIPE"))) \
        else None

    empty_sp = (torch.zeros(B + 1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                torch.zeros(0, dtype=torch.float32, device=dev))
    # the three header words of a batch travel over a host-side group: over RCCL reading them would park the host
    # behind the previous batch's kernels and the device would idle while it catches up (sharded.bcast_queries)
    hdr_group = dist.new_group(backend="gloo") if (world > 1 and backend == "nccl") else None
    # ... and the payload over a communicator of its own: ProcessGroupNCCL runs one group's collectives in order on one
    # internal stream, so on the stages' group broadcast(i + 1) would queue behind all-gather(i), which waits for the
    # local stage of batch i -- the exchange would sit on the critical path of every step instead of beside it
    bq_group = dist.new_group(backend="nccl") if (world > 1 and backend == "nccl") else None
    host_group = hdr_group          # a host-side group for decisions all ranks must take alike (gloo: the default group)

    def step():
        if world == 1:
            if mode == "h1":
                return ix.hybrid_query(Q, qip_d, qix_d, qv_d, hp)    # whole pipeline behind one ABI call
            return sh.search_dense(Q, 10)
        q, ip, ixx, vv = bcast_queries(Q, *((qip_d, qix_d, qv_d) if mode == "h1" else (empty_sp if rank == 0 else (None,) * 3)),
                                       src=0, group=bq_group, device=dev, header_group=hdr_group)
        if mode == "h1":
            if pipe is not None:
                return pipe.submit(q, ip, ixx, vv)
            return sh.hybrid_h1(q, ip, ixx, vv, 100, 100, 10)
        return sh.search_dense(q, 10)

    pipe_note = None
    werr = None
    try:
        for _ in range(args.warmup):
            step()
        if pipe is not None:
            pipe.wait()
    except Exception as e:      # noqa: BLE001
        werr = e
    if world > 1:
        # The decision is COLLECTIVE: an exception on one rank only would otherwise leave the ranks issuing different
        # collective sequences.  Every rank learns over the host-side group whether any rank failed its warm-up.
        f = torch.tensor([1 if werr is not None else 0], dtype=torch.int32)
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=host_group)
        if int(f.item()) and werr is None:
            werr = RuntimeError("another rank failed its warm-up")
    if werr is not None:
        # N > 1 only: an error in the pipelined exchange must not cost the line -- every rank falls back to the
        # synchronous exchange together and the line says so; without a pipeline there is nothing to fall back to
        if pipe is None:
            raise werr
        pipe_note = repr(werr)[:300]
        pipe = None
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
    ix.profile(True)
    ix.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if pipe is not None:
        pipe.wait()          # every batch's flag words looked at (a flagged batch is redone here at the latest)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ix.profile_read()
    ix.profile(False)
    # The timed step runs its sparse stage on a second stream BESIDE the dense scans: a launch's duration above is what the
    # kernel took while it shared the chip.  The same K steps once more with every stage on one stream (hx_set_stream_overlap)
    # give each kernel's duration alone -- reported beside the in-situ figures as `roofline.alone`, never as `value`.
    alone = None
    if world == 1 and mode == "h1" and not args.no_alone:
        ix.set_stream_overlap(False)
        step()
        torch.cuda.synchronize()
        ix.profile(True)
        ix.profile_read()
        t0a = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        alone = dict(dt=time.perf_counter() - t0a, prof=ix.profile_read())
        ix.profile(False)
        ix.set_stream_overlap(True)
    # Rehearsal aid (HX_BENCH_VERIFY=1, N > 1, small --rows): rank 0 also builds the UNSHARDED corpus and
    # checks the last step's lists against it, key for key.  Never part of the timed region.
    verified = None
    if world > 1 and mode == "h1" and os.environ.get("HX_BENCH_VERIFY"):
        if rank == 0:
            one = eng.HxIndex(dim, (64, 128, 256), device=local, id_base=0)
            one.reserve(rows)
            one.synth_fill(rows, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
            k1, c1 = one.hybrid_query(Q, qip_d, qix_d, qv_d, hp)      # (rank 0 holds the batch)
            verified = bool(torch.equal(k1, res[0]) and torch.equal(c1, res[1]))
            one.close()
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # N > 1, side measurement (never part of `value`): the reference's own query -- the tree -- through the sharded
    # path: one all-gather per cascade level, the stages' flags deferred into one all-reduced word (distributed.py)
    tree_sharded = None
    if world > 1 and mode == "h1" and not args.no_secondary:
        Pt = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=40,
                  quantized_limit=40, sparse_limit=50, final_limit=30, hnsw_ef=128)

        def tree_step():
            q, ip, ixx, vv = bcast_queries(Q, qip_d, qix_d, qv_d, src=0, group=bq_group, device=dev, header_group=hdr_group)
            return sh.hybrid_tree(q, ip, ixx, vv, Pt)

        try:        # (a failure here must not cost the line its main measurement, which is complete at this point)
            for _ in range(2):
                tree_step()
            torch.cuda.synchronize()
            dist.barrier()
            t0t = time.perf_counter()
            nt = 5
            for _ in range(nt):
                tree_step()
            torch.cuda.synchronize()
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t0t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tree_sharded = dict(what="reference tree (qdrant_handler.py:305-372), P-mcp limits, row-sharded: one all-gather "
                                     "per cascade level, flags deferred (hx_search_*_async)", steps=nt,
                                ms_per_step=float(tt.item()) / nt * 1e3, queries_per_sec=B * nt / float(tt.item()),
                                batches_redone=int(getattr(sh, "redone", 0)))
        except Exception as e:
            tree_sharded = dict(error=repr(e)[:300])

    # ---- roofline of the dominant kernel (the dense stage's candidate scan), measured with HIP events ------
    use8 = prof["scan_cand8"]["launches"] > 0
    sc = prof["scan_cand8"] if use8 else prof["scan_f16"]
    peak_mfma = PEAK_I8_TOPS if use8 else PEAK_FP16_TFLOPS
    roof = None
    if sc["launches"]:
        sec = sc["ms"] / 1e3
        tf = sc["flops"] / sec / 1e12
        gbs = sc["bytes"] / sec / 1e9
        mfma_bound = (sc["flops"] / (peak_mfma * 1e12)) >= (sc["bytes"] / (PEAK_HBM_GBS * 1e9))
        # HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE pass of this same command (gfx950
        # correction x2, MI355X_MICROARCH.md), kept under profiles/ -- counters cannot be read from
        # inside the timed process.  null when no pass for this configuration is committed.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if world == 1 and os.path.exists(tpath):
            with open(tpath) as f:
                tp = json.load(f)
            if tp.get("config") == {"rows": rows, "dim": dim, "batch": B, "n_gpus": 1}:
                traffic = tp["scan_traffic_gb_per_step"] / max(sc["launches"] / args.steps, 1)
        if B > 128:
            kname = ("k_scan8<int8, v_mfma_i32_16x16x64_i8> (256x256 tile) over the per-row-scaled int8 copy" if use8
                     else "k_scan8<fp16, v_mfma_f32_16x16x32_f16> (256x256 tile)")
        else:
            kname = "k_scan<int8> (128-row tiles)" if use8 else "k_scan<fp16> (128-row tiles)"
        roof = dict(kernel=kname,
                    bound="mfma" if mfma_bound else "hbm",
                    achieved=tf if mfma_bound else gbs, peak=peak_mfma if mfma_bound else PEAK_HBM_GBS,
                    unit="TFLOP/s" if mfma_bound else "GB/s",
                    unit_note=("int8 multiply-accumulates counted as 2 ops each, against the dense int8 MFMA peak (2x fp16)"
                               if use8 else "fp16 MFMA"),
                    frac=(tf / peak_mfma) if mfma_bound else (gbs / PEAK_HBM_GBS), traffic=traffic,
                    traffic_unit=f"GB per launch, from the committed PMC pass profiles/{PMC_PROFILE} (not measured in this run)",
                    launches=sc["launches"], avg_launch_ms=sc["ms"] / sc["launches"],
                    alg_tflop_per_launch=sc["flops"] / sc["launches"] / 1e12,
                    alg_gb_per_launch=sc["bytes"] / sc["launches"] / 1e9,
                    other_bound_frac=(gbs / PEAK_HBM_GBS) if mfma_bound else (tf / peak_mfma),
                    sparse_ms_per_step=prof["sparse"]["ms"] / max(args.steps, 1))
        if alone is not None:
            asc = alone["prof"]["scan_cand8"] if use8 else alone["prof"]["scan_f16"]
            asp = alone["prof"]["sparse"]
            if asc["launches"] and asc["ms"] > 0:
                atf, agb = asc["flops"] / asc["ms"] / 1e9, asc["bytes"] / asc["ms"] / 1e6
                roof["frac_alone"] = (atf / peak_mfma) if mfma_bound else (agb / PEAK_HBM_GBS)
                roof["note"] = ("`achieved` / `frac` / `avg_launch_ms` are IN SITU: in the timed region k_sparse_select runs on a second "
                                "stream beside this kernel and the two share the CUs, so a launch lasts longer than the kernel needs "
                                "(the step is shorter for it).  `frac_alone` / `alone`: the same K steps with one kernel at a time.")
                roof["alone"] = dict(
                    what="the same K steps with every stage on ONE stream (hx_set_stream_overlap(0)): the kernels' durations when "
                         "nothing runs beside them; the timed region above runs the sparse stage beside the dense scans",
                    ms_per_step=alone["dt"] / args.steps * 1e3, launches=asc["launches"],
                    avg_launch_ms=asc["ms"] / asc["launches"],
                    achieved=atf if mfma_bound else agb, frac=(atf / peak_mfma) if mfma_bound else (agb / PEAK_HBM_GBS),
                    second_kernel=(dict(avg_launch_ms=asp["ms"] / asp["launches"], achieved=asp["bytes"] / asp["ms"] / 1e6,
                                        frac=asp["bytes"] / asp["ms"] / 1e6 / PEAK_HBM_GBS) if asp["launches"] and asp["ms"] > 0 else None))
        sp = prof["sparse"]
        if sp["launches"] and sp["ms"] > 0:   # second kernel of the step, HBM-bound by construction
            roof["second_kernel"] = dict(kernel="k_sparse_select", bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS,
                                         achieved=sp["bytes"] / sp["ms"] / 1e6,
                                         frac=sp["bytes"] / sp["ms"] / 1e6 / PEAK_HBM_GBS,
                                         alg_gb_per_launch=sp["bytes"] / sp["launches"] / 1e9,
                                         avg_launch_ms=sp["ms"] / sp["launches"],
                                         note="8 B per posting of the queries' terms (SURVEY 8d); HIP events around the select launches"
                                              + ("; IN SITU the launch shares the chip with the dense scans and its workgroups wait for theirs "
                                                 "-- its duration is not the kernel's: see roofline.alone.second_kernel" if alone is not None else ""))

    # ---- CPU baseline + parity of the TIMED result (rank 0, N = 1 only) ------------------------------
    cpu = None
    side = None
    st_main = ix.stats() if rank == 0 else None
    if rank == 0 and world == 1:
        nq = max(1, min(args.cpu_queries, B))
        sel = np.arange(0, B, max(B // nq, 1))[:nq]
        if not args.no_cpu_baseline:      # (before the side measurements: they end with this index freed)
            gs, gi = eng.unpack(res[0])
            res_np = (gs.cpu().numpy(), gi.cpu().numpy(), res[1].cpu().numpy())
            cpu = cpu_baseline(wl, sel, dim, tabs, res_np, share_note=share_note, threads=share)
        if not args.no_secondary:
            hp_tree = eng.make_params(P, mode=eng.HX_MODE_TREE)
            side = secondary(eng, synth, torch, ix, wl, tabs, Q, (qip_d, qix_d, qv_d) if mode == "h1" else None,
                             local, hp_tree, hp, res)
            side["ingest"] = ingest_leg(eng, synth, torch, local, t_build, rows, st_main["nnz"], tabs)
            # K9 against SURVEY 8(d)'s formula: nnz * 8 B * 2 (read + write) * passes of the sort over the postings
            k9_passes = int(st_main.get("sort_passes", 0)) or 4
            k9_bytes = float(st_main["nnz"]) * 8.0 * 2.0 * k9_passes
            side["ingest"]["index_build_1e9" if rows == WORKLOADS["cfg3"]["rows"] else "index_build"] = dict(
                kernel="K9 build_sparse_index (spbuild.hip): term-major postings + per-(term, segment) offsets from the "
                       "document-major CSR", postings=st_main["nnz"], ms=k9_ms, first_build_of_the_process_ms=k9_cold_ms,
                sort_passes=k9_passes,
                alg_gb=k9_bytes / 1e9, gbs=k9_bytes / k9_ms / 1e6, frac_of_hbm_peak=k9_bytes / k9_ms / 1e6 / PEAK_HBM_GBS,
                bound="hbm", peak_gbs=PEAK_HBM_GBS, chunks_per_sec=rows / (k9_ms / 1e3),
                note="HIP events around the SECOND build of the timed index (hx_rebuild_sparse; generation of the synthetic "
                     "corpus excluded; the first build also pays for its temporary allocations); bytes = nnz * 8 * 2 * sort "
                     "passes (SURVEY 8d); the sort is rocPRIM's 8-bit onesweep over the 31-bit term ids, 4 passes")
            if mode == "h1":
                try:
                    side.update(boundary_legs(eng, torch, ix, Q, (qip, qix, qv), P, local))
                except Exception as e:      # noqa: BLE001
                    side["boundary_error"] = repr(e)[:300]
            if mode == "h1" and args.workload == "cfg3" and not args.rows:
                # BASELINE config 4's per-GPU shape on this one GPU: the 10M index is freed first
                ix.close()
                side["cfg4_shard"] = cfg4_shard_leg(eng, synth, torch, local, tabs, Q, (qip_d, qix_d, qv_d), hp, B, sel,
                                                    dim, not args.no_cpu_baseline, share_note, share)

    if rank == 0:
        st = st_main
        line = {
            "metric": "queries/sec, 10M x 768 hybrid dense+BM25 (RRF top-10)"
                      if (args.workload == "cfg3" and rows == WORKLOADS["cfg3"]["rows"] and B == WORKLOADS["cfg3"]["batch"])
                      else f"queries/sec, {wl['desc']} [rows={rows}, batch={B}]",
            "value": B * args.steps / dt, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "rows": rows, "dim": dim, "batch": B, "top_k": 10,
                       "arithmetic": "dense: " + (DENSE_ARITH if st["cand8_queries"] else "fp16 MFMA candidate scan + exact fp32 "
                                                  "re-score (certified)") + "; sparse: " + SPARSE_ARITH,
                       "dense_candidates": {"kind": "int8" if st["cand8_queries"] else "fp16",
                                            "queries": st["cand8_queries"], "uncertified_queries": st["cand8_uncertified_queries"],
                                            "largest_row_quantisation_error": st["cand8_row_error_max"],
                                            "int8_copy_gb": st["bytes_i8_cand"] / 1e9},
                       "sharding": f"rows/{world}", "nnz_per_shard": st["nnz"],
                       # N > 1: the exchange + fusion of batch i overlap the local stage of batch i + 1
                       "batches_in_flight": 2 if pipe is not None else 1,
                       **({"pipeline_fallback": pipe_note} if pipe_note else {}),
                       "exact_fallback_queries": st["dense_fallback_queries"], "retry_queries": st["retry_queries"],
                       "sparse_fallback_queries": st["sparse_fallback_queries"], "build_s": round(t_build, 2),
                       **({"sharded_equals_single_index": verified} if verified is not None else {})},
            # recall@10 of the LAST TIMED STEP's lists against the host brute force over the whole corpus
            "recall_at_10": cpu["recall_at_10"] if cpu else None,
            "roofline": roof, "cpu_baseline": cpu,
            "secondary": side if side is not None else ({"tree_mode_sharded": tree_sharded} if tree_sharded else None),
        }
        print(json.dumps(line), flush=True)
    ix.close()      # (idempotent: the config 4 leg may have freed it already)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
