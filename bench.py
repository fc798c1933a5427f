#!/usr/bin/env python3
"""Hybrid-retrieval benchmark: queries/sec on the 10M x 768 dense+BM25 workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4]

One "step" = one batch of B queries through the hot path with every input already
resident in HBM: per-shard cosine top-100 (fp16 MFMA scan + exact fp32 re-score,
certified) and BM25 top-100 over the on-device inverted index, per-stage all-gather of
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
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override corpus rows (testing)")
    ap.add_argument("--batch", type=int, default=0, help="override query batch (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=4_000_000)
    ap.add_argument("--cpu-queries", type=int, default=1024)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (box share per GPU)")
    return ap.parse_args()


def cpu_baseline(wl, B, dim, tabs, cpu_rows, cpu_queries, gpu_check):
    """The C restatement of the path (oracle/hx_oracle.c, kind "port") timed on this
    box's host cores on a bounded sample: `cpu_queries` queries against the first
    `cpu_rows` rows of the corpus.  Brute force is linear in rows, so the rate is scaled
    by cpu_rows / rows.  Also returns the sample's exact lists for the parity gate."""
    from oracle import c_oracle as CO
    from rag_application_amd import synth
    rows = wl["rows"]
    ns = min(cpu_rows, rows)
    bs = min(cpu_queries, B)
    threads = CO.num_threads()
    X = CO.synth_dense(synth.SEED_CORPUS, 0, ns, dim)
    Xn = CO.cosine_preprocess(X)
    del X
    Q = CO.synth_dense(synth.SEED_QUERY, 0, bs, dim)
    Qn = CO.cosine_preprocess(Q)
    inv = None
    if wl["mode"] == "h1":
        ip, ix, v = CO.synth_sparse_docs(synth.SEED_SPDOC, 0, ns, tabs)
        inv = CO.InvIndex(ip, ix, v)
        qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, bs, tabs)
    t0 = time.perf_counter()
    if wl["mode"] == "h1":
        ds, di, dc = CO.search_dense(Xn, Qn, 100)
        ss, si, sc = inv.search(qip, qix, qv, 100)
        out = [CO.rrf(di[b, :dc[b]], si[b, :sc[b]], 2.0, 0, 10) for b in range(bs)]
    else:
        ds, di, dc = CO.search_dense(Xn, Qn, 10)
        out = [(ds[b, :dc[b]], di[b, :dc[b]]) for b in range(bs)]
    dt = time.perf_counter() - t0
    ok = None
    recall = None
    if gpu_check is not None:
        ok, recall = gpu_check(ns, bs, out)
    return dict(value=bs / dt * (ns / rows), unit="queries/sec", cores=threads, kind="port",
                sample=f"{bs} of {B} queries x rows [0,{ns}) of {rows} ({dt:.2f} s of CPU work); "
                       f"brute force is linear in rows, rate scaled by {ns}/{rows}",
                parity_on_sample=ok, recall_at_10=recall)


def main():
    args = parse()
    os.environ.setdefault("OMP_NUM_THREADS", str(args.cpu_threads))
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
    ix.finalize()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    # ---- queries, resident in HBM ------------------------------------------------------
    Q = eng.synth_queries_dense(dim, 0, B, synth.SEED_QUERY, device=local)
    if mode == "h1":
        qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
        qip_d, qix_d, qv_d = (torch.from_numpy(a).to(dev) for a in (qip, qix, qv))
    P = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
             quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128)
    hp = eng.make_params(P, mode=eng.HX_MODE_H1)

    from rag_application_amd.distributed import ShardedIndex, H1Pipeline
    sh = ShardedIndex(ix)      # one process per GPU; exchange = one RCCL all-gather per stage
    # N > 1: the exchange + fusion of a batch run on a side stream beside the local stage of the next
    pipe = H1Pipeline(sh, 100, 100, 10) if (world > 1 and (backend == "nccl" or os.environ.get("HX_BENCH_PIPE"))) \
        else None

    def step():
        if mode == "h1":
            if world == 1:
                return ix.hybrid_query(Q, qip_d, qix_d, qv_d, hp)    # whole pipeline behind one ABI call
            if pipe is not None:
                return pipe.submit(Q, qip_d, qix_d, qv_d)
            return sh.hybrid_h1(Q, qip_d, qix_d, qv_d, 100, 100, 10)
        return sh.search_dense(Q, 10)

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
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ix.profile_read()
    ix.profile(False)
    # Rehearsal aid (HX_BENCH_VERIFY=1, N > 1, small --rows): rank 0 also builds the UNSHARDED corpus and
    # checks the last step's lists against it, key for key.  Never part of the timed region.
    verified = None
    if world > 1 and mode == "h1" and os.environ.get("HX_BENCH_VERIFY"):
        if rank == 0:
            one = eng.HxIndex(dim, (64, 128, 256), device=local, id_base=0)
            one.reserve(rows)
            one.synth_fill(rows, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
            k1, c1 = one.hybrid_query(Q, qip_d, qix_d, qv_d, hp)
            verified = bool(torch.equal(k1, res[0]) and torch.equal(c1, res[1]))
            one.close()
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel (fp16 MFMA scan), measured with HIP events ------
    sc = prof["scan_f16"]
    roof = None
    if sc["launches"]:
        sec = sc["ms"] / 1e3
        tf = sc["flops"] / sec / 1e12
        gbs = sc["bytes"] / sec / 1e9
        mfma_bound = (sc["flops"] / (PEAK_FP16_TFLOPS * 1e12)) >= (sc["bytes"] / (PEAK_HBM_GBS * 1e9))
        # HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE pass of this same command (gfx950
        # correction x2, MI355X_MICROARCH.md), kept under profiles/ -- counters cannot be read from
        # inside the timed process.  null when no pass for this configuration is committed.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_scan_v13.json")
        if world == 1 and os.path.exists(tpath):
            with open(tpath) as f:
                tp = json.load(f)
            if tp.get("config") == {"rows": rows, "dim": dim, "batch": B, "n_gpus": 1}:
                traffic = tp["scan_traffic_gb_per_step"] / max(sc["launches"] / args.steps, 1)
        roof = dict(kernel="k_scan8<fp16, v_mfma_f32_16x16x32_f16> (256x256 tile)" if B > 128 else "k_scan<fp16> (128-row tiles)",
                    bound="mfma" if mfma_bound else "hbm",
                    achieved=tf if mfma_bound else gbs, peak=PEAK_FP16_TFLOPS if mfma_bound else PEAK_HBM_GBS,
                    unit="TFLOP/s" if mfma_bound else "GB/s",
                    frac=(tf / PEAK_FP16_TFLOPS) if mfma_bound else (gbs / PEAK_HBM_GBS), traffic=traffic,
                    traffic_unit="GB per launch (profiles/r01_pmc_scan_v13.json)",
                    launches=sc["launches"], avg_launch_ms=sc["ms"] / sc["launches"],
                    alg_tflop_per_launch=sc["flops"] / sc["launches"] / 1e12,
                    alg_gb_per_launch=sc["bytes"] / sc["launches"] / 1e9,
                    other_bound_frac=(gbs / PEAK_HBM_GBS) if mfma_bound else (tf / PEAK_FP16_TFLOPS),
                    sparse_ms_per_step=prof["sparse"]["ms"] / max(args.steps, 1))
        sp = prof["sparse"]
        if sp["launches"] and sp["ms"] > 0:   # second kernel of the step, HBM-bound by construction
            roof["second_kernel"] = dict(kernel="k_sparse_score", bound="hbm", unit="GB/s", peak=PEAK_HBM_GBS,
                                         achieved=sp["bytes"] / sp["ms"] / 1e6,
                                         frac=sp["bytes"] / sp["ms"] / 1e6 / PEAK_HBM_GBS,
                                         alg_gb_per_launch=sp["bytes"] / sp["launches"] / 1e9,
                                         avg_launch_ms=sp["ms"] / sp["launches"],
                                         note="8 B per posting visited; latency-bound today (DESIGN.md)")

    # ---- CPU baseline + parity gate on the sample (rank 0, N = 1 only) ---------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_check(ns, bs, cpu_lists):
            six = eng.HxIndex(dim, (64, 128, 256), device=local, id_base=0)
            six.synth_fill(ns, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)
            if mode == "h1":
                qs = qip_d[: bs + 1].clone()
                nnz = int(qs[-1].item())
                k, c = six.hybrid_query(Q[:bs], qs, qix_d[:nnz], qv_d[:nnz], hp)
            else:
                k, c = six.search_dense(Q[:bs], 10)
            s, i = eng.unpack(k)
            s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
            six.close()
            ok, hit, want = True, 0, 0
            for b in range(bs):
                es, ei = cpu_lists[b]
                ok &= int(c[b]) == len(ei) and np.array_equal(i[b, :len(ei)], ei) and \
                    np.array_equal(s[b, :len(ei)].view(np.uint32), np.asarray(es, np.float32).view(np.uint32))
                hit += len(np.intersect1d(i[b, :int(c[b])], ei))      # recall@10 vs the brute-force lists
                want += len(ei)
            return bool(ok), (hit / want if want else None)
        cpu = cpu_baseline(wl, B, dim, tabs, args.cpu_rows, args.cpu_queries, gpu_check)

    if rank == 0:
        st = ix.stats()
        line = {
            "metric": "queries/sec, 10M x 768 hybrid dense+BM25 (RRF top-10)"
                      if (args.workload == "cfg3" and rows == WORKLOADS["cfg3"]["rows"] and B == WORKLOADS["cfg3"]["batch"])
                      else f"queries/sec, {wl['desc']} [rows={rows}, batch={B}]",
            "value": B * args.steps / dt, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["desc"], "rows": rows, "dim": dim, "batch": B, "top_k": 10,
                       "arithmetic": "fp16 MFMA candidate scan + exact fp32 re-score (certified); sparse: exact 2^40 "
                                     "fixed-point sums",
                       "sharding": f"rows/{world}", "nnz_per_shard": st["nnz"],
                       # N > 1: the exchange + fusion of batch i overlap the local stage of batch i + 1
                       "batches_in_flight": 2 if pipe is not None else 1,
                       "exact_fallback_queries": st["dense_fallback_queries"], "build_s": round(t_build, 2),
                       **({"sharded_equals_single_index": verified} if verified is not None else {})},
            "recall_at_10": cpu["recall_at_10"] if cpu else None,   # vs brute force on the cpu_baseline sample
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
