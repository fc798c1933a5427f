"""Diagnostic: the dense stage between the bandwidth-bound small batches and the 256-wide matrix-bound kernel
(33 <= B <= 255) on 10M x 768: ms per pass, the candidate scan alone (HIP events) against the HBM roofline.
argv: rows [batches...]"""
import sys, os, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Bs = [int(x) for x in sys.argv[2:]] or [1, 32, 33, 48, 64, 65, 96, 128, 129, 192, 255, 256]
D = 768
ix = eng.HxIndex(D, (64,))
ix.synth_fill(N, synth.SEED_CORPUS)
for B in Bs:
    Q = eng.synth_queries_dense(D, 0, B, synth.SEED_QUERY)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    f = lambda: ix.search_dense(Q, 10, flag=flag)
    f(); f(); torch.cuda.synchronize()
    ix.profile(True); ix.profile_read()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    p = ix.profile_read()["scan_cand8"]; ix.profile(False)
    alg = N * D + B * D                       # bytes of the scanned int8 copy per pass
    print(json.dumps(dict(B=B, ms_per_pass=round(ms, 3), scan_ms=round(p["ms"] / 5, 3), launches=p["launches"] // 5,
                          scan_frac_of_hbm=round(alg / (p["ms"] / 5) / 1e6 / 8000.0, 3) if p["ms"] else None,
                          pass_frac_of_hbm=round(alg / ms / 1e6 / 8000.0, 3), flagged=int(flag.item()))), flush=True)
