"""Side measurement for SURVEY.md 8f-3: search_across_spaces (4 spaces x N rows x 768, B queries per call,
top-k 10) -- queries/s end to end (GPU top-k + host filter/merge) and the GPU part alone."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rag_application_amd import engine as eng, synth
from rag_application_amd.scout import ScoutIndex, SPACES
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim = 768
sc = ScoutIndex(dim)
for k, name in enumerate(SPACES):          # synthetic rows generated on the device, tenants on the host
    sp = sc.spaces[name]
    sp.ix.synth_fill(N, synth.SEED_CORPUS + k)
    sp.user = ["u%d" % (r % 4) for r in range(N)]
    sp.org = ["o"] * N
    sp.props = [{}] * N
Q = eng.synth_queries_dense(dim, 0, B, synth.SEED_QUERY).cpu().numpy()
for _ in range(3): sc.search_across_spaces_batch(Q, 10, "u1", "o")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): sc.search_across_spaces_batch(Q, 10, "u1", "o")
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
qd = torch.from_numpy(Q).cuda()
t0 = time.perf_counter()
for _ in range(5):
    for name in SPACES: sc.spaces[name].ix.search_dense(qd, 10)
torch.cuda.synchronize(); gms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"spaces": 4, "rows_per_space": N, "dim": dim, "batch": B, "top_k": 10, "ms_per_call": round(ms, 3),
                  "queries_per_s": round(B / ms * 1e3, 1), "gpu_topk_ms": round(gms, 3),
                  "gpu_bytes_fp16_gb": round(4 * N * dim * 2 / 1e9, 3), "gpu_gbs": round(4 * N * dim * 2 / gms / 1e6, 1)}))
if os.environ.get("HX_SCOUT_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    sc.search_across_spaces_batch(Q, 10, "u1", "o")
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
