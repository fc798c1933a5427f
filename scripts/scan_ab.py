"""Diagnostic: dense top-100 step time + scan kernel time/TFLOPs (HIP events) for the library in HX_LIB_PATH.  argv: rows batch
Also checks the lists of the first 64 queries against the default library's lists when HX_AB_REF is a file to write/read."""
import sys, os, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ix = eng.HxIndex(768, (64,))
ix.synth_fill(N, synth.SEED_CORPUS)
Q = eng.synth_queries_dense(768, 0, B, synth.SEED_QUERY)
for _ in range(3): k, c = ix.search_dense(Q, 100)
torch.cuda.synchronize(); ix.profile(True); ix.profile_read(); t0 = time.perf_counter()
for _ in range(10): k, c = ix.search_dense(Q, 100)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
p = ix.profile_read()["scan_f16"]
ref = os.environ.get("HX_AB_REF")
same = None
if ref:
    if os.path.exists(ref): same = bool(torch.equal(torch.load(ref), k.cpu()))
    else: torch.save(k.cpu(), ref)
print(json.dumps({"lib": os.environ.get("HX_LIB_PATH", "default"), "dense_step_ms": dt * 1e3, "scan_ms_per_step": p["ms"] / 10,
                  "tflops": p["flops"] / p["ms"] / 1e9, "same_lists_as_ref": same, "stats": {k2: v for k2, v in ix.stats().items() if "fallback" in k2 or "retry" in k2}}))
