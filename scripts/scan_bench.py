"""Diagnostic: scan-kernel throughput (HIP events around k_scan launches)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
ix = eng.HxIndex(D, (64,))
ix.synth_fill(N, synth.SEED_CORPUS)
for B in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1024", "256", "32"])]:
    Q = eng.synth_queries_dense(D, 0, B, synth.SEED_QUERY)
    for name, fn in (("f16 top100", lambda: ix.search_dense(Q, 100)), ("i8 top100", lambda: ix.search_i8(Q, 100)),
                     ("f16 m64 top100", lambda: ix.search_dense(Q, 100, 64))):
        fn(); fn(); torch.cuda.synchronize()
        ix.profile(True); ix.profile_read()
        t0 = time.time()
        for _ in range(3): fn()
        torch.cuda.synchronize(); ms = (time.time() - t0) / 3 * 1e3
        p = ix.profile_read(); ix.profile(False)
        k = p["scan_i8"] if name.startswith("i8") else p["scan_f16"]
        print(f"B={B:5d} {name:15s} total {ms:7.3f} ms | scan {k['ms']/3:7.3f} ms  {k['flops']/k['ms']/1e9:7.1f} TFLOP/s  {k['bytes']/k['ms']/1e6:7.1f} GB/s  launches {k['launches']//3}", flush=True)
