"""Diagnostic: dense stage time against the number of scan workgroups (HX_DEBUG_SCAN8_GRID is read once per
process, so this script re-runs itself per value).  argv: rows"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("HX_SCAN_GRID_CHILD"):
    import time, torch
    sys.path.insert(0, ROOT)
    from rag_application_amd import engine as eng, synth
    N = int(sys.argv[1])
    ix = eng.HxIndex(768, (64,))
    ix.synth_fill(N, synth.SEED_CORPUS)
    Q = eng.synth_queries_dense(768, 0, 1024, synth.SEED_QUERY)
    for _ in range(2):
        ix.search_dense(Q, 100)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        ix.search_dense(Q, 100)
    torch.cuda.synchronize()
    print("grid", os.environ.get("HX_DEBUG_SCAN8_GRID"), "dense stage ms", round((time.perf_counter() - t) / 5 * 1e3, 3), flush=True)
else:
    for g in (256, 224, 192, 176, 160, 128):
        env = dict(os.environ, HX_SCAN_GRID_CHILD="1", HX_DEBUG_SCAN8_GRID=str(g))
        subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1] if len(sys.argv) > 1 else "10000000"], env=env, check=True)
