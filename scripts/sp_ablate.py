"""Diagnostic: time k_sparse_score with parts of the loop ablated (results are wrong, timing only)."""
import os, sys, subprocess, json
sys.path.insert(0, '.')
from rag_application_amd import build
code = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rag_application_amd import engine as eng, synth
tabs = synth.tables()
ix = eng.HxIndex(768, (64,)); ix.synth_fill(2000000, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, 1024, tabs)
t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
for _ in range(2): ix.search_sparse(*t, 100)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): ix.search_sparse(*t, 100)
torch.cuda.synchronize(); print("%.3f" % ((time.time() - t0) / 3 * 1e3))
"""
for mask in (0, 1, 2, 3, 4, 8, 12, 15):
    lib = build.build(defines=(f"HX_SP_ABLATE={mask}",), lib=f"/tmp/libhx_abl{mask}.so", objdir=f"/tmp/hx_abl{mask}")
    env = dict(os.environ, HX_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print("ablate mask", mask, "-> sparse ms", r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else "", flush=True)
