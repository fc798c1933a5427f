"""Diagnostic: per-phase cycle shares of k_sparse_score (builds a -DHX_SP_STAMP library)."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
from rag_application_amd import build
lib = build.build(defines=("HX_SP_STAMP",), lib="/tmp/libhx_stamp.so", objdir="/tmp/hx_stamp_obj")
os.environ["HX_LIB_PATH"] = lib
import torch
from rag_application_amd import engine as eng, synth, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
B = 1024
tabs = synth.tables()
ix = eng.HxIndex(768, (64,))
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
for _ in range(2): ix.search_sparse(*t, 100)
torch.cuda.synchronize()
import time; t0=time.time(); ix.search_sparse(*t, 100); torch.cuda.synchronize(); print("sparse ms", (time.time()-t0)*1e3, "segments", ix.stats()["n_segments"])
buf = np.zeros(8*1024, np.uint64)
_lib.lib().hx_debug_sp_stamps.argtypes=[ctypes.c_void_p, ctypes.c_int]
rc = _lib.lib().hx_debug_sp_stamps(buf.ctypes.data, 8*1024); assert rc == 0
st = buf.reshape(1024, 8).astype(np.float64)
names = ["wait-slots", "make-room", "adds(+tails)", "probe+publish", "barrier X", "load-issue", "harvest+Y", "-"]
nseg = ix.stats()["n_segments"]
tot = st.sum(1)
print("cycles per visit: mean %.0f  p50 %.0f  max %.0f" % (tot.mean()/nseg, np.median(tot)/nseg, tot.max()/nseg))
for i, n in enumerate(names[:7]):
    print("%-16s mean %.0f  max-block %.0f  (%.1f%%)" % (n, st[:, i].mean()/nseg, st[:, i].max()/nseg, 100*st[:, i].sum()/tot.sum()))
for i, n in enumerate(names[:7]):
    v = st[:, i] / nseg
    print("%-16s p10 %.0f p50 %.0f p90 %.0f p99 %.0f" % (n, *np.percentile(v, [10, 50, 90, 99])))
T = np.diff(qip)
# postings per visit per query (from the CSR of the corpus is not available here): use T as proxy
for tt in range(3, 13):
    m = T == tt
    if m.any(): print("T=%2d  n=%3d  cycles/visit mean %.0f  adds %.0f probe %.0f harvest %.0f" % (tt, m.sum(), tot[m].mean()/nseg, st[m,2].mean()/nseg, st[m,3].mean()/nseg, st[m,6].mean()/nseg))
order = np.argsort(tot)
print("slowest blocks: T=", T[order[-5:]], "cycles/visit", (tot[order[-5:]]/nseg).astype(int))
print("fastest blocks: T=", T[order[:5]], "cycles/visit", (tot[order[:5]]/nseg).astype(int))
