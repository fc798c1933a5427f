"""Diagnostic: per-phase cycle shares of k_sparse_select (needs the -DHX_SP_STAMP library built in-tree:
python -c "from rag_application_amd import build; build.build(defines=('HX_SP_STAMP',), lib='rag_application_amd/csrc/build/libhx_stamp.so', objdir='rag_application_amd/csrc/build/stamp')")"""
import os, sys, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HX_LIB_PATH"] = os.path.abspath("rag_application_amd/csrc/build/libhx_stamp.so")
os.environ["HX_DEBUG_SEG_DOCS"] = "65536"
import torch
from rag_application_amd import engine as eng, synth, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = 1024
tabs = synth.tables()
ix = eng.HxIndex(64, ())
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
qip, qix, qv = synth.sparse_queries(synth.SEED_SPQUERY, 0, B, tabs)
CUT = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # drop query terms more frequent than this rank
if CUT:
    rank = (qix.astype(np.int64) * pow(0x9E3779B1, -1, 1 << 31)) & 0x7FFFFFFF
    keep = rank >= CUT
    ip = np.zeros(B + 1, np.int64)
    ip[1:] = np.cumsum([int(keep[qip[b]:qip[b + 1]].sum()) for b in range(B)])
    qip, qix, qv = ip, qix[keep], qv[keep]
t = [torch.from_numpy(a).cuda() for a in (qip, qix, qv)]
for _ in range(2): ix.search_sparse(*t, 100)
torch.cuda.synchronize()
import time; t0 = time.time(); ix.search_sparse(*t, 100); torch.cuda.synchronize()
nseg = ix.stats()["n_segments"]
print("sparse ms", (time.time() - t0) * 1e3, "segments", nseg)
buf = np.zeros(2 * 8 * 1024, np.uint64)
_lib.lib().hx_debug_sp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = _lib.lib().hx_debug_sp_stamps(buf.ctypes.data, 2 * 8 * 1024); assert rc == 0
st = buf.reshape(1024, 2, 8).astype(np.float64)
names = ["loop+cut check", "dir+issue(s+2)", "offsets+accumulate", "barrier X", "harvest", "barrier Y", "cuts (sort + threshold)", "output"]
raw = buf.reshape(1024, 2, 8)
cuts = (raw[:, 0, 6] >> np.uint64(40)).astype(np.float64)
st[:, :, 6] = (raw[:, :, 6] & np.uint64((1 << 40) - 1)).astype(np.float64)
print("cuts per workgroup: mean %.1f  p10 %.0f  p90 %.0f  max %.0f;  cycles per cut %.0f" % (cuts.mean(), np.percentile(cuts, 10), np.percentile(cuts, 90), cuts.max(), st[:, 0, 6].sum() / max(cuts.sum(), 1)))
for w, wn in ((0, "wave 0"), (1, "wave 5")):
    s = st[:, w, :8]
    tot = s.sum(1)
    print(wn, "cycles per visit: mean %.0f  p50 %.0f  max %.0f  (s_memtime = shader cycles)" % (tot.mean() / nseg, np.median(tot) / nseg, tot.max() / nseg))
    for i, n in enumerate(names):
        print("  %-20s mean %.1f  p10 %.1f p90 %.1f  (%.1f%%)" % (n, s[:, i].mean() / nseg, np.percentile(s[:, i], 10) / nseg, np.percentile(s[:, i], 90) / nseg, 100 * s[:, i].sum() / tot.sum()))
