"""Diagnostic: the dense stage on a NON-exchangeable row order.  The synthetic corpus is i.i.d.; a real one
is ingested document by document, i.e. topically clustered.  Rows are re-ordered so that the rows most similar to a few
"topic" queries sit together in contiguous runs (placed early, in the middle, or last in scan order); reports the step
time of dense top-100 for a batch that contains those topic queries, the retry / exact-fallback counters, and checks the
lists against the i.i.d.-order index (same rows, ids mapped back).  argv: rows [batch]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D, L, NT = 768, 100, 8                      # NT topic queries, each with a cluster of CL rows
CL = 20000
X = CO.synth_dense(synth.SEED_CORPUS, 0, N, D)
Q = CO.synth_dense(synth.SEED_QUERY, 0, B, D)
Xt = torch.from_numpy(X).cuda()
sc = (Xt @ torch.from_numpy(Q[:NT]).cuda().T).cpu().numpy()      # raw scores are enough to pick the clusters
del Xt
used = np.zeros(N, bool)
clusters = []
for t in range(NT):
    top = np.argsort(-sc[:, t])
    top = top[~used[top]][:CL]
    used[top] = True
    clusters.append(top[::-1].copy())       # ascending similarity inside the run: every row beats the ones before it
rest = np.nonzero(~used)[0]
def run(name, perm):
    ix = eng.HxIndex(D, (64,))
    for a in range(0, N, 250000):
        ix.add(X[perm[a:a + 250000]])
    q = torch.from_numpy(Q).cuda()
    for _ in range(2): k, c = ix.search_dense(q, L)
    torch.cuda.synchronize(); s0 = ix.stats(); t0 = time.perf_counter()
    for _ in range(3): k, c = ix.search_dense(q, L)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    s1 = ix.stats()
    sco, ids = eng.unpack(k)
    ids = perm[ids.cpu().numpy().clip(0)]                 # back to original row numbers
    ix.close()
    print(f"{name:34s} {dt * 1e3:8.2f} ms/step  retries/step {(s1['retry_queries'] - s0['retry_queries']) / 3:7.1f}  "
          f"exact fallbacks/step {(s1['dense_fallback_queries'] - s0['dense_fallback_queries']) / 3:6.1f}", flush=True)
    return sco.cpu().numpy(), ids
base = run("i.i.d. order", np.arange(N))
cl = np.concatenate(clusters)
for name, perm in (("clusters first", np.concatenate([cl, rest])),
                   ("clusters in the middle", np.concatenate([rest[:len(rest) // 2], cl, rest[len(rest) // 2:]])),
                   ("clusters last", np.concatenate([rest, cl]))):
    s, i = run(name, perm)
    same = all(set(i[b].tolist()) == set(base[1][b].tolist()) for b in range(B))
    print("    same top-100 sets as the i.i.d. order:", same)
