"""MaxScore on this collection: with tau = the exact L-th best sparse score of a query, which of its
terms are NON-ESSENTIAL (ascending by upper bound ub_t = q_t * max_d w_{t,d}; the longest prefix whose bounds sum below
tau: a document that holds only such terms cannot reach tau), what share of the query's postings lies in their lists,
and how many documents a select pass over the ESSENTIAL lists alone would have to hand to the exact pass
(partial score >= L-th best partial score - sum of the non-essential bounds).

CPU only, the oracle's C generator.  N documents, L = the same quantile as top-100 of 10M.
    python scripts/maxscore_probe.py [n_docs=1000000] [n_queries=256]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as CO, oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = max(1, round(100 * n / 10_000_000))
tabs = O.synth_tables()
t0 = time.time()
ip, ix, v = CO.synth_sparse_docs(O.SEED_SPDOC, 0, n, tabs)
doc = np.repeat(np.arange(n, dtype=np.int32), np.diff(ip))
o = np.argsort(ix, kind="stable")
tix, tdoc, tv = ix[o], doc[o], v[o]
ut, start = np.unique(tix, return_index=True)
end = np.append(start[1:], len(tix))
print(f"corpus: {n} docs, {len(ix)} postings, {len(ut)} live terms, {time.time() - t0:.1f} s", flush=True)
qip, qix, qv = O.synth_sparse_queries(O.SEED_SPQUERY, 0, nq, tabs)
tot_post = ne_post = 0
shares, cand_full, cand_ess, n_terms, n_ne = [], [], [], [], []
for b in range(nq):
    terms, w = qix[qip[b]:qip[b + 1]], qv[qip[b]:qip[b + 1]]
    lists = []
    for t, qw in zip(terms, w):
        k = np.searchsorted(ut, t)
        if k < len(ut) and ut[k] == t:
            lists.append((qw, tdoc[start[k]:end[k]], tv[start[k]:end[k]]))
    if not lists:
        continue
    score = np.zeros(n, np.float64)
    for qw, d, wv in lists:
        score[d] += qw * wv.astype(np.float64)
    touched = np.flatnonzero(score)
    if len(touched) < L:
        continue
    tau = np.partition(score[touched], -L)[-L]
    ub = np.array([qw * wv.max() for qw, d, wv in lists])
    df = np.array([len(d) for qw, d, wv in lists])
    order = np.argsort(ub, kind="stable")
    cs = np.cumsum(ub[order])
    k_ne = int(np.searchsorted(cs, tau, side="left"))          # prefix sums strictly below tau
    ne = order[:k_ne]
    tot_post += df.sum()
    ne_post += df[ne].sum()
    shares.append(df[ne].sum() / df.sum())
    n_terms.append(len(lists))
    n_ne.append(k_ne)
    # what an essential-lists-only select pass must keep
    part = np.zeros(n, np.float64)
    for j in order[k_ne:]:
        qw, d, wv = lists[j]
        part[d] += qw * wv.astype(np.float64)
    pt = np.flatnonzero(part)
    pL = np.partition(part[pt], -L)[-L] if len(pt) >= L else 0.0
    cand_ess.append(int((part[pt] >= pL - ub[ne].sum()).sum()))
    cand_full.append(int((score[touched] >= tau).sum()))
res = dict(n_docs=n, L=L, queries=len(shares), postings_of_the_queries=int(tot_post),
           share_of_postings_in_non_essential_lists=float(ne_post / tot_post),
           per_query_share=dict(mean=float(np.mean(shares)), median=float(np.median(shares)), p90=float(np.quantile(shares, 0.9)),
                                zero=float(np.mean(np.array(shares) == 0))),
           terms_per_query=float(np.mean(n_terms)), non_essential_terms_per_query=float(np.mean(n_ne)),
           candidates_full_pass=dict(median=float(np.median(cand_full)), p90=float(np.quantile(cand_full, 0.9))),
           candidates_essential_only_pass=dict(median=float(np.median(cand_ess)), p90=float(np.quantile(cand_ess, 0.9)),
                                               max=int(np.max(cand_ess))))
print(json.dumps(res, indent=1))
