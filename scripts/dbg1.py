import numpy as np, sys
sys.path.insert(0,'.')
from oracle import oracle as O
from rag_application_amd import engine as eng
n, dim = 300, 768
rng = np.random.default_rng(7)
scale = rng.uniform(0.01, 3.0, n).astype(np.float32); scale[:50]=1.0
X = O.synth_dense(O.SEED_CORPUS, 0, n, dim)
X[:25] = O.cosine_preprocess(X[:25]); X[25]=0; X[26,:]=0; X[26,5]=1.0
X = (X*scale[:,None]).astype(np.float32)
ora = O.OracleIndex(dim,(64,128,256)); ora.add(X); ora.finalize()
ix = eng.HxIndex(dim,(64,128,256)); ix.add(X)
bad = {0:[],1:[],2:[],3:[],4:[]}
for r in range(n):
    if not np.array_equal(ix.debug_row(0,r).view(np.uint32), ora.dense[r].view(np.uint32)): bad[0].append(r)
    for w,m in enumerate((64,128,256)):
        if not np.array_equal(ix.debug_row(w+1,r).view(np.uint32), ora.prefix[m][r].view(np.uint32)): bad[w+1].append(r)
    if not np.array_equal(ix.debug_row(4,r), ora.q8[r]): bad[4].append(r)
print({k:(len(v), v[:10]) for k,v in bad.items()})
for w,m in ((3,256),(2,128),(1,64),(0,768)):
    for r in bad[w][:3]:
        x = X[r,:m]; g = ix.debug_row(w,r) if w else ix.debug_row(0,r)
        n2 = O.spec_dot(x[None,:], x)[0]
        # infer GPU len from ratio
        print(m, r, "n2", n2, float(n2).hex(), "oracle ln", np.sqrt(n2,dtype=np.float32), "gpu implied ln", (x[0]/g[0]), "scale", scale[r], "absdiff1", abs(n2-1))
