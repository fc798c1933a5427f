"""Side measurement for BASELINE config 5 (ingest): on-device derivation of the stored vectors
(normalise, prefixes, fp16 / int8 copies: K1/K2) and inverted-index build (K9) for N synthetic chunks
on one GPU -- chunks/s with the raw vectors and the doc-major CSR already resident in HBM.  The
encoder (PyTorch-ROCm, no checkpoint ships) and the host -> device copy are outside this number."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tabs = synth.tables()
ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(N)
torch.cuda.synchronize(); t0 = time.perf_counter()
ix.synth_fill(N, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs)      # generate + K1/K2 + CSR append
torch.cuda.synchronize(); t1 = time.perf_counter()
ix.finalize()                                                    # K9
torch.cuda.synchronize(); t2 = time.perf_counter()
st = ix.stats()
print(json.dumps({"chunks": N, "dim": 768, "nnz": st["nnz"], "derive_s": round(t1 - t0, 3), "index_build_s": round(t2 - t1, 3),
                  "chunks_per_s": round(N / (t2 - t0)), "postings_per_s_build": round(st["nnz"] / (t2 - t1)),
                  "live_terms": st["n_groups"], "bytes_sparse_gb": round(st["bytes_sparse"] / 1e9, 2)}))
