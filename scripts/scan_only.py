"""Diagnostic: a few fp16 scans (for rocprofv3 --pmc runs).  argv: rows dim batch reps"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from rag_application_amd import engine as eng, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ix = eng.HxIndex(D, (64,))
ix.synth_fill(N, synth.SEED_CORPUS)
Q = eng.synth_queries_dense(D, 0, B, synth.SEED_QUERY)
for _ in range(R):
    ix.search_dense(Q, 100)
torch.cuda.synchronize()
