"""Diagnostic: the torch.distributed calls of the N > 1 path against RCCL itself, with the one rank a one-GPU box
allows (the collectives degenerate to copies, but the process group, the host-side header group, the collective on a
side stream and the pinned flag read are the real code paths)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hdr = dist.new_group(backend="gloo")
head = torch.tensor([1024, 768, 7000], dtype=torch.int64)
dist.broadcast(head, 0, group=hdr)
buf = torch.arange(1 << 20, dtype=torch.uint8, device="cuda")
dist.broadcast(buf, 0)
from rag_application_amd import engine as eng, synth
from rag_application_amd.distributed import ShardedIndex, H1Pipeline
tabs = synth.tables()
ix = eng.HxIndex(768, (64, 128, 256)); ix.synth_fill(200000, synth.SEED_CORPUS, synth.SEED_SPDOC, tabs); ix.finalize()
Q = eng.synth_queries_dense(768, 0, 256, synth.SEED_QUERY)
qip, qix, qv = (torch.from_numpy(a).cuda() for a in synth.sparse_queries(synth.SEED_SPQUERY, 0, 256, tabs))
sh = ShardedIndex(ix)
real_gather = sh.gather_raw
def gather(keys):      # world is 1: call the collective all the same
    out = torch.empty_like(keys)
    dist.all_gather_into_tensor(out, keys.contiguous())
    return out
sh.gather_raw = gather
sh.world = 1
pipe = H1Pipeline(sh, 100, 100, 10, force_side_stream=True)
orig = pipe._exchange_and_fuse
def ex(mine, B):       # (world == 1 skips the gather inside the pipeline: put it back for this probe)
    return orig(gather(mine), B)
pipe._exchange_and_fuse = ex
outs = [pipe.submit(Q, qip, qix, qv) for _ in range(4)]
pipe.wait(); torch.cuda.synchronize()
hp = eng.make_params(dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=100,
                          quantized_limit=40, sparse_limit=100, final_limit=10, hnsw_ef=128), mode=eng.HX_MODE_H1)
k1, c1 = ix.hybrid_query(Q, qip, qix, qv, hp)
print("rccl one-rank probe: pipeline equals the single call:", all(torch.equal(o[0], k1) and torch.equal(o[1], c1) for o in outs),
      "redone", pipe.redone)
dist.destroy_process_group()
