"""Side measurement for BASELINE config 5 (ingest), the "batched encode" leg: a bge-base-shaped BERT encoder
(12 layers, hidden 768, 12 heads, FFN 3072, vocab 30522; RANDOMLY INITIALISED -- no checkpoint ships and none
can be fetched) on PyTorch-ROCm, pooled the way the reference pools (UNMASKED mean of last_hidden_state,
app/core/models/huggingface/huggingface.py:165-170), its output handed to the engine's ingest (hx_add_dense_dev:
normalise, prefixes, fp16 / int8 copies) on the same GPU.  chunks/s for token-id batches already on the
device; tokenisation is host work outside this number.  argv: batch seq_len n_batches dtype(bf16|fp16|fp32)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transformers import BertConfig, BertModel
from rag_application_amd import engine as eng
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
torch.manual_seed(0)
cfg = BertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512)
model = BertModel(cfg, add_pooling_layer=False).to("cuda").to(dt).eval()
ids = torch.randint(1000, 30000, (B, S), device="cuda")
mask = torch.ones((B, S), dtype=torch.long, device="cuda")
ix = eng.HxIndex(768, (64, 128, 256)); ix.reserve(B * (NB + 3))

def step():
    with torch.no_grad():
        e = model(input_ids=ids, attention_mask=mask).last_hidden_state.mean(dim=1)   # the reference's pooling
    return e.float()

for _ in range(3):
    e = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(NB):
    e = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
# same, with the vectors appended to the index where they lie (hx_add_dense_dev)
t2 = time.perf_counter()
for _ in range(NB):
    ix.add_device(step())
torch.cuda.synchronize(); t3 = time.perf_counter()
flops = 2.0 * B * S * (12 * (4 * 768 * 768 + 2 * 768 * 3072)) + 2.0 * 12 * B * 12 * S * S * 64 * 2
print(json.dumps({"encoder": "BERT-base shape (bge-base), random init", "dtype": str(dt).split(".")[-1], "batch": B, "seq_len": S,
                  "encode_chunks_per_s": round(B * NB / (t1 - t0)), "encode_tflops": round(flops * NB / (t1 - t0) / 1e12, 1),
                  "encode_plus_ingest_chunks_per_s": round(B * NB / (t3 - t2)), "rows_in_index": ix.count()}))
