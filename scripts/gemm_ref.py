"""Diagnostic: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for the scan's shape
(rows x 768 fp16) @ (768 x 1024 fp16), next to a large-K GEMM.  The scan kernel does the same contraction with a
threshold epilogue instead of writing the product."""
import torch, time, json
def run(M, N, K, dt=torch.float16, n=20):
    a = torch.randn(M, K, device="cuda", dtype=dt); b = torch.randn(K, N, device="cuda", dtype=dt)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): c = a @ b
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / n
    return dict(M=M, N=N, K=K, ms=dtm * 1e3, tflops=2.0 * M * N * K / dtm / 1e12)
out = [run(1 << 20, 1024, 768), run(1 << 21, 1024, 768), run(1 << 20, 256, 768), run(8192, 8192, 8192), run(16384, 16384, 2048)]
for o in out: print(json.dumps(o))
