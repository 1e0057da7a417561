"""Dev tool: epilogue cost of the N = 3072 GEMMs: the same tile grid with K = 64 (one K block, almost no MFMA work) and K = 3072,
bf16 store vs fp32 gate-residual read-modify-write."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, N = 23296, 3072


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


x = torch.randn(M, N, device=dev)
gate = torch.randn(4, N, device=dev)
rows = torch.randint(0, 4, (M,), dtype=torch.int32, device=dev)
out = torch.empty(M, N, dtype=BF, device=dev)
for K in (64, 3072, 14336):
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.5).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    t0 = timeit(lambda: H.gemm(a, w, b, out=out))
    t2 = timeit(lambda: H.gemm_gate_residual(a, w, b, x, gate, rows))
    t3 = timeit(lambda: H.gemm_gate_residual(a, w, b, x))
    print(f"K={K:5d}: bf16 store {t0 * 1e6:7.1f} us | gate-residual (per-row gate) {t2 * 1e6:7.1f} us | residual only {t3 * 1e6:7.1f} us", flush=True)
