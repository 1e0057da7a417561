"""Dev tool: the block's GEMMs at the per-rank row counts of the multi-GPU layouts (2 x N/2: one CFG row, L/(N/2) tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
d, f = 3072, 14336
w = {"qkv": (3 * d, d), "oproj": (d, d), "ffn1": (f, d), "ffn2": (d, f)}
W = {k: (torch.randn(*s, device=dev) * 0.02).to(BF) for k, s in w.items()}
for M in (23296, 11648, 5824, 2912, 1456):
    line = [f"M {M:6d} (ranks {max(1, 2 * 11648 // M)}):"]
    for name, (n, k) in w.items():
        a = (torch.randn(M, k, device=dev) * 0.5).to(BF)
        bias = torch.zeros(n, device=dev)
        x = torch.zeros(M, n, device=dev)
        gate = torch.ones(1, n, device=dev)
        out = torch.empty(M, n, dtype=BF, device=dev)
        if name in ("oproj", "ffn2"):
            fn = lambda: H.gemm_gate_residual(a, W[name], bias, x, gate=gate, rows_per_batch=M)
        else:
            fn = lambda: H.gemm(a, W[name], bias, out=out, epilogue=H.EPI_GELU_TANH if name == "ffn1" else H.EPI_NONE)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        line.append(f"{name} {dt * 1e6:7.1f} us {2.0 * M * n * k / dt / 1e12:6.0f} TF/s")
    print("  ".join(line))
