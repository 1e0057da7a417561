"""Dev tool: the 256 x 160 GEMM tile shape (VAE encoder's 160-channel layers) over K at M = 450k rows: slope = K loop, intercept = per-tile fixed cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N = 450560, 160
for K in (576, 1152, 2304, 4608):
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    x = torch.zeros(M, N, device=dev)
    res = []
    for name, fn in (("bf16 out", lambda: H.gemm(a, w, bias, out=out)), ("residual", lambda: H.gemm_gate_residual(a, w, bias, x))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res.append(f"{name} {dt * 1e6:7.1f} us {2.0 * M * N * K / dt / 1e12:6.0f} TF/s")
    print(f"K {K:5d}: " + "   ".join(res))
    del a
