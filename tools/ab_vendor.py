"""Dev tool: this library's GEMMs against the vendor path behind torch (hipBLASLt / rocBLAS) on the DiT shapes, round-robin medians in
one process (sequential timings are not comparable on this part).  The vendor calls have no fused epilogue: plain bf16 GEMM (+ bias)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M = 23296
std_w = float(os.environ.get("FLEXAM_AB_WSTD", "0.05"))
for name, N, K in (("qkv", 9216, 3072), ("ffn1", 14336, 3072), ("ffn2", 3072, 14336), ("oproj", 3072, 3072)):
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * std_w).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    bb = b.to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    arms = {"flexam gemm+bias": lambda: H.gemm(a, w, b, out=out),
            "torch F.linear": lambda: F.linear(a, w),
            "torch F.linear+bias": lambda: F.linear(a, w, bb),
            "torch.mm out=": lambda: torch.mm(a, w.t(), out=out)}
    res = {k: [] for k in arms}
    names = list(arms)
    for rnd in range(7):
        for k in (names if rnd % 2 == 0 else names[::-1]):
            arms[k](); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6): arms[k]()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 6)
    fl = 2.0 * M * N * K
    print(f"{name:6s} N={N:5d} K={K:5d}: " + "  ".join(f"{k}: {fl / statistics.median(v) / 1e12:6.0f} TF/s" for k, v in res.items()), flush=True)
