"""Dev tool: flexam_gemm_bf16 vs the vendor GEMM behind torch over output widths (M = 23296, K = 3072), round-robin medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, K = int(os.environ.get("FLEXAM_AB_M", "23296")), int(os.environ.get("FLEXAM_AB_K", "3072"))
a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
for N in (3072, 6144, 8192, 9216, 10240, 12288, 14336, 16384):
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    arms = {"flexam": lambda: H.gemm(a, w, b, out=out), "vendor": lambda: torch.mm(a, w.t(), out=out)}
    res = {k: [] for k in arms}
    for rnd in range(7):
        for k in (list(arms) if rnd % 2 == 0 else list(arms)[::-1]):
            arms[k](); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6): arms[k]()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 6)
    fl = 2.0 * M * N * K
    tiles = ((M + 255) // 256) * (N // 256)
    f_, v_ = (fl / statistics.median(res[k]) / 1e12 for k in ("flexam", "vendor"))
    print(f"N={N:6d} tiles {tiles:5d} = {tiles / 256:6.2f} rounds: flexam {f_:6.0f} TF/s  vendor {v_:6.0f} TF/s  flexam/vendor {f_ / v_:.3f}", flush=True)
