"""Dev tool: the gate-residual GEMMs of a DiT block with staggered workgroup starts (FLEXAM_GEMM_STAGGER = ticks of 10 ns per step of
blockIdx / 8, read per launch), same process, same buffers, values alternated.  Prints the median time per value and shape."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


M = 23296
x = torch.randn(M, 3072, device=dev)
gate = torch.randn(4, 3072, device=dev)
rows = torch.randint(0, 4, (M,), dtype=torch.int32, device=dev)
values = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,20,40,60,80,120,160".split(","))]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for name, N, K in (("ffn2", 3072, 14336), ("oproj", 3072, 3072)):
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    fn = lambda: H.gemm_gate_residual(a, w, b, x, gate=gate, gate_row=rows)
    res = {v: [] for v in values}
    for r in range(rounds):
        for v in values:
            os.environ["FLEXAM_GEMM_STAGGER"] = str(v)
            res[v].append(timeit(fn) * 1e6)
    base = statistics.median(res[values[0]])
    for v in values:
        m = statistics.median(res[v])
        print(f"{name:6s} stagger {v:4d} ticks/step (spread {v * 31 / 100:5.1f} us): {m:7.1f} us  {2.0 * M * N * K / m / 1e6:7.1f} TF/s  x{base / m:.3f}   ({min(res[v]):.1f}-{max(res[v]):.1f})", flush=True)
os.environ.pop("FLEXAM_GEMM_STAGGER", None)
