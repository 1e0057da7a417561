"""Dev tool: the DiT block's big kernels at the per-rank shapes of N = 1, 2, 4, 8 GPUs (M = 2*11648/N rows; CFG rows split
first, then token chunks), communication ignored: what tile quantisation alone does to strong scaling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d, f = 11648, 3072, 14336


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


w = {k: (torch.randn(n_, k_, generator=g) * 0.5).to(BF).to(dev) for k, (n_, k_) in
     {"qkv": (3 * d, d), "o": (d, d), "f1": (f, d), "f2": (d, f)}.items()}
kv = (torch.randn(1, L, 2 * d, generator=g) * 0.5).to(BF).to(dev)
base = None
for n in (1, 2, 4, 8):
    M = 2 * L // n
    b, lq = (2, L) if n == 1 else (1, 2 * L // n)
    a = (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev)
    af = (torch.randn(M, f, generator=g) * 0.5).to(BF).to(dev)
    x = torch.randn(M, d, device=dev)
    t = {}
    # outputs preallocated like the engine's workspaces.  (The shapes run back to back, largest first, so the small ones are timed on
    # a chip that is already at its power limit -- as they would be in a real N-rank run; a fresh process that times only M = 2912
    # reads 20-25 % less for the same launches: tools/ab_env.py with FLEXAM_AB_M=2912.)
    oq, of_ = torch.empty(M, 3 * d, dtype=BF, device=dev), torch.empty(M, f, dtype=BF, device=dev)
    t["qkv"] = timeit(lambda: H.gemm(a, w["qkv"], out=oq))
    t["ffn1"] = timeit(lambda: H.gemm(a, w["f1"], out=of_, epilogue=H.EPI_GELU_TANH))
    t["ffn2"] = timeit(lambda: H.gemm_gate_residual(af, w["f2"], None, x))
    t["o x3"] = 3 * timeit(lambda: H.gemm_gate_residual(a, w["o"], None, x))
    q = a.view(b, lq, 24, 128)
    k, v = (kv.expand(b, L, 2 * d)[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(2))
    o = torch.empty(b, lq, 24, 128, dtype=BF, device=dev)
    t["attn"] = timeit(lambda: H.attn_fwd(q, k, v, out=o), 5)
    tot = sum(t.values())
    base = base or tot
    print(f"N={n} M={M:6d}: " + "  ".join(f"{k} {v * 1e3:6.3f}" for k, v in t.items()) + f"  | block {tot * 1e3:6.3f} ms  scaling eff {base / (n * tot) * 100:5.1f} %", flush=True)
