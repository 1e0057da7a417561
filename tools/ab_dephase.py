"""Dev tool: does running ONE GEMM as two concurrent launches with different tile heights (their epilogues drift against each other
instead of all 256 workgroups storing at once) beat the single lockstep launch?  Rows split at `cut`; part A: MT = 8 on stream 1, part B:
MT = 7 (or 6) on stream 2, each planned for half of the CUs.  Round-robin medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M = 23296
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run_split(fn, cut, mt_b):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    H.set_cu_budget(128)
    try:
        for s, lo, hi, mt in ((s1, 0, cut, "8"), (s2, cut, M, mt_b)):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                os.environ["FLEXAM_GEMM_MT"] = mt
                fn(lo, hi)
                e = torch.cuda.Event(); e.record(s)
            main.wait_event(e)
    finally:
        H.set_cu_budget(0)
        os.environ.pop("FLEXAM_GEMM_MT", None)


for name, N, K, epi in (("qkv", 9216, 3072, "none"), ("ffn1", 14336, 3072, "gelu"), ("oproj", 3072, 3072, "gate"), ("ffn2", 3072, 14336, "gate")):
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    x = torch.zeros(M, N, device=dev)
    if epi == "gate":
        fn = lambda lo, hi: H.gemm_gate_residual(a[lo:hi], w, b, x[lo:hi])
    else:
        fn = lambda lo, hi: H.gemm(a[lo:hi], w, b, out=out[lo:hi], epilogue=H.EPI_GELU_TANH if epi == "gelu" else H.EPI_NONE)
    arms = {"single": lambda: fn(0, M)}
    for cut_tiles, mt_b in ((48, "7"), (46, "7"), (50, "6"), (45, "8")):
        cut = cut_tiles * 256
        arms[f"split@{cut_tiles} MT8|MT{mt_b}"] = (lambda c=cut, m=mt_b: run_split(fn, c, m))
    res = {k: [] for k in arms}
    names = list(arms)
    for rnd in range(7):
        for k in (names if rnd % 2 == 0 else names[::-1]):
            arms[k](); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): arms[k]()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 5)
    base = statistics.median(res["single"])
    fl = 2.0 * M * N * K
    print(f"{name:6s}: " + "  ".join(f"{k}: {statistics.median(v) * 1e6:7.1f} us ({100 * (base / statistics.median(v) - 1):+.1f}%)" for k, v in res.items()), flush=True)
