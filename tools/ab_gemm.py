"""Dev tool: A/B of two builds of the library in ONE process on the same buffers (box-to-box and minute-to-minute clock
differences are larger than most kernel changes).  A = tools/probes/libflexam_base.so (built by hand from another
checkout), B = the in-tree library.  Alternates A, B, A, B ... per shape and prints the median TF/s of each."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
here = os.path.dirname(os.path.abspath(__file__))
libs = {"A": os.environ.get("FLEXAM_AB_A", os.path.join(here, "probes", "libflexam_ab_base.so")), "B": os.environ.get("FLEXAM_AB_B", H.LIB_PATH)}
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def use(tag):
    H.load_library(libs[tag])          # both builds take the split-K scratch per call (flexam_hip.h version >= 2)


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


M = int(os.environ.get("FLEXAM_AB_M", "23296"))          # 2912 = one rank of eight
x = torch.randn(M, 3072, device=dev)
gate = torch.randn(4, 3072, device=dev)
rows = torch.randint(0, 4, (M,), dtype=torch.int32, device=dev)
cases = (("qkv", 9216, 3072, "bias"), ("ffn1", 14336, 3072, "gelu"), ("crossq", 3072, 3072, "bias"),
         ("ffn2", 3072, 14336, "gate"), ("oproj", 3072, 3072, "gate"))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for name, N, K, epi in cases:
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    if epi == "gate":
        fn = lambda: H.gemm_gate_residual(a, w, b, x, gate=gate, gate_row=rows)
    else:
        fn = lambda: H.gemm(a, w, b, out=out, epilogue=H.EPI_GELU_TANH if epi == "gelu" else H.EPI_NONE)
    res = {"A": [], "B": []}
    for r in range(rounds):
        for tag in ("A", "B"):
            use(tag)
            res[tag].append(2.0 * M * N * K / timeit(fn) / 1e12)
    ma, mb = statistics.median(res["A"]), statistics.median(res["B"])
    print(f"{name:7s} {epi:5s} N={N:5d} K={K:5d}:  A {ma:7.1f}  B {mb:7.1f} TF/s   B/A {mb / ma:.3f}   (A {min(res['A']):.0f}-{max(res['A']):.0f}, B {min(res['B']):.0f}-{max(res['B']):.0f})", flush=True)
