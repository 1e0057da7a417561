"""Dev tool: one environment switch of the GEMM launcher (read per call: FLEXAM_GEMM_GM, FLEXAM_GEMM_MT ...) on the
DiT shapes, round-robin in one process, medians.  usage: ab_env.py VAR v1 v2 ..."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
if os.environ.get("FLEXAM_AB_LIB"):                         # a probe build instead of the in-tree library
    H.load_library(os.environ["FLEXAM_AB_LIB"])
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d, f = 11648, 3072, 14336
M = int(os.environ.get("FLEXAM_AB_M", str(2 * L)))          # rows: 23296 = the CFG pair on one GPU; 2912 = one of 8 ranks
r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)
x, hmid = r(M, d), r(M, f)
w_o, w_qkv, w_f1, w_f2 = r(d, d), r(3 * d, d), r(f, d), r(d, f)
out_q, out_f = torch.empty(M, 3 * d, dtype=BF, device=dev), torch.empty(M, f, dtype=BF, device=dev)
xres = torch.zeros(M, d, device=dev)
b3, bf_, bd = torch.zeros(3 * d, device=dev), torch.zeros(f, device=dev), torch.zeros(d, device=dev)
gate = torch.randn(4, d, device=dev)
rows = (torch.arange(M, device=dev) % 2).to(torch.int32)
cases = {
    "qkv": (lambda: H.gemm(x, w_qkv, b3, out=out_q), 2.0 * M * 3 * d * d),
    "cross-q": (lambda: H.gemm(x, w_o, bd, out=out_q[:, :d]), 2.0 * M * d * d),
    "ffn1 + gelu": (lambda: H.gemm(x, w_f1, bf_, out=out_f, epilogue=H.EPI_GELU_TANH), 2.0 * M * f * d),
    "ffn2 + residual": (lambda: H.gemm_gate_residual(hmid, w_f2, bd, xres, gate=gate, gate_row=rows), 2.0 * M * d * f),
    "o-proj + residual": (lambda: H.gemm_gate_residual(x, w_o, bd, xres, gate=gate, gate_row=rows), 2.0 * M * d * d),
}
var, arms = sys.argv[1], sys.argv[2:]
for name, (fn, fl) in cases.items():
    res = {a: [] for a in arms}
    for rnd in range(int(os.environ.get('FLEXAM_AB_ROUNDS', '7'))):
        for a in (arms if rnd % 2 == 0 else arms[::-1]):
            os.environ[var] = a
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6): fn()
            torch.cuda.synchronize()
            res[a].append((time.perf_counter() - t0) / 6)
    base = statistics.median(res[arms[0]])
    print(f"{name:20s} " + "  ".join(f"{var}={a}: {statistics.median(v) * 1e6:7.1f} us ({fl / statistics.median(v) / 1e12:5.0f} TF/s, {100 * (base / statistics.median(v) - 1):+.1f}%)" for a, v in res.items()), flush=True)
