"""Dev probe (r6): would a rank of eight gain from running its block as TWO half-row chains on two streams (each launch half the rows, the
other chain's launches filling the CUs a launch's prologue / epilogue leaves idle)?  The GEMM / LayerNorm chain of one block (no attention:
it needs all keys) at M = 2912 on one stream against 2 x M = 1456 on two streams, same total work, medians of 9 rounds of 10 chains."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
d, f = 3072, 14336
r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)
w_qkv, w_o, w_cq, w_co, w_f1, w_f2 = r(3 * d, d), r(d, d), r(d, d), r(d, d), r(f, d), r(d, f)
b3, bf_, bd = torch.zeros(3 * d, device=dev), torch.zeros(f, device=dev), torch.zeros(d, device=dev)
gate = torch.randn(4, 6, d, device=dev)


def buffers(m):
    return dict(x=torch.randn(m, d, device=dev), h=torch.empty(m, d, device=dev, dtype=BF), qkv=torch.empty(m, 3 * d, device=dev, dtype=BF),
                ao=r(m, d), ffn=torch.empty(m, f, device=dev, dtype=BF), rows=(torch.arange(m, device=dev) % 2).to(torch.int32))


def chain(b):
    H.ln_modulate(b["x"], out=b["h"], shift=gate[:, 0], scale=gate[:, 1], row_index=b["rows"])
    H.gemm(b["h"], w_qkv, b3, out=b["qkv"])
    H.gemm_gate_residual(b["ao"], w_o, bd, b["x"], gate=gate[:, 2], gate_row=b["rows"])
    H.ln_modulate(b["x"], out=b["h"])
    H.gemm(b["h"], w_cq, bd, out=b["qkv"][:, :d])
    H.gemm_gate_residual(b["ao"], w_co, bd, b["x"])
    H.ln_modulate(b["x"], out=b["h"], shift=gate[:, 3], scale=gate[:, 4], row_index=b["rows"])
    H.gemm(b["h"], w_f1, bf_, out=b["ffn"], epilogue=H.EPI_GELU_TANH)
    H.gemm_gate_residual(b["ffn"], w_f2, bd, b["x"], gate=gate[:, 5], gate_row=b["rows"])


M = int(os.environ.get("PROBE_M", "2912"))
one = buffers(M)
halves = [buffers(M // 2), buffers(M // 2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
N_IT = 10


def run_one():
    for _ in range(N_IT):
        chain(one)


def run_two():
    for _ in range(N_IT):
        for s, b in zip(streams, halves):
            with torch.cuda.stream(s):
                chain(b)


res = {"one stream, M rows": [], "two streams, M/2 rows each": []}
for rnd in range(10):
    for name, fn in ((("one stream, M rows", run_one), ("two streams, M/2 rows each", run_two)) if rnd % 2 == 0 else (("two streams, M/2 rows each", run_two), ("one stream, M rows", run_one))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if rnd:
            res[name].append((time.perf_counter() - t0) / N_IT * 1e6)
base = statistics.median(res["one stream, M rows"])
for k, v in res.items():
    print(f"M = {M}: {k:28s} {statistics.median(v):8.1f} us per block chain ({100 * (statistics.median(v) / base - 1):+.1f} %)  min {min(v):.1f}")
