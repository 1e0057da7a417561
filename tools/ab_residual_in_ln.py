"""Dev tool: what would moving the gated residual add out of the o-projection's epilogue into the LayerNorm behind it buy?
S1 = [GEMM with the fp32 read-modify-write epilogue, LayerNorm]  (today);  S2 = [GEMM with a plain bf16 store, gate_residual kernel, LayerNorm]
(the unfused form of the alternative: a fused LayerNorm would save one write and one read of X against S2).  Round-robin medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, d = 23296, 3072
a = (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev)
w = (torch.randn(d, d, generator=g) * 0.05).to(BF).to(dev)
b = torch.randn(d, device=dev)
x = torch.randn(M, d, device=dev)
gate = torch.randn(4, d, device=dev)
rows = (torch.arange(M, device=dev) % 4).to(torch.int32)
y = torch.empty(M, d, dtype=BF, device=dev)
h = torch.empty(M, d, dtype=BF, device=dev)
junk = torch.empty(160 << 20, dtype=torch.float32, device=dev)      # 640 MB touched between sequences: nothing of the last one stays cached


def s1():
    H.gemm_gate_residual(a, w, b, x, gate=gate, gate_row=rows)
    H.ln_modulate(x, out=h)


def s2():
    H.gemm(a, w, b, out=y)
    H.gate_residual(x, y, gate=gate, row_index=rows)
    H.ln_modulate(x, out=h)


def parts():
    out = {}
    for name, fn in (("gemm rmw", lambda: H.gemm_gate_residual(a, w, b, x, gate=gate, gate_row=rows)), ("gemm plain", lambda: H.gemm(a, w, b, out=y)),
                     ("gate_residual", lambda: H.gate_residual(x, y, gate=gate, row_index=rows)), ("ln_modulate", lambda: H.ln_modulate(x, out=h))):
        ts = []
        for _ in range(7):
            junk.zero_(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[name] = statistics.median(ts) * 1e6
    return out


res = {"S1": [], "S2": []}
for r in range(9):
    for name, fn in ((("S1", s1), ("S2", s2)) if r % 2 == 0 else (("S2", s2), ("S1", s1))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 5)
print({k: round(statistics.median(v) * 1e6, 1) for k, v in res.items()}, "us per sequence (back to back, caches warm)")
print({k: round(v, 1) for k, v in parts().items()}, "us single cold calls (incl. ~10 us of launch + sync)")
