"""Dev tool: the hot kernels on random operands (what the workload looks like) and on zero operands (the same instruction stream with
almost no switching in the operand and result paths).  The difference is what the chip's power management takes: MI355X holds
1.6-1.8 GHz under these kernels on random data and clocks higher when the data does not toggle (MI355X_MICROARCH.md, DVFS
give-back).  Round-robin in one process, medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d, f = 11648, 3072, 14336
M = 2 * L


def operands(kind):
    r = (lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)) if kind == "random" else (lambda *s: torch.zeros(*s, dtype=BF, device=dev))
    return dict(q=r(2, L, 24, 128), k=r(2, L, 24, 128), v=r(2, L, 24, 128), x=r(M, d), w_qkv=r(3 * d, d), w_f1=r(f, d), h=r(M, f), w_f2=r(d, f))


sets = {k: operands(k) for k in ("random", "zero")}
out_a = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
out_q = torch.empty(M, 3 * d, dtype=BF, device=dev)
out_f = torch.empty(M, f, dtype=BF, device=dev)
xres = torch.zeros(M, d, device=dev)
bias3, biasf, biasd = torch.zeros(3 * d, device=dev), torch.zeros(f, device=dev), torch.zeros(d, device=dev)
cases = {
    "self-attention": (lambda o: H.attn_fwd(o["q"], o["k"], o["v"], out=out_a, prescaled=True), 4.0 * 2 * 24 * L * L * 128),
    "gemm qkv": (lambda o: H.gemm(o["x"], o["w_qkv"], bias3, out=out_q), 2.0 * M * 3 * d * d),
    "gemm ffn1 + gelu": (lambda o: H.gemm(o["x"], o["w_f1"], biasf, out=out_f, epilogue=H.EPI_GELU_TANH), 2.0 * M * f * d),
    "gemm ffn2 + residual": (lambda o: H.gemm_gate_residual(o["h"], o["w_f2"], biasd, xres), 2.0 * M * d * f),
}
for name, (fn, fl) in cases.items():
    res = {k: [] for k in sets}
    for r in range(6):
        for kind in (list(sets) if r % 2 == 0 else list(sets)[::-1]):
            fn(sets[kind]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): fn(sets[kind])
            torch.cuda.synchronize()
            res[kind].append((time.perf_counter() - t0) / 5)
    a, b = statistics.median(res["random"]), statistics.median(res["zero"])
    print(f"{name:22s} random operands {fl / a / 1e12:7.0f} TF/s ({fl / a / 2.5e15:.1%} of 2.5 PF)   zero operands {fl / b / 1e12:7.0f} TF/s ({fl / b / 2.5e15:.1%})   x{a / b:.2f}")
