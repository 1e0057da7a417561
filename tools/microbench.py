"""Dev tool: per-kernel timing at the BASELINE shapes (97x512x896: L = 11648, B = 2, d = 3072)."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    L, B, d, f, T = 11648, 2, 3072, 14336, 512
    M = B * L
    res = {}
    g = torch.Generator(device="cpu").manual_seed(0)
    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)
    for name, (n, k, epi) in {"qkv": (3 * d, d, 0), "oproj": (d, d, 0), "ffn1": (f, d, 1), "ffn2": (d, f, 0)}.items():
        a, w, b = rnd(M, k), rnd(n, k), torch.randn(n, device=dev)
        out = torch.empty(M, n, dtype=BF, device=dev)
        t = timeit(lambda: H.gemm(a, w, b, out=out, epilogue=epi))
        res["gemm_" + name] = dict(ms=t * 1e3, tflops=2.0 * M * n * k / t / 1e12)
    a, w, b = rnd(M, d), rnd(d, d), torch.randn(d, device=dev)
    x = torch.randn(M, d, device=dev)
    gate = torch.randn(4, d, device=dev)
    rows = torch.randint(0, 4, (M,), dtype=torch.int32, device=dev)
    t = timeit(lambda: H.gemm_gate_residual(a, w, b, x, gate, rows))
    res["gemm_oproj_gate_residual"] = dict(ms=t * 1e3, tflops=2.0 * M * d * d / t / 1e12)
    qkv = rnd(B, L, 3 * d)
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
    o = torch.empty(B, L, 24, 128, dtype=BF, device=dev)
    t = timeit(lambda: H.attn_fwd(q, k, v, out=o), iters=5)
    res["attn_self"] = dict(ms=t * 1e3, tflops=4.0 * B * L * L * d / t / 1e12)
    kc, vc = rnd(B, T, 24, 128), rnd(B, T, 24, 128)
    t = timeit(lambda: H.attn_fwd(q, kc, vc, out=o))
    res["attn_cross"] = dict(ms=t * 1e3, tflops=4.0 * B * L * T * d / t / 1e12)
    xx = torch.randn(M, d, device=dev)
    tab = torch.randn(4, 2, d, device=dev)
    hb = torch.empty(M, d, dtype=BF, device=dev)
    t = timeit(lambda: H.ln_modulate(xx, out=hb, shift=tab[:, 0], scale=tab[:, 1], row_index=rows))
    res["ln_modulate"] = dict(ms=t * 1e3, gbs=M * d * 6 / t / 1e9)
    t = timeit(lambda: H.gate_residual(xx, hb, gate, rows))
    res["gate_residual"] = dict(ms=t * 1e3, gbs=M * d * 10 / t / 1e9)
    wq = torch.ones(d, device=dev)
    cos, sin = torch.ones(L, 64, device=dev), torch.zeros(L, 64, device=dev)
    q2, k2 = qkv.view(M, 3 * d)[:, :d], qkv.view(M, 3 * d)[:, d:2 * d]
    t = timeit(lambda: H.rmsnorm_rope(q2, wq, k2, wq, rope_cos=cos, rope_sin=sin, tokens_per_batch=L))
    res["rmsnorm_rope_qk"] = dict(ms=t * 1e3, gbs=M * d * 2 * 2 * 2 / t / 1e9)
    for k_, v_ in res.items():
        print(k_, json.dumps({a: round(b, 3) for a, b in v_.items()}))


if __name__ == "__main__":
    main()
