"""Dev tool: the q-norm fusion piece by piece at the DiT shapes -- (QKV GEMM, rmsnorm_rope on q and k, attention) against (QKV GEMM with the
row statistic, rmsnorm_rope on k, attention that norms q in its prologue), and the same for the cross-attention chain.  Round-robin medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
from flexam_amd.rope import rope_tables
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
B, L, d, nh, hd = 2, 11648, 3072, 24, 128
M = B * L
h = (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev)
wqkv = (torch.randn(3 * d, d, generator=g) * 0.02).to(BF).to(dev)
wq = wqkv[:d].contiguous()
b3 = torch.zeros(3 * d, device=dev)
nq = (torch.ones(d) * hd ** -0.5 * 1.4426950408889634).to(dev); nk = torch.ones(d, device=dev)
cos, sin = (t.to(dev) for t in rope_tables((1, 1, L), L, hd))
qkv = torch.empty(M, 3 * d, dtype=BF, device=dev)
sq = torch.empty(M, 48, device=dev)
o = torch.empty(B, L, nh, hd, dtype=BF, device=dev)
q4, k4, v4 = (qkv[:, i * d:(i + 1) * d].view(B, L, nh, hd) for i in range(3))
ckv = (torch.randn(B, 128, 2 * d, generator=g) * 0.5).to(BF).to(dev)
kc, vc = ckv[:, :127, :d].unflatten(2, (nh, hd)), ckv[:, :127, d:].unflatten(2, (nh, hd))
qc = qkv[:, :d]
rr = dict(rope_cos=cos, rope_sin=sin, tokens_per_batch=L, token_offset=0, head_dim=hd)
parts = {
    "qkv gemm": lambda: H.gemm(h, wqkv, b3, out=qkv), "qkv gemm + statistic": lambda: H.gemm_rowsq(h, wqkv, b3, qkv, sq, cols=d),
    "rmsnorm_rope q,k": lambda: H.rmsnorm_rope(qkv[:, :d], nq, qkv[:, d:2 * d], nk, **rr), "rmsnorm_rope k": lambda: H.rmsnorm_rope(qkv[:, d:2 * d], nk, **rr),
    "self-attention": lambda: H.attn_fwd(q4, k4, v4, out=o, prescaled=True),
    "self-attention, q normed in the prologue": lambda: H.attn_fwd_qnorm(q4, k4, v4, sq, nq, out=o, rope_cos=cos, rope_sin=sin, tokens_per_batch=L),
    "cross-q gemm": lambda: H.gemm(h, wq, b3[:d], out=qc), "cross-q gemm + statistic": lambda: H.gemm_rowsq(h, wq, b3[:d], qc, sq),
    "rmsnorm cross q": lambda: H.rmsnorm_rope(qc, nq),
    "cross-attention": lambda: H.attn_fwd_lastkey(q4, kc, vc, 386.0, out=o, prescaled=True),
    "cross-attention, q normed in the prologue": lambda: H.attn_fwd_qnorm(q4, kc, vc, sq, nq, out=o, last_key_multiplicity=386.0),
}
H.gemm(h, wqkv, b3, out=qkv); H.gemm_rowsq(h, wqkv, b3, qkv, sq, cols=d); H.rmsnorm_rope(qkv[:, :d], nq, qkv[:, d:2 * d], nk, **rr)      # sane q, k values for the attention timings
res = {k: [] for k in parts}
names = list(parts)
for r in range(7):
    for k in (names if r % 2 == 0 else names[::-1]):
        if k.startswith("rmsnorm") or "gemm" in k:
            pass
        parts[k](); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            parts[k]()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 5)
for k, v in res.items():
    print(f"{k:45s} {statistics.median(v) * 1e6:8.1f} us", flush=True)
