"""Dev tool: same-process A/B of one attention launcher switch read per call (FLEXAM_ATTN_FUSED_TAIL, FLEXAM_ATTN_BODY ...) on the
self-attention shapes of 1 / 2 / 4 / 8 ranks (B, Lq) with all keys.  usage: ab_attn_env.py VAR v1 v2 ...; checks that the arms agree."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
var, arms = sys.argv[1], sys.argv[2:]
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
for name, B, Lq in (("N=1 B=2 Lq=11648", 2, L), ("block 0 B=1 Lq=11648", 1, L), ("N=4 B=1 Lq=5824", 1, L // 2), ("N=8 B=1 Lq=2912", 1, L // 4)):
    qq, kk, vv = q[:B, :Lq], k[:B], v[:B]
    o = {a: torch.empty(B, Lq, 24, 128, dtype=BF, device=dev) for a in arms}
    fl = 4.0 * B * 24 * Lq * L * 128
    res = {a: [] for a in arms}
    n = 5
    for r in range(7):
        for a in (arms if r % 2 == 0 else arms[::-1]):
            os.environ[var] = a
            H.attn_fwd(qq, kk, vv, out=o[a], prescaled=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                H.attn_fwd(qq, kk, vv, out=o[a], prescaled=True)
            torch.cuda.synchronize()
            res[a].append((time.perf_counter() - t0) / n)
    base = statistics.median(res[arms[0]])
    same = all(torch.equal(o[arms[0]], o[a]) for a in arms[1:])
    print(f"{name:22s} " + "  ".join(f"{var}={a}: {statistics.median(t) * 1e6:7.1f} us ({fl / statistics.median(t) / 1e12:5.0f} TF/s, {100 * (base / statistics.median(t) - 1):+.1f}%)" for a, t in res.items()) + f"   outputs identical: {same}", flush=True)
