"""Dev tool: self-attention time at the per-rank shapes of N = 1, 2, 4, 8 GPUs (CFG rows first, then token chunks), with and
without split-KV -- the part of strong scaling that tile quantisation on 256 CUs decides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
kv = (torch.randn(2, L, 2 * d, generator=g) * 0.5).to(BF).to(dev)
for n, (b, lq) in {1: (2, L), 2: (1, L), 4: (1, L // 2), 8: (1, L // 4)}.items():
    q = (torch.randn(b, lq, d, generator=g) * 0.5).to(BF).to(dev).unflatten(2, (24, 128))
    k, v = (kv[:b, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(2))
    o = torch.empty(b, lq, 24, 128, dtype=BF, device=dev)
    fl = 4.0 * b * 24 * lq * L * 128
    line = f"N={n}: B={b} Lq={lq:5d}"
    plan = H.attn_split_plan(b * 24, lq, L)
    for label, kw in (("one pass", dict(kv_splits=1)), (f"plan S={plan[0]} from unit {plan[1]}", dict()),
                      (f"all units S={plan[0]}", dict(kv_splits=plan[0]))):
        for _ in range(2):
            H.attn_fwd(q, k, v, out=o, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            H.attn_fwd(q, k, v, out=o, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        line += f" | {label}: {dt * 1e3:6.3f} ms {fl / dt / 1e12:6.1f} TF/s"
    print(line, flush=True)
