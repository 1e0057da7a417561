"""Dev tool: per-call fixed cost of the attention launch(es): time at Lk and 2 Lk keys -> slope (main loop) and intercept (prologues,
epilogues, launch turnaround, split-KV + merge of the last partial round).  Lq = 12288 fills whole rounds of 256 CUs (no split launch).
r2: intercept 0.10-0.14 ms of 2.8-3.0 ms at the DiT shape, 0.065-0.08 ms at Lq 12288; a persistent variant (one workgroup per CU walking its
work items, K/V ring and Q prefetched across items) measured the same time for self-attention and +4 % for the text cross-attention, at twice
the code (two ring phases) -- not kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
for Lq in (11648, 12288):
    q = (torch.randn(2, Lq, 24, 128, generator=g) * 0.5).to(BF).to(dev)
    out = torch.empty_like(q)
    for mode in ("",):
        res = {}
        for Lk in (5824, 11648, 23296):
            k = (torch.randn(2, Lk, 24, 128, generator=g) * 0.5).to(BF).to(dev)
            v = (torch.randn(2, Lk, 24, 128, generator=g) * 0.5).to(BF).to(dev)
            ts = []
            for r in range(5):
                H.attn_fwd(q, k, v, out=out, prescaled=True); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    H.attn_fwd(q, k, v, out=out, prescaled=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 5)
            res[Lk] = sorted(ts)[2]
        print(f"Lq {Lq} {mode:9s}: " + "  ".join(f"Lk {Lk}: {t * 1e3:.3f} ms ({4.0 * 2 * 24 * Lq * Lk * 128 / t / 1e12:.0f} TF/s)" for Lk, t in res.items())
              + f"  | intercept {(2 * res[11648] - res[23296]) * 1e3:.3f} ms, asymptote {4.0 * 2 * 24 * Lq * 11648 * 128 / (res[23296] - res[11648]) / 1e12:.0f} TF/s")
