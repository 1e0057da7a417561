"""Dev tool: self-attention calls with fewer work units than CUs (per-rank shapes of the multi-GPU layouts) over the number of key ranges per
unit: is the launcher's plan (hip.attn_split_plan) the best forced split?  Round-robin medians."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L = 11648
ka = (torch.randn(2, L, 24, 128, generator=g) * 0.5).to(BF).to(dev)
va = (torch.randn(2, L, 24, 128, generator=g) * 0.5).to(BF).to(dev)
qa = (torch.randn(2, L, 24, 128, generator=g) * 0.5).to(BF).to(dev)
for name, B, Hh, Lq in (("sp8 one sample: 3 heads x 46 blocks = 138 units", 1, 3, L), ("sp4 head group: 12 heads x 12 blocks = 144 units", 1, 12, 2912),
                        ("sp8 pair: 2 x 3 heads x 46 = 276 units", 2, 3, L), ("sp4 row: 24 heads x 12 = 288 units", 1, 24, 2912), ("sp4 ulysses row: 6 heads x 46 = 276", 1, 6, L)):
    q = qa[:B, :Lq, :Hh]
    k, v = ka[:B, :, :Hh], va[:B, :, :Hh]
    o = torch.empty(B, Lq, Hh, 128, dtype=BF, device=dev)
    units = B * Hh * ((Lq + 255) // 256)
    plan = H.attn_split_plan(B * Hh, Lq, L)
    arms = {"plan %s" % (plan,): None}
    for s in (3, 5, 7, 9, 11, 13):
        arms[f"S={s} all"] = (s, 0)
    if units > 256:
        for s in (5, 7, 9, 11):
            arms[f"S={s} from 256"] = (s, 256)
    res = {a: [] for a in arms}
    for r in range(5):
        for a, sp in (list(arms.items()) if r % 2 == 0 else list(arms.items())[::-1]):
            fn = (lambda: H.attn_fwd(q, k, v, out=o, prescaled=True)) if sp is None else (lambda: H.attn_fwd(q, k, v, out=o, prescaled=True, kv_splits=sp[0], split_from_unit=sp[1]))
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            res[a].append((time.perf_counter() - t0) / 5)
    ideal = 4.0 * B * Hh * Lq * L * 128 / 1.16e15
    print(f"{name} (ideal at 1160 TF/s: {ideal * 1e6:.0f} us): " + "  ".join(f"{a}: {statistics.median(t) * 1e6:.0f}" for a, t in res.items()), flush=True)
