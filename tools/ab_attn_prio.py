"""Dev tool: same-process A/B of FLEXAM_ATTN_PRIO (static s_setprio 1 for waves 4-7) on the self- and cross-attention shapes."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d, T = 11648, 3072, 512
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
ckv = (torch.randn(2, T, 2 * d, generator=g) * 0.5).to(BF).to(dev)
kc, vc = ckv[:, :, :d].unflatten(2, (24, 128)), ckv[:, :, d:].unflatten(2, (24, 128))
o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
cases = {"self": (lambda: H.attn_fwd(q, k, v, out=o, prescaled=True), 4.0 * 2 * 24 * L * L * 128, 5),
         "cross": (lambda: H.attn_fwd(q, kc, vc, out=o, prescaled=True), 4.0 * 2 * 24 * L * T * 128, 20)}
for name, (fn, fl, n) in cases.items():
    res = {"0": [], "1": []}
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for tag in (("0", "1") if r % 2 == 0 else ("1", "0")):
            os.environ["FLEXAM_ATTN_PRIO"] = tag
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            res[tag].append(fl / ((time.perf_counter() - t0) / n) / 1e12)
    a, b = statistics.median(res["0"]), statistics.median(res["1"])
    print(f"{name}: prio off {a:7.1f}  prio on {b:7.1f} TF/s  ratio {b / a:.3f}  (off {min(res['0']):.0f}-{max(res['0']):.0f}, on {min(res['1']):.0f}-{max(res['1']):.0f})")
