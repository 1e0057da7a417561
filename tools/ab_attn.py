"""Dev tool: same-process A/B of self-attention (B=2, H=24, L=11648, D=128, tail split as in the DiT) between
tools/probes/libflexam_base.so (A) and the in-tree library (B); also prints the max abs difference of the two outputs."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
here = os.path.dirname(os.path.abspath(__file__))
libs = {"A": os.path.join(here, "probes", "libflexam_base.so"), "B": H.LIB_PATH}
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
outs = {t: torch.empty(2, L, 24, 128, dtype=BF, device=dev) for t in libs}
fl = 4.0 * 2 * 24 * L * L * 128
res = {"A": [], "B": []}
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for tag in ("A", "B"):
        H.load_library(libs[tag])
        fn = lambda: H.attn_fwd(q, k, v, out=outs[tag])
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[tag].append(fl / ((time.perf_counter() - t0) / 5) / 1e12)
ma, mb = statistics.median(res["A"]), statistics.median(res["B"])
diff = (outs["A"].float() - outs["B"].float()).abs()
print(f"self-attn: A {ma:7.1f}  B {mb:7.1f} TF/s  B/A {mb / ma:.3f}  (A {min(res['A']):.0f}-{max(res['A']):.0f}, B {min(res['B']):.0f}-{max(res['B']):.0f});  |A-B| max {diff.max().item():.3e} mean {diff.mean().item():.3e}, |A| mean {outs['A'].float().abs().mean().item():.3e}")
