"""Dev tool: same-process A/B of self-attention (B=2, H=24, L=11648, D=128, tail split as in the DiT): A = softmax scale applied
per score (generic form), B = q pre-scaled by its producer (FLEXAM_ATTN_PRESCALED form); alternating, several rounds."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
c = 128 ** -0.5 * 1.4426950408889634
qs = (q.float() * c).to(BF)
outs = {t: torch.empty(2, L, 24, 128, dtype=BF, device=dev) for t in "AB"}
fl = 4.0 * 2 * 24 * L * L * 128
here = os.path.dirname(os.path.abspath(__file__))
base = os.path.join(here, "probes", "libflexam_base.so")        # optional third column: another build's generic form
def run_base():
    H.load_library(base); H.attn_fwd(q, k, v, out=outs["A"]); H.load_library(H.LIB_PATH)
def run_base_n(n):
    H.load_library(base)
    for _ in range(n):
        H.attn_fwd(q, k, v, out=outs["A"])
    H.load_library(H.LIB_PATH)
fns = {"A": lambda: H.attn_fwd(q, k, v, out=outs["A"]), "B": lambda: H.attn_fwd(qs, k, v, out=outs["B"], prescaled=True)}
res = {"A": [], "B": [], "C": []}
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for tag in (("A", "B") if r % 2 == 0 else ("B", "A")):
        fn = fns[tag]
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[tag].append(fl / ((time.perf_counter() - t0) / 5) / 1e12)
    if os.path.exists(base):
        run_base(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_base_n(5)
        torch.cuda.synchronize()
        res["C"].append(fl / ((time.perf_counter() - t0) / 5) / 1e12)
ma, mb = statistics.median(res["A"]), statistics.median(res["B"])
diff = (outs["A"].float() - outs["B"].float()).abs()
if res["C"]:
    print(f"base build, generic form: {statistics.median(res['C']):7.1f} TF/s ({min(res['C']):.0f}-{max(res['C']):.0f})")
print(f"self-attn: generic {ma:7.1f}  pre-scaled {mb:7.1f} TF/s  ratio {mb / ma:.3f}  (A {min(res['A']):.0f}-{max(res['A']):.0f}, B {min(res['B']):.0f}-{max(res['B']):.0f});  |A-B| mean {diff.mean().item():.3e}, |A| mean {outs['A'].float().abs().mean().item():.3e}")
