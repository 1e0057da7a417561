"""Dev tool: same-process A/B of two builds of the library on the DiT's attention calls (pre-scaled form): A = tools/probes/
libflexam_base.so (a copy of an earlier build), B = the in-tree library; alternating, medians; also checks that both give the same bits."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
here = os.path.dirname(os.path.abspath(__file__))
libs = {"A": os.environ.get("FLEXAM_AB_A", os.path.join(here, "probes", "libflexam_ab_base.so")), "B": os.environ.get("FLEXAM_AB_B", H.LIB_PATH)}
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d, T = 11648, 3072, 512
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
ckv = (torch.randn(2, T, 2 * d, generator=g) * 0.5).to(BF).to(dev)
kc, vc = ckv[:, :, :d].unflatten(2, (24, 128)), ckv[:, :, d:].unflatten(2, (24, 128))
outs = {t: torch.empty(2, L, 24, 128, dtype=BF, device=dev) for t in "AB"}
cases = {"self": (lambda o: H.attn_fwd(q, k, v, out=o, prescaled=True), 4.0 * 2 * 24 * L * L * 128, 5),
         "cross": (lambda o: H.attn_fwd(q, kc, vc, out=o, prescaled=True), 4.0 * 2 * 24 * L * T * 128, 20)}
for name, (fn, fl, n) in cases.items():
    res = {"A": [], "B": []}
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for tag in (("A", "B") if r % 2 == 0 else ("B", "A")):
            H.load_library(libs[tag])
            fn(outs[tag]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn(outs[tag])
            torch.cuda.synchronize()
            res[tag].append(fl / ((time.perf_counter() - t0) / n) / 1e12)
    H.load_library(libs["B"])
    a, b = statistics.median(res["A"]), statistics.median(res["B"])
    print(f"{name}: base {a:7.1f}  new {b:7.1f} TF/s  ratio {b / a:.3f}  (base {min(res['A']):.0f}-{max(res['A']):.0f}, new {min(res['B']):.0f}-{max(res['B']):.0f})  same bits: {bool(torch.equal(outs['A'], outs['B']))}")
