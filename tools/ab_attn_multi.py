"""Dev tool: times the DiT's self-attention call under several builds of the library (tools/probes/libflexam_<tag>.so given as tags on
the command line, plus the in-tree build as "tree"), round-robin in one process, medians.  Used for ablation builds that drop one
ingredient of the loop (their results are wrong by construction; only the time is read)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
here = os.path.dirname(os.path.abspath(__file__))
tags = sys.argv[1:]
libs = {"tree": H.LIB_PATH, **{t: os.path.join(here, "probes", f"libflexam_{t}.so") for t in tags}}
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
out = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
fl = 4.0 * 2 * 24 * L * L * 128
res = {t: [] for t in libs}
for r in range(7):
    order = list(libs) if r % 2 == 0 else list(libs)[::-1]
    for tag in order:
        H.load_library(libs[tag])
        H.attn_fwd(q, k, v, out=out, prescaled=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            H.attn_fwd(q, k, v, out=out, prescaled=True)
        torch.cuda.synchronize()
        res[tag].append((time.perf_counter() - t0) / 5)
H.load_library(libs["tree"])
base = statistics.median(res["tree"])
for tag, ts in res.items():
    m = statistics.median(ts)
    print(f"{tag:8s} {m * 1e3:7.3f} ms  {fl / m / 1e12:7.1f} TF/s-equivalent  time ratio {m / base:.3f}")
