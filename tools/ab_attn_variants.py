"""Dev tool: same-process round-robin of several builds of the library (tools/build_attn_variants.py) on the DiT's self-attention
call (pre-scaled form, B = 2, L = 11648, 24 heads; the tail split as in the engine).  usage: ab_attn_variants.py NAME NAME ...
(names of tools/probes/libflexam_var_NAME.so; `tree` = the in-tree library).  Medians over rounds; max |difference| to the first arm."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
here = os.path.dirname(os.path.abspath(__file__))
names = sys.argv[1:]
libs = {n: (H.LIB_PATH if n == "tree" else os.path.join(here, "probes", f"libflexam_var_{n}.so")) for n in names}
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
outs = {n: torch.empty(2, L, 24, 128, dtype=BF, device=dev) for n in names}
fl = 4.0 * 2 * 24 * L * L * 128
res = {n: [] for n in names}
rounds, n_it = int(os.environ.get("AB_ROUNDS", 7)), 5
for r in range(rounds):
    for n in (names if r % 2 == 0 else names[::-1]):
        H.load_library(libs[n])
        H.attn_fwd(q, k, v, out=outs[n], prescaled=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_it):
            H.attn_fwd(q, k, v, out=outs[n], prescaled=True)
        torch.cuda.synchronize()
        res[n].append((time.perf_counter() - t0) / n_it)
base = statistics.median(res[names[0]])
for n in names:
    t = statistics.median(res[n])
    diff = (outs[n].float() - outs[names[0]].float()).abs().max().item()
    print(f"{n:16s} {t * 1e6:8.1f} us  {fl / t / 1e12:6.0f} TF/s  {100 * (base / t - 1):+5.1f} %   (min {min(res[n]) * 1e6:7.1f})  max|diff to {names[0]}| {diff:.3g}", flush=True)
