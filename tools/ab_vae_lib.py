"""Dev tool: VAE encode / decode of the 97 x 512 x 896 clip under two builds of the library in ONE process, alternated
(A = FLEXAM_AB_A, default tools/probes/libflexam_var_base.so; B = the in-tree build).  usage: ab_vae_lib.py [rounds]"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
here = os.path.dirname(os.path.abspath(__file__))
libs = {"A": os.environ.get("FLEXAM_AB_A", os.path.join(here, "probes", "libflexam_var_base.so")), "B": H.LIB_PATH}
torch.manual_seed(0)
with torch.device("cuda:0"):
    vae = AutoencoderKLWan3_8(spatial_compression_ratio=16)
    for n, p in vae.named_parameters():
        if p.dim() > 1 and p.shape[1:].numel() > 1 and not n.endswith("gamma"):
            torch.nn.init.normal_(p, std=(1.0 / p.shape[1:].numel()) ** 0.5)
        elif n.endswith("gamma"):
            torch.nn.init.ones_(p)
        else:
            torch.nn.init.zeros_(p)
vae = vae.to(torch.bfloat16)
z = torch.randn(1, 48, 25, 32, 56, device="cuda:0")
x = torch.rand(1, 3, 97, 512, 896, device="cuda:0") * 2 - 1


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {(t, w): [] for t in "AB" for w in ("enc", "dec")}
outs = {}
for tag in "AB":                                   # warm both
    H.load_library(libs[tag])
    vae.encode(x); vae.decode(z)
for r in range(rounds):
    for tag in "AB":
        H.load_library(libs[tag])
        dt, e = timed(lambda: vae.encode(x).latent_dist.mode())
        res[(tag, "enc")].append(dt)
        dt, d = timed(lambda: vae.decode(z).sample)
        res[(tag, "dec")].append(dt)
        outs[tag] = (e.float().clone(), d.float().clone())
for w in ("enc", "dec"):
    a, b = statistics.median(res[("A", w)]), statistics.median(res[("B", w)])
    print(f"{w}: A {a * 1e3:7.1f} ms   B {b * 1e3:7.1f} ms   B/A {b / a:.4f}   (A {min(res[('A', w)]) * 1e3:.1f}-{max(res[('A', w)]) * 1e3:.1f}, B {min(res[('B', w)]) * 1e3:.1f}-{max(res[('B', w)]) * 1e3:.1f})")
print("encode max |A - B|", float((outs["A"][0] - outs["B"][0]).abs().max()), " decode max |A - B|", float((outs["A"][1] - outs["B"][1]).abs().max()))
