"""Dev tool: same-process A/B of the full-size VAE encode ([1,3,97,512,896]) and decode ([1,48,25,32,56]) between the tree's library and
tools/probes/libflexam_var_NAME.so (usage: ab_vae_variants.py NAME); medians over 3 runs per arm, and the largest output difference."""
import os, sys, time, statistics
sys.path.insert(0, "/root/repo")
import torch
from flexam_amd import hip as H
from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
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
x = torch.rand(1, 3, 97, 512, 896, device="cuda:0") * 2 - 1
z = torch.randn(1, 48, 25, 32, 56, device="cuda:0")
libs = {"tree": H.LIB_PATH, "var": "/root/repo/tools/probes/libflexam_var_%s.so" % sys.argv[1]}
outs = {}
res = {(l, w): [] for l in libs for w in ("enc", "dec")}
for rnd in range(4):
    for l, path in libs.items():
        H.load_library(path)
        for w, fn in (("enc", lambda: vae.encode(x).latent_dist.parameters), ("dec", lambda: vae.decode(z).sample)):
            torch.cuda.synchronize(); t0 = time.perf_counter(); o = fn(); torch.cuda.synchronize()
            res[(l, w)].append(time.perf_counter() - t0)
            outs[(l, w)] = o.float()
for k, v in res.items():
    print(k, "median %.4f s" % statistics.median(v[1:]))
for w in ("enc", "dec"):
    a, b = outs[("tree", w)], outs[("var", w)]
    print(w, "max |diff|", float((a - b).abs().max()))
