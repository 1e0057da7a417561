"""Dev tool: full-size Wan2.2 VAE encode ([1,3,97,512,896], random weights) timed under values of ONE environment switch that the
library reads once per process (e.g. FLEXAM_GEMM_N160_TALL): a fresh child process per value, medians of 3 runs, and a checksum of the
latents so that arms that must agree bit for bit can be seen to.  usage: vae_encode_ab.py VAR v1 v2 ..."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    var, vals = sys.argv[1], sys.argv[2:]
    for v in vals:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        print(f"{var}={v}: {out.stdout.strip() or out.stderr[-500:]}", flush=True)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time, torch
from flexam_amd import AutoencoderKLWan3_8
dev = torch.device("cuda", 0)
torch.manual_seed(1)
with torch.device(dev):
    vae = AutoencoderKLWan3_8(spatial_compression_ratio=16)
    for n, prm in vae.named_parameters():
        if n.endswith("gamma"):
            torch.nn.init.ones_(prm)
        elif prm.dim() > 1:
            torch.nn.init.normal_(prm, std=(1.0 / prm.shape[1:].numel()) ** 0.5)
        else:
            torch.nn.init.zeros_(prm)
vae = vae.to(torch.bfloat16)
x = torch.rand(1, 3, 97, 512, 896, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 2 - 1
vae.encode(x)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mu = vae.encode(x).latent_dist.mode()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"encode {sorted(ts)[1] * 1e3:.1f} ms (3 runs: {[round(t * 1e3, 1) for t in ts]}), latents sum {float(mu.double().sum()):.6f} absmax {float(mu.abs().max()):.4f}")
