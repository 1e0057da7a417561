"""Dev tool: full-size Wan2.2 VAE decode [1,48,25,32,56] -> [1,3,97,512,896] with random weights: time + sanity."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
torch.manual_seed(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 25
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
z = torch.randn(1, 48, frames, 32, 56, device="cuda:0")
torch.cuda.synchronize()
for it in range(2):
    t0 = time.perf_counter()
    out = vae.decode(z).sample
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode {tuple(z.shape)} -> {tuple(out.shape)}: {dt:.3f} s, finite={bool(torch.isfinite(out.float()).all())}, "
          f"absmax={float(out.float().abs().max()):.3f}, mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
tflop = 33.7 * (frames - 1) + 33.7 / 4
print(f"~{tflop:.0f} TFLOP -> {tflop / dt:.0f} TFLOP/s")
