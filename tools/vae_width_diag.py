"""Dev tool: true-width VAE decode chunk vs the oracle, with per-stage taps (where does the HIP decoder leave the oracle?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, vae as OV
from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
BF = torch.bfloat16
dd = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hh, ww = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 14)
v = dict(z_dim=48, dec_dim=dd, dim_mult=(1, 2, 4, 4), temporal_up=(True, True, False))
sd = C.vae_weights(v, seed=61, prefix="model.")
vae = AutoencoderKLWan3_8(latent_channels=48, dec_dim=dd, dim_mult=[1, 2, 4, 4], temperal_downsample=[False, True, True], spatial_compression_ratio=16)
vae.load_state_dict(sd, strict=False)
vae = vae.to("cuda:0").to(BF)
z = C.vae_case(seed=62, frames=2, h=hh, w=ww)
out = vae.decode(z.cuda()).sample.float().cpu()
sd_b = {k: u.to(BF).float() for k, u in sd.items()}
with torch.no_grad():
    want = OV.vae_decode(sd_b, z, v["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
print("SPLITK", os.environ.get("FLEXAM_GEMM_SPLITK"), "dec_dim", dd, "latent", hh, ww, "psnr", C.psnr(out, want, peak=2.0),
      "per-frame", [round(C.psnr(out[:, :, i], want[:, :, i], peak=2.0), 1) for i in range(out.shape[2])],
      "out absmean", float(out.abs().mean()), "want absmean", float(want.abs().mean()), "frac clamped", float((want.abs() >= 1).float().mean()))
