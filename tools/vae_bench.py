"""Dev tool: full-size Wan2.2 VAE decode [1,48,25,32,56] -> [1,3,97,512,896] with random weights: time + sanity."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
torch.manual_seed(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 25
what = sys.argv[2] if len(sys.argv) > 2 else "all"          # all | decode | encode
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
for it in range(2 if what in ("all", "decode") else 0):
    t0 = time.perf_counter()
    out = vae.decode(z).sample
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode {tuple(z.shape)} -> {tuple(out.shape)}: {dt:.3f} s, finite={bool(torch.isfinite(out.float()).all())}, "
          f"absmax={float(out.float().abs().max()):.3f}, mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
tflop = 33.7 * (frames - 1) + 33.7 / 4
if what in ("all", "decode"):
    print(f"~{tflop:.0f} TFLOP -> {tflop / dt:.0f} TFLOP/s")

# ---- encode: [1,3,1+4(frames-1),512,896] -> [1,96,frames,32,56]
pix = 1 + 4 * (frames - 1)
x = torch.rand(1, 3, pix, 512, 896, device="cuda:0") * 2 - 1
torch.cuda.reset_peak_memory_stats()
for it in range(2 if what in ("all", "encode") else 0):
    t0 = time.perf_counter()
    post = vae.encode(x).latent_dist
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = post.parameters.float()
    print(f"encode {tuple(x.shape)} -> {tuple(p.shape)}: {dt:.3f} s, finite={bool(torch.isfinite(p).all())}, "
          f"absmax={float(p.abs().max()):.3f}, mem={torch.cuda.max_memory_allocated()/2**30:.1f} GiB")


def enc_flops(frames_pix, h=256, w=448, dim=160):
    """2*M*N*K of every conv GEMM of Encoder3d on a clip (interior positions only)."""
    dims = [dim, dim, 2 * dim, 4 * dim, 4 * dim]
    fl, t, hh, ww = 2.0 * frames_pix * h * w * 27 * 12 * dim, float(frames_pix), h, w
    for i in range(4):
        ci, co = dims[i], dims[i + 1]
        px = t * hh * ww
        fl += 2 * px * 27 * (ci * co + 3 * co * co) + (2 * px * ci * co if ci != co else 0)
        if i < 3:
            hh, ww = hh // 2, ww // 2
            fl += 2 * t * hh * ww * 9 * co * co
            if i > 0:
                t = 1 + (t - 1) / 2
                fl += 2 * t * hh * ww * 3 * co * co
    px = t * hh * ww
    fl += 2 * px * 27 * 4 * dims[-1] ** 2 + 2 * px * 27 * dims[-1] * 96 + px * (4 * 2 * dims[-1] ** 2 + 4 * hh * ww * dims[-1])
    return fl


tf = enc_flops(pix) / 1e12
if what in ("all", "encode"):
    print(f"encode ~{tf:.0f} TFLOP -> {tf / dt:.0f} TFLOP/s")
if what != "all":
    sys.exit(0)

# ---- row-band (parallel) decode: time of ONE rank's band for world = 2, 4, 8 (all ranks do the same amount of work)
eng = vae.engine()
for world in (2, 4, 8):
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        band = eng.decode(z[0], stripe=(world // 2, world))
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
    print(f"band decode world={world}: {tuple(band.shape)} {dtb:.3f} s per rank")
