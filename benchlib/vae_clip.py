"""VAE decode / encode timing and ONE clip end to end through the drop-in pipeline call."""
import time

import torch


def time_vae(device, frames, height, width):
    """Wan2.2 3D-VAE (random-init weights): decode of one clip (PIPE.py:951-955) and encode of one conditioning
    video stream + the reference image (PIPE.py:655-822) -> (decode s, encode-stream s, encode-image s, finite)."""
    from flexam_amd import AutoencoderKLWan3_8
    torch.manual_seed(1)
    with torch.device(device):
        vae = AutoencoderKLWan3_8(spatial_compression_ratio=16)
        for n, prm in vae.named_parameters():
            if n.endswith("gamma"):
                torch.nn.init.ones_(prm)
            elif prm.dim() > 1:
                torch.nn.init.normal_(prm, std=(1.0 / prm.shape[1:].numel()) ** 0.5)
            else:
                torch.nn.init.zeros_(prm)
    vae = vae.to(torch.bfloat16)
    z = torch.randn(1, 48, (frames - 1) // 4 + 1, height // 16, width // 16, device=device)
    vae.decode(z)                                   # warm-up (allocations, tap tables)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = vae.decode(z).sample
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    finite = bool(torch.isfinite(out.float()).all())
    enc = []
    for nf in (frames, 1):
        x = torch.rand(1, 3, nf, height, width, device=device) * 2 - 1
        vae.encode(x)                               # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mu = vae.encode(x).latent_dist.mode()
        torch.cuda.synchronize()
        enc.append(time.perf_counter() - t0)
        finite = finite and bool(torch.isfinite(mu.float()).all())
    return sec, enc[0], enc[1], finite, vae


def time_decode_band(vae, device, frames, height, width, world, which=None):
    """One rank's share of the tiled parallel decode (AutoencoderKLWan3_8.enable_parallel_decode: every rank decodes 1/N of the output
    pixels exactly -- a tile of the grid _DecoderEngine.band_grid picks --, one all-gather assembles the clip): the SLOWEST tile of a
    `world`-rank decode by the plan's own area model unless `which` names a rank, timed on this GPU; the all-gather is not included."""
    eng = vae.engine()
    z = torch.randn(48, (frames - 1) // 4 + 1, height // 16, width // 16, device=device)
    if which is None:
        cost = eng._stage_cost()

        def work(rank):
            crops, _, _ = eng.stripe_plan(z.shape[2], z.shape[3], rank, world)
            hh, ww, c = z.shape[2], z.shape[3], 0.0
            for si, st in enumerate(eng.stages):
                if si in crops:
                    hh, ww = crops[si][1] - crops[si][0], crops[si][3] - crops[si][2]
                c += cost[si] * hh * ww
                if st["up"]:
                    hh, ww = 2 * hh, 2 * ww
            return c
        which = max(range(world), key=work)
    r = which
    eng.decode(z, stripe=(r, world))                 # warm-up (this band's buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.decode(z, stripe=(r, world))
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def time_clip(model, vae, inp, frames, height, width, steps, device):
    """ONE clip end to end through the drop-in call the reference's demo makes (PIPE.py:505-965 via pipelines.py:1174-1190):
    pixel-space conditioning streams -> VAE encode of the 8 streams -> `steps` denoise steps -> VAE decode -> frames on the
    host.  Synthetic pixel videos (seeded), the bench's prompt embeddings; a 2-step call first takes the allocations."""
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    g = torch.Generator(device=device).manual_seed(7)
    vid = lambda: torch.rand(1, 3, frames, height, width, device=device, generator=g)
    mask = torch.full((1, 1, frames, height, width), 255.0, device=device)
    mask[:, :, 0] = 0                                      # motion_transfer: frame 0 kept, the rest regenerated
    streams = dict(video=vid(), control_video=vid(), depth_video=vid(), cos_control_videos={k: vid() for k in range(4)},
                   ref_image=torch.rand(1, 3, 1, height, width, device=device, generator=g), mask_video=mask)
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model, vae=vae)
    call = dict(prompt_embeds=inp["ctx_c"], negative_prompt_embeds=inp["ctx_u"], height=height, width=width, num_frames=frames,
                guidance_scale=6.0, density=0.1, latents=inp["latents"], output_type="pt", **streams)
    pipe(num_inference_steps=2, **call)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(num_inference_steps=steps, **call).videos
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    return sec, tuple(out.shape), bool(torch.isfinite(out.float()).all())



def time_raster(device, frames, height, width, cpu_leg=True):
    """The conditioning rasteriser at the clip's size: tracked points -> the six conditioning videos on the GPU
    (flexam_amd.conditioning_raster.visualize_tracking_DELTA: host colour tables + csrc/raster.hip), best of 3 after a warm-up run."""
    from flexam_amd import conditioning_raster as P
    from .cpu_baseline import raster_inputs, raster_port_leg
    pts, vis = raster_inputs(frames, height, width)
    best = None
    for it in range(4):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        tr, cos, dep = P.visualize_tracking_DELTA(pts, vis, False, 4, height, width, 4, device=device)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if it:
            best = dt if best is None else min(best, dt)
    painted = float((tr[0].sum(0) > 0).float().mean().item())
    out = {"sec": best, "points_per_frame": int(pts.shape[1]), "frames": frames, "videos": 6, "painted_fraction_tracking": painted,
           "what": "tracks [T, N, 3] + visibility on the host -> tracking, 4 cosine-level and depth videos [1, 3, T, H, W] fp32 on the GPU "
                   "(pipelines.py:1852-1902); the reference's PIL loop for the same input: ~125 s in the build container (8 cores, 1 used)"}
    del tr, cos, dep
    torch.cuda.empty_cache()
    if cpu_leg:
        out["cpu_baseline"] = raster_port_leg(pts, vis, height, width)
    return out
