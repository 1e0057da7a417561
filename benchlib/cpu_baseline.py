"""`cpu_baseline` legs of the JSON line: the fp32 oracle (oracle/: test infrastructure, pinned to the reference by golden vectors) timed on the
host cores AFTER the timed region.  The only bench module that imports `oracle`."""
import time

import torch


BASELINE_THREADS = 32      # r5, tools/oracle_threads_probe.py on the GPU box (256 cores): the L = 11648 block takes 10.1 s on 32 threads, 12.0 s on 64
                           # and 20.1 s on torch's default 128 -- the baseline is quoted at the thread count that is FASTEST for it


def _baseline_threads():
    n = torch.get_num_threads()
    if n > BASELINE_THREADS:
        torch.set_num_threads(BASELINE_THREADS)
    return n


def cpu_baseline(L, cfg):
    """fp32 oracle (oracle/dit.py: the restatement pinned to the reference by golden vectors) on the
    host cores: ONE WanAttentionBlock forward on ONE sample at the full token count = 1/60 of a step."""
    from oracle import cases as C
    from oracle import dit as O
    d, f, nh, T = cfg["dim"], cfg["ffn_dim"], cfg["num_heads"], cfg["text_len"]
    one = dict(cfg, num_layers=1)
    shapes = {k: v for k, v in O.dit_param_shapes(one).items() if k.startswith("blocks.0.")}
    sd = O.seeded_state_dict(shapes, 3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, L, d, generator=g)
    e0 = torch.randn(1, 6, d, generator=g) * 0.1
    dens0 = torch.randn(1, 2, d, generator=g) * 0.1
    ctx = torch.randn(1, T, d, generator=g)
    grid = (26, 16, 28) if L == 11648 else (1, 1, L)
    ang = O.rope_angles(1024, d // nh)
    restore = _baseline_threads()
    t0 = time.perf_counter()
    with torch.no_grad():
        O.block_forward(sd, "blocks.0", x, e0, dens0, grid, ang, ctx, nh)
    sec = time.perf_counter() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(restore)
    steps_per_sec = 1.0 / (sec * 60.0)
    return dict(value=steps_per_sec, unit="denoise-steps/sec", cores=used, kind="port",
                sample=f"1 of the 60 block-forwards of one step (oracle/dit.py block_forward, fp32, L={L}, d={d}) "
                       f"took {sec:.1f} s; value = 1/(60 x that), extrapolated", block_seconds=sec)


def cpu_baseline_legs(cfg, layers=3):
    """The other two legs of SURVEY 8(d)'s CPU baseline, each a bounded sample on the host cores (fp32 oracle):
    (i)  BASELINE config 1 end to end -- 9x256x256 (latent [1,48,3,16,16], L = 256), 4 Euler steps, CFG pair -- with `layers` of
         the 30 layers of the 5B-width model (the weights of 30 would be 20 GB of fp32), block time extrapolated to 30;
    (ii) the Wan2.2 VAE decoder at its true widths on a 1/16-area latent [1,48,2,8,14] (first chunk + one cached 4-frame chunk),
         extrapolated x16 in area and to the 25 latent frames of a 97-frame clip."""
    from oracle import cases as C
    from oracle import dit as O
    from oracle import sampler as S
    from oracle import vae as OV
    restore = _baseline_threads()
    c1 = dict(cfg, num_layers=layers)
    sd = C.dit_weights(c1, 5)
    sc = C.sampler_case(c1)
    ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    t0 = time.perf_counter()
    with torch.no_grad():
        S.denoise_loop(lambda **k: O.dit_forward(sd, c1, **k), S.FlowMatchEulerSchedule(1000, 5.0), 4, sc["latents"], sc["context_uncond"],
                       sc["context_cond"], sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"],
                       sc["ref_latents"], mask, pinned, 0.1, 6.0)
    sec1 = time.perf_counter() - t0
    del sd
    v = dict(z_dim=48, dec_dim=256, dim_mult=(1, 2, 4, 4), temporal_up=(True, True, False))
    vsd = C.vae_weights(v, seed=61, prefix="model.")
    z = C.vae_case(seed=62, frames=2, h=8, w=14)
    t0 = time.perf_counter()
    with torch.no_grad():
        OV.vae_decode(vsd, z, v["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    sec2 = time.perf_counter() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(restore)
    return {
        "cores": used,
        "config1_4_steps": dict(seconds=sec1, layers_run=layers, sample=f"9x256x256, 4 Euler steps, CFG pair, {layers} of 30 layers at d=3072 (oracle loop + dit_forward)",
                                extrapolated_seconds_30_layers=sec1 * 30.0 / layers),
        "vae_decode_chunk": dict(seconds=sec2, sample="true-width decoder, latent [1,48,2,8,14] (1/16 area): first chunk + one 4-frame chunk",
                                 extrapolated_seconds_97x512x896=sec2 / 5.0 * 97.0 * 16.0),
    }



def raster_inputs(frames, height, width, step=4, seed=0):
    """Synthetic tracks of the clip's size: a `step`-pixel grid of points that drifts apart over the frames (28672 points per frame at
    512 x 896), random depths, 5 % invisible.  [T, N, 3] float32 (u, v, depth), [T, N] bool."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(step // 2, height, step), np.arange(step // 2, width, step), indexing="ij")
    base = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.float32)
    n = base.shape[0]
    pts = np.zeros((frames, n, 3), np.float32)
    drift = rng.normal(0, 0.6, (n, 2)).astype(np.float32)
    for t in range(frames):
        pts[t, :, :2] = base + drift * t
    pts[:, :, 2] = rng.uniform(0.5, 9, (frames, n))
    return pts, rng.random((frames, n)) > 0.05


def raster_port_leg(pts, vis, height, width, sample_frames=2):
    """CPU leg of the conditioning rasteriser: the numpy oracle (oracle/raster.py: the reference's six videos, vectorised -- NOT its
    per-point PIL loop, which takes ~125 s per clip in the build container) on `sample_frames` frames, extrapolated to the clip."""
    from oracle import raster as O
    t_n = pts.shape[0]
    sub = [0] + list(range(t_n // 2, t_n // 2 + sample_frames - 1))
    t0 = time.perf_counter()
    O.visualize_tracking(pts[sub], vis[sub], 4, height, width, 4)
    sec = time.perf_counter() - t0
    return dict(seconds=sec, cores=1, kind="port", sample=f"{len(sub)} of {t_n} frames, all six videos (oracle/raster.py, numpy)",
                extrapolated_seconds_clip=sec / len(sub) * t_n)
