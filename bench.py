#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec of the FlexAM hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N rank processes, see launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --mask blob                           (BASELINE configs[3]: foreground_edit masks, its own JSON line)

Workload (BASELINE config 2): Wan2.2-Fun-5B-FLEXAM DiT (dim 3072, 24 heads x 128, ffn 14336,
30 layers, 5.03 B parameters, random-init bf16), 97x512x896 -> latent [1,48,25,32,56], L = 11648 tokens
(448 reference-image tokens + 11200 video tokens), CFG pair (B = 2), flow-match Euler, 50-step schedule,
synthetic seeded conditioning.  One *step* = one iteration of the reference loop
(pipeline_wan2_2_fun_control_FlexAM.py:844-949): two DiT sample-forwards + CFG + Euler + masked blend.
N > 1: one process per GPU (strong scaling of ONE clip).  Default layout: the two CFG rows are split
first (independent until the guidance combine: no per-block traffic), then the token sequence into N/2 contiguous chunks whose
post-norm K|V are all-gathered per block over RCCL/xGMI while the rank attends to its local chunk ("allgather", the collective
BASELINE.json's north_star names).  FLEXAM_SP_MODE=ulysses selects the all-to-all-over-heads exchange instead (then N >= 4 runs
N-way token chunks with the CFG pair batched), FLEXAM_SP_OVERLAP=0 the all-gather without local-chunk-first attention,
FLEXAM_CFG_PARALLEL=0/1 overrides the CFG split; `config.parallelism` names what ran.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the self-attention flash kernel, timed
live with events on the launch stream) and `cpu_baseline` (the fp32 oracle on the host cores, one of the
60 block-forwards of a step, extrapolated -- a baseline, not a target).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 MFMA (block-scaled f8f6f4 instructions)
PEAK_HBM_GBS = 8000.0


def block_flops(L, d, f, T):
    """Algorithmic FLOPs of one WanAttentionBlock on one sample."""
    return 8 * L * d * d + 4 * L * L * d + 4 * L * d * d + 4 * T * d * d + 4 * L * T * d + 4 * L * d * f


def blob_mask_pixels(frames, height, width, mode):
    """Pixel-space mask video [1,1,F,H,W] of the foreground_edit mode (demo.py:87-124: 1 = regenerate; frame 0 is always 0):
    a disc that drifts and breathes over the frames.  "blob": demo.py's form (frame 0 kept -> PIPE.py:688-690 pins frame 0 and
    sets every later frame to 1: two distinct per-token timesteps, like motion_transfer, but fractional mask latents and a
    zeroed masked video);  "blob-open": the disc also covers frame 0 (not pinned: the trilinear latent mask has soft edges,
    a dozen distinct timesteps per sample);  "soft": uniform random mask values (stress: hundreds of distinct timesteps)."""
    if mode == "soft":
        return torch.rand(1, 1, frames, height, width, generator=torch.Generator().manual_seed(3))
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    m = torch.zeros(1, 1, frames, height, width)
    for f in range(0 if mode == "blob-open" else 1, frames):
        r = height * 0.22 * (1.0 + 0.2 * math.sin(0.2 * f))
        m[0, 0, f] = (((yy - height * 0.5 - 0.5 * f) ** 2 + (xx - width * 0.4 - 1.5 * f) ** 2) < r * r).float()
    return m


def synthetic_inputs(frames, height, width, text_dim, mask_mode="motion"):
    """Seeded synthetic conditioning of SURVEY 8(d) (CPU generators -> identical on every rank)."""
    f, h, w = (frames - 1) // 4 + 1, height // 16, width // 16
    g0 = torch.Generator().manual_seed(1245644)          # demo.py seed
    latents = torch.randn(1, 48, f, h, w, generator=g0)
    g1 = torch.Generator().manual_seed(1)
    control = torch.randn(1, 48, f, h, w, generator=g1)
    additional = torch.randn(1, 240, f, h, w, generator=g1)
    masked = torch.randn(1, 48, f, h, w, generator=g1)
    ref = torch.randn(1, 48, h, w, generator=g1)
    g2 = torch.Generator().manual_seed(2)
    ctx_u = [torch.randn(77, text_dim, generator=g2) * 0.1]
    ctx_c = [torch.randn(126, text_dim, generator=g2) * 0.1]
    mask_pixels = None
    if mask_mode == "motion":
        mask = torch.ones(1, 1, f, h, w)
        mask[:, :, 0] = 0                                 # motion_transfer: frame 0 known
        mask_latents = torch.zeros(1, 4, f, h, w)
        mask_latents[:, :, 0] = 1                         # resize_mask(1 - mask_condition) for that mask
    else:                                                 # foreground_edit (BASELINE configs[3]): PIPE.py:675-690 builds both from the pixel mask
        from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import prepare_masks
        mask_pixels = blob_mask_pixels(frames, height, width, mask_mode)
        ml, _, _ = prepare_masks(mask_pixels.clone(), (1, 48, f, h, w))
        masked = masked * ml[:, :1]                       # the masked video is zero where the mask says "regenerate"
        mask = mask_latents = None
    return dict(latents=latents, control=control, additional=additional, masked=masked, ref=ref, ctx_u=ctx_u, ctx_c=ctx_c,
                mask=mask, mask_latents=mask_latents, mask_pixels=mask_pixels)


def build_model(cfg, device):
    from flexam_amd import Wan2_2Transformer3DModel_FlexAM
    kw = dict(cfg)
    kw.pop("eps", None)
    torch.manual_seed(0)
    with torch.device(device):
        model = Wan2_2Transformer3DModel_FlexAM(**kw)
    model.randomize_zero_init(seed=0)
    return model.to(torch.bfloat16)


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()                 # recorded on torch's current stream = the stream every flexam_* call launches on
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def self_attention_in_step(pipe, step_index, B):
    """The dominant kernel timed where it runs: HIP events (torch's current stream = the stream every flexam_* call launches on)
    around every self-attention call of ONE more denoise step, outside the timed region.  Returns the mean over the calls that run
    the whole batch (29 of 30: block 0's call covers one sample when the CFG pair shares its self-attention half) in seconds --
    the figure rocprofv3 --kernel-trace reports as that kernel's average inside the step.  The isolated back-to-back timing
    (kernel_rooflines) runs at another clock: inside the step the clock is set by the GEMMs around the call."""
    from flexam_amd import hip
    real, marks = hip.attn_fwd, []

    def timed_attn(q, k, v, *a, **kw):
        if k.shape[1] <= 1024 or q.shape[1] != k.shape[1]:          # text cross-attention / partial calls: not the kernel in question
            return real(q, k, v, *a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = real(q, k, v, *a, **kw)
        e.record()
        marks.append((q.shape[0], s, e))
        return out
    real8 = hip.attn_fwd_fp8

    def timed_attn8(bufs, L, *a, **kw):                 # --sage: the MXFP8 kernel in the same place (its pack launch is not part of this figure)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = real8(bufs, L, *a, **kw)
        e.record()
        marks.append((kw["out"].shape[0] if kw.get("out") is not None else bufs[0].shape[0], s, e))
        return out
    hip.attn_fwd, hip.attn_fwd_fp8 = timed_attn, timed_attn8
    try:
        pipe.denoise_step(step_index)
        torch.cuda.synchronize()
    finally:
        hip.attn_fwd, hip.attn_fwd_fp8 = real, real8
    full = [s.elapsed_time(e) * 1e-3 for b, s, e in marks if b == B]
    every = [s.elapsed_time(e) * 1e-3 for b, s, e in marks]
    return {"sec": sum(full) / len(full), "calls": len(full), "calls_other_batch": len(marks) - len(full),
            "sec_all_calls": sum(every) / len(every)} if full else None


def kernel_rooflines(eng, B, L, lc):
    """Live per-launch timing of the hot kernels at this run's shapes, on the engine's own buffers."""
    from flexam_amd import hip
    d, f, nh, hd = eng.dim, eng.ffn, eng.nh, eng.hd
    ws = eng._workspace(B, lc)             # the step's own buffers (made here if the step ran another layout, e.g. the dual-stream mode)
    p = eng.blocks[0]
    qkv, ao, hbuf, ffn = ws["qkv"], ws["ao"], ws["h"], ws["ffn"]
    M = B * lc
    q4 = qkv.view(B, lc, 3 * d)[:, :, 0:d].unflatten(2, (nh, hd))
    out = {}
    if eng.sp_size > 1 and eng.sp_mode == "ulysses":
        # this rank's attention: all L tokens of nh / sp heads (the q|k|v it received in the last block's all-to-all)
        hg = nh // eng.sp_size
        full = ws["a2a_recv"].view(B, L, 3, hg, hd)
        t = time_kernel(lambda: hip.attn_fwd(full[:, :, 0], full[:, :, 1], full[:, :, 2], out=ws["a2a_out"], prescaled=True), iters=8)
        out["attn_self"] = dict(flops=4.0 * B * L * L * hg * hd, sec=t)
    else:
        if eng.sp_size == 1:
            k4 = qkv.view(B, lc, 3 * d)[:, :, d:2 * d].unflatten(2, (nh, hd))
            v4 = qkv.view(B, lc, 3 * d)[:, :, 2 * d:].unflatten(2, (nh, hd))
            t = time_kernel(lambda: hip.attn_fwd(q4, k4, v4, out=ao.view(B, lc, nh, hd), prescaled=True), iters=8)
        else:
            kv = ws["kv_cat"]                  # gathered K|V pieces of the last block [G, B, L, 2C/G] (same shape in every block)
            G = kv.shape[0]
            cb, hg = d // G, nh // G
            ao4 = ao.view(B, lc, nh, hd)

            def all_groups():
                for g in range(G):
                    hip.attn_fwd(q4[:, :, g * hg:(g + 1) * hg], kv[g, :, :, 0:cb].unflatten(2, (hg, hd)), kv[g, :, :, cb:].unflatten(2, (hg, hd)),
                                 out=ao4[:, :, g * hg:(g + 1) * hg], prescaled=True)
            t = time_kernel(all_groups, iters=8)
        out["attn_self"] = dict(flops=4.0 * B * lc * L * d, sec=t)
    t = time_kernel(lambda: hip.gemm(hbuf, p["wqkv"], p["bqkv"], out=qkv))
    out["gemm_qkv"] = dict(flops=2.0 * M * 3 * d * d, sec=t)
    t = time_kernel(lambda: hip.gemm(hbuf, p["w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH))
    out["gemm_ffn1_gelu"] = dict(flops=2.0 * M * f * d, sec=t)
    xs = torch.zeros(M, d, device=qkv.device, dtype=torch.float32)
    t = time_kernel(lambda: hip.gemm_gate_residual(ffn, p["w2"], p["b2"], xs))
    out["gemm_ffn2_residual"] = dict(flops=2.0 * M * d * f, sec=t)
    t = time_kernel(lambda: hip.gemm_gate_residual(ao, p["wo"], p["bo"], xs))
    out["gemm_oproj_residual"] = dict(flops=2.0 * M * d * d, sec=t)
    if getattr(eng, "fp8", False):
        w8 = eng._fp8_w[0]
        fused = d % 512 == 0 and d <= 4096            # the LN launch writes e4m3 + row scales + FFN1's output scales (DiTEngine._ln_fp8)
        if fused:
            a8, sa = hip.ln_modulate_fp8(ws["x"], ws["a8d"], ws["sa"], next_scale=ws["so"], next_wnorm=w8["w1_norm"], next_bias=w8["b1_max"])
        else:
            a8, sa = hip.quantize_rows_fp8(hbuf, ws["a8d"], ws["sa"])
        t = time_kernel(lambda: hip.gemm_fp8(a8, sa, w8["wqkv"], w8["s_wqkv"], p["bqkv"], out=qkv))
        out["gemm_fp8_qkv"] = dict(flops=2.0 * M * 3 * d * d, sec=t)
        if fused:                                      # FFN1 writes FFN2's e4m3 operand itself: no quantise pass in between
            t = time_kernel(lambda: hip.gemm_fp8_gelu_q(a8, sa, w8["w1"], w8["s_w1"], p["b1"], ws["so"], ws["a8"]))
            out["gemm_fp8_ffn1_gelu_e4m3_out"] = dict(flops=2.0 * M * f * d, sec=t)
            a8f, saf = ws["a8"], ws["so"]
            t = time_kernel(lambda: hip.ln_modulate_fp8(ws["x"], ws["a8d"], ws["sa"], next_scale=ws["so"], next_wnorm=w8["w1_norm"], next_bias=w8["b1_max"]))
            out["ln_modulate_fp8"] = dict(bytes=M * d * 5.0, sec=t)
        else:
            t = time_kernel(lambda: hip.gemm_fp8(a8, sa, w8["w1"], w8["s_w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH))
            out["gemm_fp8_ffn1_gelu"] = dict(flops=2.0 * M * f * d, sec=t)
            a8f, saf = hip.quantize_rows_fp8(ffn, ws["a8"], ws["sa"])
            t = time_kernel(lambda: hip.quantize_rows_fp8(ffn, ws["a8"], ws["sa"]))
            out["quantize_rows_fp8_ffn"] = dict(bytes=M * f * 3.0, sec=t)
        t = time_kernel(lambda: hip.gemm_fp8_gate_residual(a8f, saf, w8["w2"], w8["s_w2"], p["b2"], xs))
        out["gemm_fp8_ffn2_residual"] = dict(flops=2.0 * M * d * f, sec=t)
    for v in out.values():
        if "flops" in v:
            v["tflops"] = v["flops"] / v["sec"] / 1e12
    # bandwidth-bound kernels: ALGORITHMIC bytes / live time, against the 8 TB/s HBM3E peak
    T = torch.randn(4, 6, d, device=qkv.device)
    rows = (torch.arange(M, device=qkv.device) % 2).to(torch.int32)
    t = time_kernel(lambda: hip.ln_modulate(xs, out=hbuf, shift=T[:, 0], scale=T[:, 1], row_index=rows))
    out["ln_modulate"] = dict(bytes=M * d * 6.0, sec=t)                       # read fp32 x, write bf16
    cd = eng.cond
    if eng.sp_size == 1:
        t = time_kernel(lambda: hip.rmsnorm_rope(qkv[:, 0:d], p["nq"], qkv[:, d:2 * d], p["nk"], rope_cos=cd["cos"], rope_sin=cd["sin"],
                                                 tokens_per_batch=lc, head_dim=hd))
        out["rmsnorm_rope_qk"] = dict(bytes=M * d * 8.0, sec=t)               # q and k: read + write bf16
    return out


def sampler_step_roofline(pipe):
    """The fused CFG + Euler + blend launch on the clip's latents (26 MB algorithmic: two head-token rows, latents r/w, known, mask)."""
    from flexam_amd import hip
    st = pipe._state
    c, f, h, w = st["shape"]
    L = st["ref_len"] + f * (h // 2) * (w // 2)
    tok = torch.randn(2, L, 4 * c, device=st["latents"].device)
    lat = st["latents"].clone()
    t = time_kernel(lambda: hip.cfg_euler_blend(tok[0], tok[1], st["ref_len"], 6.0, -0.01, lat, st["known"], st["mask"]))
    n = c * f * h * w
    return dict(bytes=4.0 * (2 * n + 2 * n + n + n / c), sec=t)


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
    t0 = time.perf_counter()
    with torch.no_grad():
        O.block_forward(sd, "blocks.0", x, e0, dens0, grid, ang, ctx, nh)
    sec = time.perf_counter() - t0
    steps_per_sec = 1.0 / (sec * 60.0)
    return dict(value=steps_per_sec, unit="denoise-steps/sec", cores=torch.get_num_threads(), kind="port",
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
    return {
        "config1_4_steps": dict(seconds=sec1, layers_run=layers, sample=f"9x256x256, 4 Euler steps, CFG pair, {layers} of 30 layers at d=3072 (oracle loop + dit_forward)",
                                extrapolated_seconds_30_layers=sec1 * 30.0 / layers),
        "vae_decode_chunk": dict(seconds=sec2, sample="true-width decoder, latent [1,48,2,8,14] (1/16 area): first chunk + one 4-frame chunk",
                                 extrapolated_seconds_97x512x896=sec2 / 5.0 * 97.0 * 16.0),
    }


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


def multi_gpu_check(pipe, model, inp, cond, step_index, total_steps, world, rank):
    """Self-validation of an N-rank run, AFTER the timed region.  Every rank applies the identical sampler update to the gathered
    head output, so "all ranks hold the same latents" alone would also pass with a wrong K|V exchange.  Two checks:
      (1) ranks_agree: checksum of every rank's latents == rank 0's (they are bit-identical by construction);
      (2) rel_rms_vs_single_gpu: one more denoise step from the current latents on the N-rank layout, and the SAME step on this
          rank alone (weights are replicated: a second engine with no parallel layout, full CFG pair, no collective); the DiT's
          head outputs of the two runs -- per CFG row, BEFORE the guidance combine, which multiplies any difference by ~8 at
          guidance 6 -- must agree to the rounding of two summation orders: relative RMS <= 1.5e-2, the library's stated bf16
          tolerance (typical 2-6e-3; a wrong or missing remote chunk shows up as O(1)).
    Returns the `check` object of the JSON line (rank 0's view + the worst rank)."""
    import torch.distributed as dist
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM, hip
    tol = 1.5e-2

    def step_and_grab(p):
        got = {}
        orig = p._sampler_update

        def grab(i, tok_u, tok_c):
            got["rows"] = [t.double().clone() for t in (tok_u, tok_c) if t is not None]
            return orig(i, tok_u, tok_c)
        p._sampler_update = grab
        try:
            p.denoise_step(step_index)
        finally:
            p._sampler_update = orig
        return got["rows"]
    st = pipe._state
    lat0 = st["latents"].clone()
    rows_multi = step_and_grab(pipe)
    sums = [None] * world
    dist.all_gather_object(sums, hip.checksum(st["latents"]))
    agree = all(tuple(c) == tuple(sums[0]) for c in sums)
    layout = model._parallel
    model._parallel, model._engine = None, None                 # a fresh engine: one GPU, no collective
    solo = Wan2_2FunControlPipeline_FlexAM(transformer=model)
    solo.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    solo._state["latents"].copy_(lat0)
    rows_solo = step_and_grab(solo)
    rel = max(float(((m - s_).pow(2).mean().sqrt() / s_.pow(2).mean().sqrt().clamp_min(1e-30)).item()) for m, s_ in zip(rows_multi, rows_solo))
    model._parallel, model._engine = layout, None
    rels = [None] * world
    dist.all_gather_object(rels, rel)
    worst = max(rels)
    ok = bool(agree and worst <= tol and math.isfinite(worst))
    if (os.environ.get("FLEXAM_BENCH_TEST_HOOKS") == "1" and os.environ.get("FLEXAM_BENCH_FORCE_CHECK_FAIL") == "1"
            and os.environ.get("FLEXAM_SP_OVERLAP") != "0"):
        ok = False                       # test hook (tests/test_bench_launch.py): exercises the launcher's fallback attempt
    return {"ok": ok, "ranks": world, "ranks_agree": bool(agree), "rel_rms_vs_single_gpu": rel, "worst_rank_rel_rms": worst, "tolerance": tol,
            "what": "DiT head output of one denoise step per CFG row (before the guidance combine): N-rank layout vs the same step on "
                    "one GPU (no collective), every rank; checksums of the N ranks' latents"}


def visible_gpus_without_hip():
    """GPUs this process may use, counted from the KFD topology in sysfs (nodes with SIMDs), cut by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when they are plain index lists.  No HIP / HSA call: the launcher must not initialise the runtime before
    it starts the ranks (torch.cuda.device_count() only avoids HIP while its amdsmi path works).  None when sysfs says nothing."""
    import glob
    n = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                     # no KFD driver: no AMD GPU on this host
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            if all(x.strip().isdigit() for x in ids):
                n = min(n, len(ids))
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks ourselves.

    This parent never touches the GPU (no HIP call, no torch.cuda.is_available()): the ranks are FRESH child processes of
    `python -m torch.distributed.run`, never an exec of a process that has initialised the device.  Rank 0's JSON line and the
    launcher's exit code are forwarded.  The first attempt runs the default exchange; when its
    self-check (`check.ok`, see multi_gpu_check) fails, it crashes or it hangs, ONE more attempt runs the conservative form of the
    same exchange (FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0: one K|V all-gather per block, waited for before attention) and the line
    says so in `launch.fallback` -- a wrong or dead overlap path must not cost the scaling measurement."""
    import signal
    import socket
    import subprocess

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    n_dev = visible_gpus_without_hip()               # None: cannot tell without touching HIP -> the rank children report it
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev is not None and n_dev < args.gpus and env0.get("FLEXAM_BENCH_ONE_DEVICE") != "1":
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible (FLEXAM_BENCH_ONE_DEVICE=1 FLEXAM_BENCH_BACKEND=gloo "
                         f"runs the rank code path on one device for validation; such a line is marked invalid)")
    attempts = [("default", {})]
    if "FLEXAM_SP_OVERLAP" not in os.environ and "FLEXAM_SP_PIECES" not in os.environ and args.gpus > 2:
        attempts.append(("FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0", {"FLEXAM_SP_PIECES": "1", "FLEXAM_SP_OVERLAP": "0"}))
    limit = float(os.environ.get("FLEXAM_BENCH_ATTEMPT_TIMEOUT", "600"))
    last_rc, last_line, notes = 1, None, []
    for name, extra in attempts:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        proc = subprocess.Popen(cmd, env={**env0, **extra, "FLEXAM_BENCH_SPAWNED": "1"}, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=limit)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the process group this launcher created, nothing else
            out, _ = proc.communicate()
            rc = 124
        line = None
        for ln in (out or "").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            else:
                print(ln, file=sys.stderr)
        ok = rc == 0 and line is not None
        if ok:
            res = json.loads(line)
            chk = res.get("check")
            ok = chk is None or bool(chk.get("ok"))
            res["launch"] = {"spawned_by": "bench.py (parent made no GPU call)", "attempt": name, "earlier_attempts": notes or None,
                             "fallback": name != "default"}
            line = json.dumps(res)
        last_rc, last_line = rc, line
        if ok:
            break
        failed = json.loads(line) if line else {}
        notes.append({"attempt": name, "rc": rc, "check": failed.get("check"), "parallelism": failed.get("config", {}).get("parallelism"),
                      "layout_probe": failed.get("layout_probe")})
    if last_line is not None:
        print(last_line, flush=True)
    sys.exit(last_rc if last_rc != 0 else (0 if last_line is not None else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=97)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=896)
    ap.add_argument("--layers", type=int, default=30, help="debug only: fewer layers makes the number INVALID")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: QKV / FFN GEMMs on fp8 MFMA; a SEPARATE line (dtype fp8), never the headline")
    ap.add_argument("--sage", action="store_true", help="self-attention on MXFP8 operands (the reference's VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION switch); "
                    "a SEPARATE line (dtype says so), never the headline")
    ap.add_argument("--fp8-oproj", action="store_true", help="with --fp8: the two output projections of a block on the fp8 pipe as well (FLEXAM_FP8_OPROJ=1; "
                    "beyond configs[4]'s 'QKV/FFN', its own line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-legs", action="store_true", help="skip the config-1 and VAE-chunk CPU baseline legs (keep the one-block leg)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the end-to-end clip (pixels -> encode -> 50 steps -> decode -> frames)")
    ap.add_argument("--mask", choices=["motion", "blob", "blob-open", "soft"], default="motion",
                    help="motion: BASELINE configs[1] (motion_transfer, frame 0 known); blob: configs[3] foreground_edit as demo.py "
                         "builds it; blob-open / soft: foreground masks that are not pinned (many per-token timesteps)")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the cross-rank / single-GPU self-check after the timed region")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                      # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # FLEXAM_BENCH_ONE_DEVICE=1 + FLEXAM_BENCH_BACKEND=gloo: validation of the multi-rank code path on a box with ONE GPU
    # (all ranks share cuda:0, collectives go through the host); such a line is marked invalid below.
    one_device = os.environ.get("FLEXAM_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("FLEXAM_BENCH_BACKEND", "nccl")
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective that one rank never joins must end the attempt long before the launcher's limit (600 s), so that the
        # conservative second attempt still fits the driver's window
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("FLEXAM_BENCH_PG_TIMEOUT", "150")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from flexam_amd.configs import WAN22_FUN_5B_FLEXAM
    cfg = dict(WAN22_FUN_5B_FLEXAM, num_layers=args.layers)
    model = build_model(cfg, device)
    if args.fp8_oproj:
        if not args.fp8:
            raise SystemExit("--fp8-oproj needs --fp8")
        os.environ["FLEXAM_FP8_OPROJ"] = "1"                        # read by the engine at every forward
    if args.fp8:
        model.enable_fp8_gemm(True)
    if args.sage:
        os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"      # read by the engine at every forward, as the reference's attention() does
    if world > 1:
        cp = os.environ.get("FLEXAM_CFG_PARALLEL")
        model.enable_multi_gpus_inference(cfg_parallel=None if cp is None else cp == "1")
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model)
    def conditioning(mode):
        i = synthetic_inputs(args.frames, args.height, args.width, cfg["text_dim"], mode)
        return i, LatentConditioning(control_latents=i["control"], additional_control=i["additional"], masked_video_latents=i["masked"],
                                     ref_latents=i["ref"], mask_latents=i["mask_latents"], mask=i["mask"], mask_pixels=i["mask_pixels"])
    inp, cond = conditioning(args.mask)
    total_steps = 50
    # N >= 4 and nothing pinned by the caller: which exchange is faster depends on what the links of THIS node deliver (the K|V
    # all-gather moves N/2 times the bytes of the all-to-all over heads but hides part of them; the all-to-all uses every link of
    # the mesh when the CFG pair is batched).  Two steps of each candidate layout on the real fabric decide; the line says so.
    layout_probe = None
    pinned = [k for k in ("FLEXAM_SP_MODE", "FLEXAM_CFG_PARALLEL", "FLEXAM_SP_OVERLAP", "FLEXAM_SP_PIECES") if k in os.environ]
    if world >= 4 and not pinned and os.environ.get("FLEXAM_BENCH_LAYOUT_PROBE", "1") != "0":
        cands = [(f"cfg2 x sp{world // 2}, K|V all-gather", "allgather", True, "1", None)]
        if world // 2 >= 4:
            cands.append((f"cfg2 x sp{world // 2}, K|V all-gather in one piece", "allgather", True, "1", "1"))
        if cfg["num_heads"] % world == 0:
            cands.append((f"cfg1 x sp{world}, all-to-all over heads, a sample's blocks leave under the other's projection", "ulysses", False, "1", None))
            cands.append((f"cfg1 x sp{world}, all-to-all over heads, samples fully pipelined (attention per sample)", "ulysses", False, "2", None))
        if cfg["num_heads"] % (world // 2) == 0:
            cands.append((f"cfg2 x sp{world // 2}, all-to-all over heads", "ulysses", True, "1", None))
        # Per-candidate guard: a candidate that raises (every rank the same way: configuration errors) or whose FIRST step takes
        # longer than the budget is recorded in `skipped` and the probe goes on; all ranks decide on all-reduced values.  A rank that
        # dies or hangs alone cannot be skipped over in-process: the process-group timeout (FLEXAM_BENCH_PG_TIMEOUT) ends the attempt
        # and the launcher's second attempt runs the conservative exchange without a probe.
        budget = float(os.environ.get("FLEXAM_BENCH_PROBE_BUDGET", "20"))       # seconds for a candidate's first step (single GPU: 0.27 s)
        layout_probe = {"candidates": [], "steps": 2, "first_step_budget_sec": budget}

        def agreed_max(x):
            tt = torch.tensor([x], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())

        for name, mode, cfgp, ovl, pcs in cands:
            os.environ["FLEXAM_SP_MODE"], os.environ["FLEXAM_SP_OVERLAP"] = mode, ovl
            os.environ.pop("FLEXAM_SP_PIECES", None)
            if pcs is not None:
                os.environ["FLEXAM_SP_PIECES"] = pcs
            t_c = time.perf_counter()
            try:
                if os.environ.get("FLEXAM_BENCH_TEST_HOOKS") == "1" and os.environ.get("FLEXAM_BENCH_PROBE_RAISE") == str(cands.index((name, mode, cfgp, ovl, pcs))):
                    raise RuntimeError("test hook: this candidate raises on every rank")
                model.enable_multi_gpus_inference(cfg_parallel=cfgp)
                model._engine = None                      # the engine (buffers, per-clip state) is rebuilt for the layout on its next use
                pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
                pipe.denoise_step(0)
                torch.cuda.synchronize()
                first = agreed_max(time.perf_counter() - t_c)
                if first > budget:
                    layout_probe.setdefault("skipped", []).append({"layout": name, "error": f"first step took {first:.1f} s (> {budget:.0f} s budget)",
                                                                    "wall_sec": first})
                    continue
                dist.barrier(); torch.cuda.synchronize()
                tq = time.perf_counter()
                for i in range(2):
                    pipe.denoise_step(1 + i)
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                ms = agreed_max(time.perf_counter() - tq) / 2 * 1e3
                layout_probe["candidates"].append({"layout": name, "ms_per_step": ms, "mode": mode, "cfg_parallel": cfgp, "overlap": ovl, "pieces": pcs,
                                                   "wall_sec": round(time.perf_counter() - t_c, 2)})
            except Exception as e:                        # noqa: BLE001  (raised identically on every rank, or the PG timeout ends the attempt)
                layout_probe.setdefault("skipped", []).append({"layout": name, "error": f"{type(e).__name__}: {e}", "wall_sec": round(time.perf_counter() - t_c, 2)})
        test_hooks = os.environ.get("FLEXAM_BENCH_TEST_HOOKS") == "1"
        if layout_probe["candidates"]:
            best = min(layout_probe["candidates"], key=lambda c: c["ms_per_step"])       # identical on every rank (all-reduced times)
            if test_hooks and os.environ.get("FLEXAM_BENCH_LAYOUT_FORCE"):               # test hook: run candidate i whatever the probe measured
                best = layout_probe["candidates"][int(os.environ["FLEXAM_BENCH_LAYOUT_FORCE"])]
            layout_probe["chosen"] = best["layout"]
            os.environ["FLEXAM_SP_MODE"] = best["mode"]
            if best["overlap"] == "1":
                os.environ.pop("FLEXAM_SP_OVERLAP", None)      # the default; left unset so that a failed self-check can still fall back to 0
            else:
                os.environ["FLEXAM_SP_OVERLAP"] = best["overlap"]
            os.environ.pop("FLEXAM_SP_PIECES", None)
            if best["pieces"] is not None:
                os.environ["FLEXAM_SP_PIECES"] = best["pieces"]
            model.enable_multi_gpus_inference(cfg_parallel=best["cfg_parallel"])
        else:                                             # every candidate refused or over budget: keep the default layout
            layout_probe["chosen"] = "none measured: default layout"
            for k in ("FLEXAM_SP_MODE", "FLEXAM_SP_OVERLAP", "FLEXAM_SP_PIECES"):
                os.environ.pop(k, None)
            model.enable_multi_gpus_inference(cfg_parallel=None)
        model._engine = None
        from flexam_amd.dist import live_subgroups
        layout_probe["communicators"] = 1 + live_subgroups()     # the world group + the cached CFG-half groups (created once per member set)
    pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    torch.cuda.synchronize()
    tp0 = time.perf_counter()                       # second call: buffers exist, this is the per-clip cost of the step-invariant work
    pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    torch.cuda.synchronize()
    prepare_sec = time.perf_counter() - tp0
    eng = model.engine()
    L = eng.cond["L"]
    B = 2

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        pipe.denoise_step(i % total_steps)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.denoise_step((args.warmup + i) % total_steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(pipe._state["latents"]).all())
    rows_u = int(pipe._state["U"])
    # host side of a step: how long Python + ctypes need to ENQUEUE one step's ~420 launches (the GPU is idle-free while this
    # stays below the step's GPU time; at N ranks the GPU time shrinks N-fold, the enqueue time does not)
    sync()
    th0 = time.perf_counter()
    for i in range(2):
        pipe.denoise_step((args.warmup + args.steps + i) % total_steps)
    host_enqueue = (time.perf_counter() - th0) / 2
    sync()

    def timed(p):
        for i in range(args.warmup):
            p.denoise_step(i % total_steps)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            p.denoise_step((args.warmup + i) % total_steps)
        sync()
        return time.perf_counter() - t0
    motion_elapsed = None
    if args.mask != "motion" and world == 1:
        # configs[3] next to configs[1]: the SAME steps on the motion_transfer conditioning, in this process (the delta is what
        # the foreground masks cost: per-token timestep rows, AdaLN table size)
        inp_m, cond_m = conditioning("motion")
        pipe.prepare(inp_m["latents"], cond_m, inp_m["ctx_c"], inp_m["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
        motion_elapsed = timed(pipe)
        pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
        del inp_m, cond_m

    lc = L // eng.sp_size
    b_local = 1 if eng.cfg_size == 2 else B
    kern = None if args.no_kernel_timing else kernel_rooflines(eng, b_local, L, lc)
    attn_in_step = None
    if kern is not None and world == 1:
        attn_in_step = self_attention_in_step(pipe, (args.warmup + args.steps + 2) % total_steps, b_local)
    if kern is not None and pipe._state.get("known") is not None:
        kern["cfg_euler_blend"] = sampler_step_roofline(pipe)
    check = None
    inproc_fallback = None
    if world > 1 and not args.no_check:
        check = multi_gpu_check(pipe, model, inp, cond, (args.warmup + args.steps) % total_steps, total_steps, world, rank)
        # Started by someone else's torch.distributed.run (not by launch_ranks, which has its own second attempt): a failed check of
        # the default exchange gets ONE more measurement in this process on the conservative form of the same exchange -- one K|V
        # all-gather per block, waited for before attention.  Every rank takes the same decision (check.ok is built from all-gathered
        # values); the engine is rebuilt on its next use (multi_gpu_check leaves model._engine = None) and reads the switches again.
        if not check["ok"] and os.environ.get("FLEXAM_BENCH_SPAWNED") != "1" and eng.sp_size > 1 and not pinned:
            inproc_fallback = {"attempt": "default", "check": check, "ms_per_step": elapsed / args.steps * 1e3}
            os.environ["FLEXAM_SP_PIECES"], os.environ["FLEXAM_SP_OVERLAP"] = "1", "0"
            os.environ.pop("FLEXAM_SP_MODE", None)        # the conservative form IS the K|V all-gather, whatever the probe had picked
            model.enable_multi_gpus_inference(cfg_parallel=None)
            model._engine = None
            pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
            eng = model.engine()
            elapsed = timed(pipe)
            tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            finite = bool(torch.isfinite(pipe._state["latents"]).all())
            check = multi_gpu_check(pipe, model, inp, cond, (args.warmup + args.steps) % total_steps, total_steps, world, rank)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(L, cfg)
        if not args.no_cpu_legs:
            base["legs"] = cpu_baseline_legs(cfg)

    eng_cfg, eng_sp, eng_mode = eng.cfg_size, eng.sp_size, getattr(eng, "sp_mode", "-")
    eng_cross_lk = eng.cond.get("cross_lk")
    eng_share0 = bool(getattr(eng, "share0_taken", False))      # what the engine DID in its last run, not what a switch asked for
    if eng_mode == "allgather":
        pcs = getattr(eng, "sp_pieces", 1)
        eng_mode = (f"K|V all-gather per block in {pcs} head-group piece(s), " +
                    ("each consumed as it lands: Q projection + local-chunk attention under piece 0 (partial softmaxes merged), piece g+1 under the attention of group g"
                     if getattr(eng, "sp_overlap", False) else "waited for before attention (overlaps the Q projection only)"))
    elif eng_mode == "ulysses":
        lvl = getattr(eng, "sp_overlap_level", 0)
        eng_mode = "all-to-all over heads (q|k|v out, attention output back)" + (
            (", a sample's blocks leave under the other sample's projection, one attention call for the pair" if lvl == 1 else
             ", the samples of the CFG pair as pipeline stages: a sample's blocks travel under the other's projection / attention")
            if (b_local > 1 and lvl > 0) else "")
    vae_sec = enc_sec = enc_stream_sec = None
    clip = None
    if rank == 0 and world == 1 and not args.no_vae:
        del pipe, eng
        torch.cuda.empty_cache()
        vae_sec, enc_stream_sec, enc_image_sec, vae_finite, vae = time_vae(device, args.frames, args.height, args.width)
        enc_sec = 7 * enc_stream_sec + enc_image_sec     # control, depth, 4 cos levels, masked video + the reference image
        finite = finite and vae_finite
        if not args.no_clip and args.mask == "motion":
            try:                                         # the measured clip must not cost the steps/s line if it fails
                sec, shape, ok = time_clip(model, vae, inp, args.frames, args.height, args.width, total_steps, device)
                clip = {"sec": sec, "steps": total_steps, "output_shape": list(shape), "finite": ok,
                        "what": "Wan2_2FunControlPipeline_FlexAM.__call__ end to end: 8 pixel-space conditioning streams -> VAE encode -> "
                                "50 denoise steps -> VAE decode -> frames on the host (synthetic pixels, prompt embeddings given)"}
                finite = finite and ok
            except Exception as e:                       # noqa: BLE001
                clip = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        steps_per_sec = args.steps / elapsed
        blk = block_flops(L, cfg["dim"], cfg["ffn_dim"], cfg["text_len"])
        step_block_flops = blk * B * cfg["num_layers"]
        # Work the build removes is not counted as achieved FLOPs: with the CFG pair on one latent, block 0's q|k|v / o
        # projections and self-attention run once, not per sample (DiTEngine.run, share0)
        shared0 = eng_share0
        removed_flops = (8 * L * cfg["dim"] ** 2 + 4 * L * L * cfg["dim"]) if shared0 else 0
        # ... and the identical padded text rows are attended to as one weighted key (DiTEngine.set_conditioning, cross_lk)
        cross_lk = eng_cross_lk if eng_cross_lk else cfg["text_len"]
        removed_flops += 4 * L * (cfg["text_len"] - cross_lk) * cfg["dim"] * B * cfg["num_layers"]
        executed_block_flops = step_block_flops - removed_flops
        result = {
            "metric": "denoise-steps/sec", "value": steps_per_sec, "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": ("fp8" if args.fp8 else "bf16") + (" + mxfp8 self-attention" if args.sage else "") + (" + fp8 output projections" if args.fp8_oproj else ""), "data": "synthetic",
            "config": {"workload": f"Wan2.2-Fun-5B-FLEXAM DiT denoise step, {args.frames}x{args.height}x{args.width}, "
                                   f"L={L} tokens, CFG pair B=2, {cfg['num_layers']} layers, flow-match Euler (50-step schedule), "
                                   f"random-init bf16 weights, synthetic conditioning "
                                   + (("(BASELINE configs[1])" if args.mask == "motion" else f"(BASELINE configs[3]: foreground_edit, mask '{args.mask}', "
                                       f"{rows_u} distinct per-token timesteps per sample)")
                                      if (args.frames, args.height, args.width) == (97, 512, 896) and not args.fp8 and not args.sage else
                                      "(BASELINE configs[4] shape family: " + ("fp8 e4m3 QKV/FFN GEMMs with per-row / per-channel scales, everything else bf16/fp32" if args.fp8 else "bf16") + (", self-attention on MXFP8 operands (VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION)" if args.sage else "") + (", o-projections on fp8 too (FLEXAM_FP8_OPROJ=1)" if args.fp8_oproj else "")
                                      + "; not the headline)"),
                       "parallelism": ((f"cfg{eng_cfg} x sp{eng_sp}: one CFG row per rank" + (", no per-block traffic" if eng_sp == 1 else
                                        f", token-chunk sequence parallel inside each half; exchange around self-attention (RCCL): {eng_mode}"))
                                       if eng_cfg == 2 else
                                       f"cfg1 x sp{eng_sp}: CFG pair batched on every rank, token-chunk sequence parallel over all ranks; "
                                       f"exchange around self-attention (RCCL): {eng_mode}") if world > 1 else "single GPU",
                       "layers": cfg["num_layers"]},
            "sec_per_clip_50_steps_denoise_only": total_steps / steps_per_sec,
            "vae_decode_sec": vae_sec, "vae_encode_sec_per_stream": enc_stream_sec, "conditioning_encode_sec_8_streams": enc_sec,
            "prepare_sec": prepare_sec,
            "clip_end_to_end": clip,
            "sec_per_clip": (enc_sec + prepare_sec + total_steps / steps_per_sec + vae_sec) if vae_sec is not None else None,
            "dit_block_executed_tflops": executed_block_flops * steps_per_sec / 1e12,
            "dit_block_executed_mfma_frac": executed_block_flops * steps_per_sec / 1e12 / (PEAK_BF16_TFLOPS * world),
            "dit_block_algorithmic_tflops": step_block_flops * steps_per_sec / 1e12,
            "dit_block_flops_note": ("executed = algorithmic block FLOPs (%.3f TF per block and sample x %d) minus what the build removes -- the block-0 self-attention "
                                     "half the CFG pair shares (taken: %s) and the identical padded text keys of cross-attention (keys attended: %d of %d): %.3f of "
                                     "%.1f TF per step; the fraction is executed FLOPs against the 2.5 PFLOP/s bf16 peak.  The algorithmic figure counts work that "
                                     "did not run: it is a throughput in the reference's units, not a roofline fraction (rounds 1-2 reported it as "
                                     "dit_block_tflops with every FLOP executed)"
                                     % (blk / 1e12, B * cfg["num_layers"], shared0, cross_lk, cfg["text_len"], removed_flops / 1e12, step_block_flops / 1e12))
                                    + (" (QKV / FFN ran on the 5 PFLOP/s fp8 pipe: not a roofline fraction)" if args.fp8 else ""),
            "finite": finite,
            "mask": args.mask, "timestep_rows_per_sample": rows_u,
            "host_enqueue_ms_per_step": host_enqueue * 1e3,
        }
        if motion_elapsed is not None:
            result["configs1_same_process"] = {"ms_per_step": motion_elapsed / args.steps * 1e3, "value": args.steps / motion_elapsed,
                                               "delta_pct": (elapsed / motion_elapsed - 1.0) * 100.0,
                                               "note": "the same steps on the motion_transfer conditioning (BASELINE configs[1]) in this process"}
        if layout_probe is not None:
            result["layout_probe"] = layout_probe
        if inproc_fallback is not None:
            result["launch"] = {"spawned_by": "the caller's torch.distributed.run", "attempt": "FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0 (second measurement in the same processes)",
                                "earlier_attempts": [inproc_fallback], "fallback": True}
        if check is not None:
            result["check"] = check
            result["rccl_ranks"] = world if backend == "nccl" else 0
            result["ranks_agree"] = check["ranks_agree"]
            if not check["ok"]:
                result["invalid"] = "multi-GPU self-check failed: " + json.dumps(check)
        if one_device or backend != "nccl":
            result["invalid"] = f"code-path validation only: {world} ranks on one device / backend {backend}"
        if kern is not None:
            a = kern["attn_self"]
            traffic = None                       # HBM bytes per launch from the committed PMC passes (not collected live)
            tpath = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r3_attn_traffic.json", "r2_attn_traffic.json", "r1m_attn_traffic.json"))
                          if os.path.exists(q)), "")
            if world == 1 and (args.frames, args.height, args.width) == (97, 512, 896) and os.path.exists(tpath):
                traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
            sec_live = attn_in_step["sec"] if attn_in_step else a["sec"]
            result["roofline"] = {"bound": "mfma", "kernel": "attn_fwd_kernel<0, true, true> (self-attention, head_dim 128, q pre-scaled by its RMSNorm weight; the one-basic-block-step instance)",
                                  "achieved": a["flops"] / sec_live / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": a["flops"] / sec_live / 1e12 / PEAK_BF16_TFLOPS,
                                  "traffic": traffic,
                                  "traffic_note": "bytes/launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over the launches of one call (" + os.path.basename(tpath) + ")",
                                  "launch_ms": sec_live * 1e3, "flops_per_launch": a["flops"],
                                  "timed": ("IN the step: HIP events around the %d whole-batch self-attention calls of one denoise step (the other %d call(s) run one "
                                            "sample: block 0 shared by the CFG pair); this is what rocprofv3 --kernel-trace --stats of the same command averages for "
                                            "the kernel (+ the 20 us merge)" % (attn_in_step["calls"], attn_in_step["calls_other_batch"])) if attn_in_step
                                           else "isolated back-to-back launches (no in-step timing in this run)",
                                  "launch_ms_all_calls": attn_in_step["sec_all_calls"] * 1e3 if attn_in_step else None,
                                  "launch_ms_all_calls_note": "mean over ALL self-attention calls of that step (the one-sample call of block 0 included) + the merge: "
                                                              "compare with the kernel's AverageNs in rocprofv3 --kernel-trace --stats of `bench.py --no-kernel-timing`",
                                  "isolated_launch_ms": a["sec"] * 1e3, "isolated_frac": a["tflops"] / PEAK_BF16_TFLOPS,
                                  "isolated_note": "8 back-to-back launches on the step's own buffers: runs at the clock the kernel holds alone, not the step's",
                                  "launch_note": "one self-attention call = ONE attn_fwd_kernel<0, true, true> launch (the full rounds of work units and, on the same "
                                                 "XCDs behind them, the last partial round with its keys cut in 3) + attn_merge_kernel; launch_ms is the whole "
                                                 "call = its AverageNs in rocprofv3 + the merge"}
            if args.sage and attn_in_step:               # the dominant kernel of THIS line is the MXFP8 one: priced against the fp8 pipe
                r = result["roofline"]
                r.update(kernel="attn8_fwd_kernel<0> (self-attention on MXFP8 operands, csrc/attn_fp8.inc)", peak=PEAK_FP8_TFLOPS,
                         frac=r["achieved"] / PEAK_FP8_TFLOPS, traffic=None, isolated_launch_ms=None, isolated_frac=None,
                         isolated_note="not timed alone in this run", launch_note="one call = ONE attn8_fwd_kernel launch + attn_merge_kernel; the "
                         "attn8_pack_kernel launch in front of it (0.12 ms) is not part of launch_ms")
            result["kernels"] = {k: ({"ms": round(v["sec"] * 1e3, 4), "tflops": round(v["tflops"], 1), "bound": "mfma",
                                      "peak": PEAK_FP8_TFLOPS if k.startswith("gemm_fp8") else PEAK_BF16_TFLOPS,
                                      "frac": round(v["tflops"] / (PEAK_FP8_TFLOPS if k.startswith("gemm_fp8") else PEAK_BF16_TFLOPS), 4)} if "flops" in v else
                                     {"ms": round(v["sec"] * 1e3, 4), "gbs": round(v["bytes"] / v["sec"] / 1e9, 1), "bound": "hbm",
                                      "frac": round(v["bytes"] / v["sec"] / 1e9 / PEAK_HBM_GBS, 4)}) for k, v in kern.items()}
            result["kernels_note"] = ("live per-launch timing at this run's shapes; mfma rows: algorithmic FLOPs / 2.5 PFLOP/s, hbm rows: "
                                      "algorithmic bytes (SURVEY 8d) / 8 TB/s; counter-side traffic and MFMA-busy: profiles/r4ae_block_kernels_pmc.txt")
        if base is not None:
            result["cpu_baseline"] = base
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
