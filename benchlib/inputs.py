"""The synthetic workload bench.py measures (BASELINE configs[1] / [3] / [4]): seeded inputs, masks, the random-init model, FLOP counts."""
import math

import torch

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



def set_logit_scale(model, logit_std: float):
    """bench.py --logit-scale S: every `blocks.*.self_attn.norm_q / norm_k` weight times sqrt(S).  q and k leave the full-width RMSNorm
    with unit RMS per element, so the random-init model's scores q.k / sqrt(128) are ~N(0, 1); afterwards ~N(0, S^2): rows with a few
    dominant keys, maxima tens of exp2 units above the mean, the flash kernel's running reference moved many times per row."""
    f = math.sqrt(float(logit_std))
    with torch.no_grad():
        for blk in model.blocks:
            blk.self_attn.norm_q.weight.mul_(f)
            blk.self_attn.norm_k.weight.mul_(f)
