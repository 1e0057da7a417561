"""TEST INFRASTRUCTURE ONLY -- seeded parity cases shared by make_golden.py (reference side),
the CPU oracle tests and the GPU parity tests.

Inputs and weights are regenerated from seeds with a CPU torch.Generator (bit-identical in
the build container and on the GPU box: same image, same torch build); the committed
fixtures hold the *reference outputs* plus checksums of the regenerated inputs/weights so a
generator drift fails loudly instead of silently comparing different problems.
"""
import math
from typing import Dict

import torch

from . import dit as O
from . import vae as OV

Tensor = torch.Tensor


def checksum(tensors: Dict[str, Tensor]) -> Tensor:
    """Order-independent fingerprint: per-tensor (sum, abs-sum) in fp64, keyed by sorted name."""
    vals = []
    for k in sorted(tensors):
        t = tensors[k].to(torch.float64)
        vals += [t.sum().item(), t.abs().sum().item()]
    return torch.tensor(vals, dtype=torch.float64)


def randn(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float32)


# ----------------------------------------------------------------------------- DiT cases
def dit_case(cfg: dict, seed: int, frames: int = 3, h: int = 16, w: int = 16, batch: int = 2, per_token_t: bool = True,
             t_value: float = 731.5, text_lens=(5, 11)) -> dict:
    """Synthetic FlexAM DiT call (all FlexAM inputs: y, additional_control, full_ref, density).
    latent [B,48,frames,h,w]; tokens = frames*(h/2)*(w/2) (+ (h/2)*(w/2) ref tokens)."""
    g = torch.Generator().manual_seed(seed)
    c = cfg["out_dim"]
    x = randn(g, batch, c, frames, h, w)
    y = randn(g, batch, cfg["in_dim"] - c, frames, h, w)
    add = randn(g, batch, cfg["in_dim_cnn_block"] - c, frames, h, w)
    ref = randn(g, batch, cfg["in_dim_ref_conv"], h, w)
    ctx = [randn(g, text_lens[i % len(text_lens)], cfg["text_dim"]) for i in range(batch)]
    seq_len = frames * (h // 2) * (w // 2)
    if per_token_t:
        t = torch.full((batch, seq_len), t_value)
        t[:, : (h // 2) * (w // 2)] = 0.0            # frame-0 tokens pinned (PIPE.py:891-898)
    else:
        t = torch.tensor([500.0, 24.4, 1000.0, 0.0][:batch])
    dens = torch.full((batch,), 0.1)
    return dict(x=x, t=t, context=ctx, seq_len=seq_len, y=y, full_ref=ref, additional_control=add, density=dens)


def dit_case_odd(cfg: dict, seed: int, frames: int = 3, h: int = 15, w: int = 16, batch: int = 2, t_value: float = 731.5, text_lens=(5, 11)) -> dict:
    """A DiT call on a latent whose height / width is ODD (H/16 odd at the pixel level): the stride-2 patch convolution drops the last row /
    column, the pipeline's seq_len = ceil(h w / 4 * f) (PIPE.py:838-839) is longer than the token sequence -> the reference pads with zero
    tokens and masks them as keys (FX.py:918-925), and its per-token timesteps come from mask[::2, ::2] (ceil(h/2) x ceil(w/2) per frame,
    PIPE.py:891-898), lined up against the padded sequence."""
    g = torch.Generator().manual_seed(seed)
    c = cfg["out_dim"]
    x = randn(g, batch, c, frames, h, w)
    y = randn(g, batch, cfg["in_dim"] - c, frames, h, w)
    add = randn(g, batch, cfg["in_dim_cnn_block"] - c, frames, h, w)
    ref = randn(g, batch, cfg["in_dim_ref_conv"], h, w)
    ctx = [randn(g, text_lens[i % len(text_lens)], cfg["text_dim"]) for i in range(batch)]
    seq_len = math.ceil(h * w / 4 * frames)
    per_frame = math.ceil(h / 2) * math.ceil(w / 2)
    t = torch.full((batch, per_frame * frames), t_value)
    t[:, :per_frame] = 0.0
    return dict(x=x, t=t[:, :seq_len] if t.size(1) >= seq_len else t, context=ctx, seq_len=seq_len, y=y, full_ref=ref, additional_control=add,
                density=torch.full((batch,), 0.1))


def dit_weights(cfg: dict, seed: int) -> Dict[str, Tensor]:
    return O.seeded_state_dict(O.dit_param_shapes(cfg), seed)


def dit_weights_threaded(cfg: dict, seed: int) -> Dict[str, Tensor]:
    """Full-depth 5B-width state dicts (5 B parameters): one generator per tensor on a thread pool (oracle.dit.seeded_state_dict_threaded)."""
    import os
    return O.seeded_state_dict_threaded(O.dit_param_shapes(cfg), seed, threads=min(32, os.cpu_count() or 8))


def scale_self_attention_logits(sd: Dict[str, Tensor], logit_std: float) -> list:
    """Peaked softmax rows: multiplies every `blocks.*.self_attn.norm_q.weight` / `norm_k.weight` by sqrt(logit_std) IN PLACE.  q and k
    leave WanRMSNorm (FX.py:173-189, over the full width) with unit RMS per element, so q.k / sqrt(128) of two different tokens is
    ~N(0, 1) with the seeded weights; after the scaling it is ~N(0, logit_std^2): rows whose maximum sits tens of exp2 units above
    their mean, a few keys carrying the row, and -- in a kernel that keeps a running reference -- many reference moves.
    Returns the names it touched."""
    names = [k for k in sd if ".self_attn.norm_q.weight" in k or ".self_attn.norm_k.weight" in k]
    for k in names:
        sd[k].mul_(math.sqrt(logit_std))
    return names


def self_attention_row_stats(sd: Dict[str, Tensor], cfg: dict, case: dict, head: int = 0, tile: int = 64) -> dict:
    """What block 0's self-attention rows look like for a DiT call (FX.py:230-262 on the block-0 input): per-row statistics of the
    scores s = q.k / sqrt(D) * log2(e) (exp2 units) of ONE head of sample 0 -- std over keys, max - mean, the effective number of
    keys 1 / sum(p^2), and how far the row maximum lies above the maximum of the first `tile` keys (a flash kernel that keeps the
    first tile's maximum as its reference has to move it when this exceeds its threshold)."""
    one = dict(cfg, num_layers=1)
    taps = {}
    with torch.no_grad():
        O.dit_forward({k: v for k, v in sd.items() if not k.startswith("blocks.") or k.startswith("blocks.0.")}, one, taps=taps, **case)
        x, e0, dens0 = taps["x_embed"][:1], taps["e0"][:1], taps["dens0"][:1]
        p = "blocks.0"
        if e0.dim() > 3:
            e = [u.squeeze(2) for u in (sd[p + ".modulation"].unsqueeze(0) + e0).chunk(6, dim=2)]
        else:
            e = (sd[p + ".modulation"] + e0).chunk(6, dim=1)
        dm = (sd[p + ".modulation_density"] + dens0).chunk(2, dim=1)
        eps = cfg.get("eps", 1e-6)
        h = O.layer_norm(x, eps) * (1 + e[1]) + e[0] + dm[0]
        nh = cfg["num_heads"]
        hd = cfg["dim"] // nh
        l = h.shape[1]
        q = O.rms_norm(O.linear(sd, p + ".self_attn.q", h), sd[p + ".self_attn.norm_q.weight"], eps).view(1, l, nh, hd)
        k = O.rms_norm(O.linear(sd, p + ".self_attn.k", h), sd[p + ".self_attn.norm_k.weight"], eps).view(1, l, nh, hd)
        f = case["x"].shape[2] + (1 if case.get("full_ref") is not None else 0)
        grid = (f, case["x"].shape[3] // 2, case["x"].shape[4] // 2)
        ang = O.rope_angles(1024, hd)
        q, k = O.rope_apply(q, grid, ang)[0, :, head], O.rope_apply(k, grid, ang)[0, :, head]
        s = (q @ k.t()) / math.sqrt(hd) * math.log2(math.e)
        pr = torch.softmax(s * math.log(2.0), dim=-1)
        return dict(std=s.std(dim=1).mean().item(), max_minus_mean=(s.max(dim=1).values - s.mean(dim=1)).mean().item(),
                    n_eff=(1.0 / pr.pow(2).sum(dim=1)).mean().item(), keys=l,
                    over_first_tile=(s.max(dim=1).values - s[:, :tile].max(dim=1).values))


def block_case(dim: int = 256, ffn: int = 512, heads: int = 2, grid=(4, 8, 8), text: int = 16, seed: int = 11) -> dict:
    """One WanAttentionBlock (G3): L = prod(grid), per-token e0 with two distinct rows."""
    g = torch.Generator().manual_seed(seed)
    l = grid[0] * grid[1] * grid[2]
    x = randn(g, 2, l, dim)
    rows = randn(g, 2, 2, 6, dim) * 0.5                 # [B, 2 distinct, 6, C]
    sel = torch.zeros(l, dtype=torch.long)
    sel[grid[1] * grid[2]:] = 1
    e0 = rows[:, sel]                                   # [B, L, 6, C]
    dens0 = randn(g, 2, 2, dim) * 0.5
    ctx = randn(g, 2, text, dim)
    return dict(x=x, e0=e0, e_rows=rows, sel=sel, dens0=dens0, context=ctx, grid=grid, heads=heads, dim=dim, ffn=ffn)


def block_weights(dim: int, ffn: int, seed: int = 12) -> Dict[str, Tensor]:
    cfg = dict(O.DIT_TINY, dim=dim, ffn_dim=ffn, num_layers=1)
    shapes = {k[len("blocks.0."):]: v for k, v in O.dit_param_shapes(cfg).items() if k.startswith("blocks.0.")}
    return O.seeded_state_dict(shapes, seed)


TEACACHE_CASE = dict(coefficients=[1.0, 0.0], num_steps=6, thresh=2.0, skip_start=1,
                     t_values=[950.0, 900.0, 880.0, 860.0, 500.0, 480.0])


# ----------------------------------------------------------------------------- sampler case (BASELINE config 1)
def sampler_case(cfg: dict, seed: int = 21, frames: int = 3, h: int = 16, w: int = 16) -> dict:
    """9x256x256 -> latent [1,48,3,16,16]; motion_transfer mask (frame 0 known)."""
    g = torch.Generator().manual_seed(seed)
    c = cfg["out_dim"]
    latents = randn(g, 1, c, frames, h, w)
    control = randn(g, 1, c, frames, h, w)
    add = randn(g, 1, cfg["in_dim_cnn_block"] - c, frames, h, w)
    masked = randn(g, 1, c, frames, h, w)
    ref = randn(g, 1, cfg["in_dim_ref_conv"], h, w)
    ctx_u = [randn(g, 4, cfg["text_dim"]) * 0.1]
    ctx_c = [randn(g, 9, cfg["text_dim"]) * 0.1]
    fpix = 1 + 4 * (frames - 1)
    mask_pix = torch.ones(1, 1, fpix, h * 16, w * 16)
    mask_pix[:, :, 0] = 0                                # frame 0 kept (demo.py:88,107-111)
    return dict(latents=latents, control_latents=control, additional_control=add, masked_video_latents=masked,
                ref_latents=ref, context_uncond=ctx_u, context_cond=ctx_c, mask_pixels=mask_pix, density=0.1,
                guidance_scale=6.0, num_steps=4)


# ----------------------------------------------------------------------------- VAE cases
VAE_SMALL = dict(z_dim=48, dec_dim=16, dim_mult=(1, 2, 4, 4), temporal_up=(True, True, False))


def vae_weights(vcfg: dict, seed: int = 31, prefix: str = "model.") -> Dict[str, Tensor]:
    shp = OV.vae_decoder_param_shapes(vcfg["z_dim"], vcfg["dec_dim"], vcfg["dim_mult"], vcfg["temporal_up"], prefix)
    return O.seeded_state_dict(shp, seed)


def vae_case(seed: int = 32, frames: int = 3, h: int = 4, w: int = 4, z_dim: int = 48) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return randn(g, 1, z_dim, frames, h, w)


VAE_ENC_SMALL = dict(z_dim=48, dim=16, dim_mult=(1, 2, 4, 4), temporal_down=(False, True, True))


def vae_enc_weights(vcfg: dict, seed: int = 41, prefix: str = "model.") -> Dict[str, Tensor]:
    shp = OV.vae_encoder_param_shapes(vcfg["z_dim"], vcfg["dim"], vcfg["dim_mult"], vcfg["temporal_down"], prefix)
    return O.seeded_state_dict(shp, seed)


def vae_enc_case(seed: int = 42, frames: int = 9, h: int = 32, w: int = 64) -> Tensor:
    """Pixel clip in [-1, 1], [1, 3, 1+4k, h, w] (h, w multiples of 16)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(1, 3, frames, h, w, generator=g, dtype=torch.float32) * 2 - 1


# ----------------------------------------------------------------------------- multistep sampler cases (G10)
SOLVER_CASES = {                                    # name -> (kind, steps, shift, constructor kwargs)
    "unipc_o2": ("unipc", 8, 5.0, {}),
    "unipc_o3": ("unipc", 20, 3.0, dict(solver_order=3)),
    "dpmpp_o2": ("dpm", 8, 5.0, {}),
    "dpmpp_o3": ("dpm", 12, 5.0, dict(solver_order=3)),
    "dpmpp_heun": ("dpm", 6, 5.0, dict(solver_type="heun")),
    # the two corners of the DPM class the reference can run besides plain dpmsolver++ (r4): the SDE form (fm_solvers.py:473-477,
    # 568-580; its noise comes from a torch.Generator seeded with SOLVER_NOISE_SEED, drawn on the CPU as the reference does when the
    # generator is a CPU one) and dynamic thresholding of the x0 prediction (fm_solvers.py:291-326)
    "sde_dpmpp_o2": ("dpm", 8, 5.0, dict(algorithm_type="sde-dpmsolver++")),
    "sde_dpmpp_heun_o1": ("dpm", 6, 3.0, dict(algorithm_type="sde-dpmsolver++", solver_type="heun", solver_order=1)),
    "sde_dpmpp_heun": ("dpm", 7, 5.0, dict(algorithm_type="sde-dpmsolver++", solver_type="heun")),
    "dpmpp_thresholding": ("dpm", 6, 5.0, dict(thresholding=True, sample_max_value=1.5)),
}
SOLVER_NOISE_SEED = 77
# seeds of the cases' inputs: the five r1 cases keep the ids they had (their position in the sorted names of that time)
SOLVER_SEED_ID = {"dpmpp_heun": 0, "dpmpp_o2": 1, "dpmpp_o3": 2, "unipc_o2": 3, "unipc_o3": 4, "sde_dpmpp_o2": 5, "sde_dpmpp_heun_o1": 6,
                  "sde_dpmpp_heun": 7, "dpmpp_thresholding": 8}


def solver_case(name: str, shape=(1, 48, 3, 4, 6)):
    """Seeded start sample and per-step model outputs (velocity predictions) for a scheduler trace."""
    kind, steps, shift, kw = SOLVER_CASES[name]
    g = torch.Generator().manual_seed(1000 + SOLVER_SEED_ID[name])
    x = randn(g, *shape)
    vs = [randn(g, *shape) for _ in range(steps)]
    return kind, steps, shift, kw, x, vs


def psnr(a: Tensor, b: Tensor, peak: float = None) -> float:
    """PSNR of a vs reference b; peak defaults to the reference's max-abs range."""
    a, b = a.double(), b.double()
    mse = (a - b).pow(2).mean().item()
    if peak is None:
        peak = (b.max() - b.min()).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)
