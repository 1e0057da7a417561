"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FlexAM sampler hot loop.

Restates /root/reference/FlexAM/pipeline/pipeline_wan2_2_fun_control_FlexAM.py
("PIPE.py") lines 840-949 (denoise loop: CFG duplication, conditioning assembly,
per-token timestep, DiT call, CFG combine, Euler step, masked-latent blend) and
PIPE.py:655-690 (mask preparation), plus the third-party scheduler it drives.

PARITY UNPINNED for the scheduler: `diffusers.FlowMatchEulerDiscreteScheduler`
(requirements.txt: diffusers>=0.30.1, version not pinned) is absent from
/root/reference and from this image, and the reference holds no test or golden
vector for it.  `FlowMatchEulerSchedule` below restates the published algorithm
of diffusers>=0.30 (schedulers/scheduling_flow_match_euler_discrete.py) for the
configuration the reference constructs (pipelines.py:1146-1148 + yaml
scheduler_kwargs: num_train_timesteps=1000, shift=5.0, use_dynamic_shifting
false); parity is anchored on the reference's call sites (PIPE.py:604-605, 931)
and on the closed-form values quoted in SURVEY.md 8(c).  The loop itself (CFG,
per-token t, blend) IS pinned: golden G9 drives the *reference DiT module*
through this loop.
"""
import math
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class FlowMatchEulerSchedule:
    """diffusers FlowMatchEulerDiscreteScheduler (>=0.30) for use_dynamic_shifting=False.

    __init__: sigma_k = k/N_train, k = N_train..1, shifted s*sig/(1+(s-1)*sig).
    set_timesteps(n): linspace(sigma_max*N, sigma_min*N, n)/N, shifted AGAIN by the same
    formula (the well-known double shift), timesteps = sigma*N (float), sigmas += [0].
    step: x32 = x.float() + (sigma_next - sigma) * v ; cast to v.dtype.
    """
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 5.0):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_max = sig[0].item()
        self.sigma_min = sig[-1].item()
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self._step_index = None

    def set_timesteps(self, num_inference_steps: int):
        n_train = self.num_train_timesteps
        ts = np.linspace(self.sigma_max * n_train, self.sigma_min * n_train, num_inference_steps)
        sig = ts / n_train
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        sig = torch.from_numpy(sig).to(torch.float32)
        self.timesteps = sig * n_train
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._step_index = None
        return self.timesteps

    def step(self, model_output: Tensor, sample: Tensor) -> Tensor:
        if self._step_index is None:
            self._step_index = 0
        i = self._step_index
        dt = self.sigmas[i + 1] - self.sigmas[i]
        prev = sample.to(torch.float32) + dt * model_output
        self._step_index += 1
        return prev.to(model_output.dtype)


def resize_mask(mask: Tensor, latent_size, first_frame_only: bool = True) -> Tensor:
    """PIPE.py:108-134: trilinear resize, first frame handled on its own."""
    tgt = list(latent_size[2:])
    if not first_frame_only:
        return F.interpolate(mask, size=tgt, mode="trilinear", align_corners=False)
    first = F.interpolate(mask[:, :, 0:1], size=[1] + tgt[1:], mode="trilinear", align_corners=False)
    if tgt[0] == 1:
        return first
    rest = F.interpolate(mask[:, :, 1:], size=[tgt[0] - 1] + tgt[1:], mode="trilinear", align_corners=False)
    return torch.cat([first, rest], dim=2)


def prepare_masks(mask_condition: Tensor, latents: Tensor):
    """PIPE.py:675-690.  mask_condition [B,1,Fpix,Hpix,Wpix] in {0,1} (1 = regenerate),
    latents [B,C,F,H,W].  Returns (mask_latents [B,4,F,H,W], mask [B,1,F,H,W], pinned) where
    `pinned` says frame 0 of the mask was all zero (then mask[:, :, 1:] is forced to 1 and the
    blend of PIPE.py:690/934 is active)."""
    b, _, fpix, hpix, wpix = mask_condition.shape
    mc = torch.cat([torch.repeat_interleave(mask_condition[:, :, 0:1], repeats=4, dim=2), mask_condition[:, :, 1:]], dim=2)
    mc = mc.view(b, mc.shape[2] // 4, 4, hpix, wpix).transpose(1, 2)
    mask_latents = resize_mask(1 - mc, latents.size(), True)
    mask = F.interpolate(mc[:, :1], size=latents.size()[-3:], mode="trilinear", align_corners=True)
    pinned = not bool(mask[:, :, 0].any())
    if pinned:
        mask[:, :, 1:] = 1
    return mask_latents, mask, pinned


def per_token_timestep(mask: Tensor, t: Tensor, seq_len: int, batch: int) -> Tensor:
    """PIPE.py:891-898: t scaled by the 2x-subsampled latent mask, right-padded with t."""
    ts = (mask[0][0][:, ::2, ::2] * t).flatten()
    ts = torch.cat([ts, ts.new_ones(seq_len - ts.size(0)) * t])
    return ts.unsqueeze(0).expand(batch, ts.size(0))


def denoise_loop(model: Callable[..., Tensor], sched: FlowMatchEulerSchedule, num_steps: int, latents: Tensor,
                 context_uncond: List[Tensor], context_cond: List[Tensor], control_latents: Tensor,
                 additional_control: Tensor, mask_latents: Tensor, masked_video_latents: Tensor,
                 ref_latents: Optional[Tensor], mask: Tensor, pinned: bool, density: float,
                 guidance_scale: float = 6.0, patch=(1, 2, 2), trace: Optional[list] = None) -> Tensor:
    """PIPE.py:840-949 with spatial_compression_ratio >= 16 (Wan2.2 VAE) and init_video given.

    `model(x=, t=, context=, seq_len=, y=, full_ref=, additional_control=, density=)` is either
    oracle.dit.dit_forward bound to a state dict, or the reference nn.Module.  Everything is done
    in the dtype of `latents` (fp32 for the oracle; the reference's GPU path is bf16 here)."""
    timesteps = sched.set_timesteps(num_steps)
    c, f, h, w = latents.shape[1:]
    seq_len = math.ceil((h * w) / (patch[1] * patch[2]) * f)                      # PIPE.py:838-839
    cfg = guidance_scale > 1.0
    rep = 2 if cfg else 1
    context = (context_uncond + context_cond) if cfg else context_cond          # PIPE.py:598-601
    if pinned:
        latents = (1 - mask) * masked_video_latents + mask * latents              # PIPE.py:690
    dens = torch.tensor([density], dtype=latents.dtype)
    for t in timesteps:
        x_in = torch.cat([latents] * rep)                                         # :850
        y = torch.cat([torch.cat([control_latents] * rep),                        # :861-875
                       torch.cat([mask_latents] * rep), torch.cat([masked_video_latents] * rep)], dim=1)
        add = torch.cat([additional_control] * rep)
        full_ref = torch.cat([ref_latents] * rep) if ref_latents is not None else None
        ts = per_token_timestep(mask, t.to(latents.dtype), seq_len, x_in.shape[0])  # :891-898
        v = model(x=x_in, t=ts, context=context, seq_len=seq_len, y=y, full_ref=full_ref,
                  additional_control=add, density=dens.expand(x_in.shape[0]))
        if cfg:                                                                   # :926-928
            v_u, v_c = v.chunk(2)
            v = v_u + guidance_scale * (v_c - v_u)
        latents = sched.step(v, latents)                                          # :931
        if pinned:
            latents = (1 - mask) * masked_video_latents + mask * latents          # :933-934
        if trace is not None:
            trace.append(latents.clone())
    return latents
