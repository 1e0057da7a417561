"""MI355X drop-in for `Wan2_2FunControlPipeline_FlexAM` (the FlexAM sampler).

Call contract = FlexAM/pipeline/pipeline_wan2_2_fun_control_FlexAM.py:505-965 (`__call__` keyword
signature, `WanPipelineOutput(videos=...)`).  The hot loop (:840-949) is restructured for the GPU:
everything that does not change across steps (the conditioning concatenations, the per-token
timestep *pattern*, the DiT's step-invariant work) is prepared once (`prepare`), and each step is
one DiT engine run + ONE fused kernel for CFG + Euler + masked blend on fp32 latents
(`denoise_step`).  Arithmetic is in libflexam_hip.so; torch is used for allocation, slicing and the
one-off mask/latent preparation.

Two ways in:
  * `__call__(prompt_embeds=..., video=..., control_video=..., ...)`: pixel-space inputs like demo.py; the
    conditioning streams go through `vae.encode` (HIP) in `encode_conditioning`; a `prompt=` string goes through
    the attached tokenizer + `WanT5EncoderModel` (HIP), `prompt_embeds=` skips them;
  * `__call__(..., conditioning=LatentConditioning(...))`: latent-space conditioning (what bench.py,
    the tests and a caller that has already VAE-encoded its streams use).
"""
import math
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch
import torch.nn.functional as F

from . import hip
from .fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .fm_solvers_unipc import FlowUniPCMultistepScheduler
from .scheduler import FlowMatchEulerDiscreteScheduler

F32 = torch.float32


@dataclass
class WanPipelineOutput:
    videos: torch.Tensor


@dataclass
class LatentConditioning:
    """Latent-space inputs of the denoise loop (what PIPE.py:655-822 produces with the VAE).
    All [1, C, F, H, W] except ref_latents [1, C, H, W]; mask_pixels [1,1,Fpix,Hpix,Wpix] in {0,1}
    (1 = regenerate) or pre-resized mask_latents [1,4,F,H,W] + mask [1,1,F,H,W]."""
    control_latents: torch.Tensor
    additional_control: torch.Tensor            # depth + cos levels, 240 channels
    masked_video_latents: torch.Tensor
    ref_latents: Optional[torch.Tensor] = None
    mask_pixels: Optional[torch.Tensor] = None
    mask_latents: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None


def resize_mask(mask, latent_size, first_frame_only=True):
    """PIPE.py:108-134 (trilinear, first frame resized on its own)."""
    tgt = list(latent_size[2:])
    if not first_frame_only:
        return F.interpolate(mask, size=tgt, mode="trilinear", align_corners=False)
    first = F.interpolate(mask[:, :, 0:1], size=[1] + tgt[1:], mode="trilinear", align_corners=False)
    if tgt[0] == 1:
        return first
    rest = F.interpolate(mask[:, :, 1:], size=[tgt[0] - 1] + tgt[1:], mode="trilinear", align_corners=False)
    return torch.cat([first, rest], dim=2)


def prepare_masks(mask_condition, latent_size):
    """PIPE.py:675-690: (mask_latents [B,4,F,H,W], mask [B,1,F,H,W], pinned)."""
    b, _, fpix, hpix, wpix = mask_condition.shape
    mc = torch.cat([torch.repeat_interleave(mask_condition[:, :, 0:1], repeats=4, dim=2), mask_condition[:, :, 1:]], dim=2)
    mc = mc.view(b, mc.shape[2] // 4, 4, hpix, wpix).transpose(1, 2)
    mask_latents = resize_mask(1 - mc, latent_size, True)
    mask = F.interpolate(mc[:, :1], size=list(latent_size[-3:]), mode="trilinear", align_corners=True)
    pinned = not bool(mask[:, :, 0].any())
    if pinned:
        mask[:, :, 1:] = 1
    return mask_latents, mask, pinned


class Wan2_2FunControlPipeline_FlexAM:
    _optional_components = ["transformer_2"]
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, transformer_2=None, scheduler=None):
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer, self.transformer_2 = transformer, transformer_2
        self.scheduler = scheduler if scheduler is not None else FlowMatchEulerDiscreteScheduler(1000, shift=5.0)
        if transformer_2 is not None:
            raise NotImplementedError("the two-expert (transformer_2) Wan2.2-A14B layout is not the FlexAM 5B path")
        self._interrupt = False
        self._guidance_scale = 6.0
        self._num_timesteps = 0
        self._state = None
        # The reference builds the per-token timesteps as `mask.to(weight_dtype) * t` (PIPE.py:679,891-898): with the bf16
        # checkpoint both the mask and the product are bf16, so the DiT embeds bf16-ROUNDED timesteps (multiples of 4 in
        # [512, 1024)).  None = follow the transformer's dtype like the reference does; torch.float32 = exact timesteps.
        self.timestep_dtype = None

    # ------------------------------------------------------------------ housekeeping (reference surface)
    def to(self, device):
        for m in (self.transformer, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    @property
    def device(self):
        return self.transformer.device

    _execution_device = device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    def maybe_free_model_hooks(self):
        pass

    def check_inputs(self, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds=None,
                     negative_prompt_embeds=None):
        """Same errors as PIPE.py:436-485."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`.")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if torch.is_tensor(prompt_embeds) and torch.is_tensor(negative_prompt_embeds) and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly")

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, prompt_embeds=None,
                      negative_prompt_embeds=None, max_sequence_length=512, device=None):
        """PIPE.py:234-313: lists of per-prompt [len_i, text_dim] embeddings (T5 output trimmed to the true token
        count).  `text_encoder` is `flexam_amd.WanT5EncoderModel` (HIP) or any module with the reference's call."""
        def as_list(e):
            return list(e) if e is not None else None
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise NotImplementedError("no tokenizer / text encoder attached: pass prompt_embeds / negative_prompt_embeds, or "
                                          "construct the pipeline with tokenizer= and text_encoder=WanT5EncoderModel")
            prompt = [prompt] if isinstance(prompt, str) else prompt
            prompt_embeds = self._t5(prompt, max_sequence_length, device)
            if do_classifier_free_guidance and negative_prompt_embeds is None:
                neg = negative_prompt or ""
                neg = len(prompt) * [neg] if isinstance(neg, str) else neg
                negative_prompt_embeds = self._t5(neg, max_sequence_length, device)
        return as_list(prompt_embeds), as_list(negative_prompt_embeds)

    def _t5(self, prompts, max_len, device):
        ids = self.tokenizer(prompts, padding="max_length", max_length=max_len, truncation=True, add_special_tokens=True, return_tensors="pt")
        lens = ids.attention_mask.gt(0).sum(dim=1).long()
        emb = self.text_encoder(ids.input_ids.to(device), attention_mask=ids.attention_mask.to(device))[0]
        return [u[:v] for u, v in zip(emb, lens)]

    # ------------------------------------------------------------------ the hot path
    @torch.no_grad()
    def prepare(self, latents, cond: LatentConditioning, context_cond, context_uncond, density, guidance_scale, num_inference_steps,
                timesteps=None, shift: float = 5):
        """Everything step-invariant (PIPE.py:598-605, 655-690, 833-842, 850-898 hoisted out of the loop)."""
        tr = self.transformer
        eng = tr.engine()
        dev = eng.device
        self._guidance_scale = guidance_scale
        cfg = guidance_scale > 1.0
        latents = latents.to(dev, F32)
        b, c, f, h, w = latents.shape
        if b != 1:
            raise NotImplementedError("one clip per call (the reference fixes num_videos_per_prompt = 1, PIPE.py:555)")
        if cond.mask_latents is not None:
            mask_latents, mask = cond.mask_latents.to(dev, F32), cond.mask.to(dev, F32)
            pinned = not bool(mask[:, :, 0].any())
        else:
            mask_latents, mask, pinned = prepare_masks(cond.mask_pixels.to(dev, F32), latents.shape)
        known = cond.masked_video_latents.to(dev, F32)
        if pinned:
            latents = (1 - mask) * known + mask * latents                       # PIPE.py:690 (once, torch)
        y = torch.cat([cond.control_latents.to(dev, F32), mask_latents, known], dim=1)   # PIPE.py:868-875
        context = (list(context_uncond) + list(context_cond)) if cfg else list(context_cond)
        if eng.cfg_size == 2:                      # CFG-parallel: this rank carries one row of the pair
            context = [context[eng.cfg_row]] if cfg else list(context_cond)
        nrow = len(context)
        dens = torch.full((nrow,), float(density), device=dev, dtype=F32)
        eng.set_conditioning(context, y, cond.ref_latents, cond.additional_control, dens, (c, f, h, w), shared=True)
        tr._cond_key = None                        # the engine's per-clip state no longer belongs to a forward() call
        cd = eng.cond
        # per-token timestep pattern: t * mask[::2, ::2] (PIPE.py:891-898); ref tokens take the last value
        tdt = self.timestep_dtype if self.timestep_dtype is not None else tr.dtype
        tdt = tdt if tdt in (torch.bfloat16, torch.float16) else F32
        sub = mask[0, 0, :, ::2, ::2].reshape(-1).to(tdt).to(F32)               # the reference's mask lives in weight_dtype
        seq = torch.cat([sub[-1:].repeat(cd["ref_len"]), sub]) if cd["ref_len"] else sub
        uniq, inv = torch.unique(seq, return_inverse=True)
        U = uniq.numel()
        row_index = torch.cat([inv + r * U for r in range(nrow)]).to(torch.int32).contiguous()
        # PIPE.py:603-616: each scheduler family builds its schedule its own way
        if isinstance(self.scheduler, FlowMatchEulerDiscreteScheduler):
            self.scheduler.set_timesteps(num_inference_steps, device=None) if timesteps is None else self.scheduler.set_timesteps(timesteps=timesteps)
        elif isinstance(self.scheduler, FlowUniPCMultistepScheduler):
            self.scheduler.set_timesteps(num_inference_steps, device=None, shift=shift)
        elif isinstance(self.scheduler, FlowDPMSolverMultistepScheduler):
            retrieve_timesteps(self.scheduler, device=None, sigmas=get_sampling_sigmas(num_inference_steps, shift))
        else:
            self.scheduler.set_timesteps(num_inference_steps)
        self._num_timesteps = len(self.scheduler.timesteps)
        tr.num_inference_steps = num_inference_steps
        self._state = dict(latents=latents[0].contiguous(), known=known[0].contiguous() if pinned else None,
                           mask=mask[0, 0].contiguous() if pinned else None, uniq=uniq.to(F32), tdt=tdt, row_index=row_index, U=U, nrow=nrow,
                           cfg=cfg, guidance=float(guidance_scale), ref_len=cd["ref_len"], shape=(c, f, h, w))
        return self._state

    @torch.no_grad()
    def denoise_step(self, i: int):
        """One iteration of PIPE.py:844-949 = DiT on the CFG pair + fused CFG/Euler/blend."""
        st, tr = self._state, self.transformer
        eng = tr.engine()
        tr.current_steps = i
        t = float(self.scheduler.timesteps[i])
        t_rows = (st["uniq"].to(st["tdt"]) * t).to(F32).repeat(st["nrow"])        # rounded like `mask.to(weight_dtype) * t`
        skip_uncond = (st["cfg"] and tr.cfg_skip_ratio is not None and tr.num_inference_steps is not None
                       and i >= tr.num_inference_steps * (1 - tr.cfg_skip_ratio))
        c, f, h, w = st["shape"]
        if skip_uncond and eng.cfg_size == 1:
            # cfg_skip (cfg_optimization.py:5-37): only the conditional row runs and is duplicated, so
            # uncond + g (cond - uncond) = cond
            # TeaCache keeps counting on these B = 1 forwards exactly as the reference's forward does (FX.py:977-1006,1119-1122)
            U = st["U"]
            head = eng.gather_tokens(eng.run(st["latents"].unsqueeze(0), t_rows[:U], st["row_index"][: st["row_index"].numel() // st["nrow"]],
                                             U, only_row=st["nrow"] - 1, teacache=tr.teacache))
            self._teacache_tick(tr.teacache)
            return self._sampler_update(i, head[0], None)
        tc = tr.teacache
        head = eng.gather_tokens(eng.run(st["latents"].unsqueeze(0), t_rows, st["row_index"], st["U"], teacache=tc, rows_shared=True))
        self._teacache_tick(tc)
        # head: [rows, L, 192] with rows = (uncond, cond) after the gather, whatever the parallel layout
        if skip_uncond:                                  # CFG-parallel ranks: both rows were computed anyway, take cond
            return self._sampler_update(i, head[1], None)
        return self._sampler_update(i, head[0], head[1] if st["cfg"] else None)

    @staticmethod
    def _teacache_tick(tc):
        """FX.py:1119-1122: one count per conditional forward; the cache resets itself after `num_steps` of them."""
        if tc is not None:
            tc.cnt += 1
            if tc.cnt == tc.num_steps:
                tc.reset()

    def _sampler_update(self, i: int, tok_uncond, tok_cond):
        """PIPE.py:926-934: CFG combine, scheduler.step, masked blend with the known latents.  Flow-match Euler: ONE
        fused kernel.  Multistep samplers (UniPC, DPM-Solver++): guided velocity -> scheduler.step (its updates are
        single `flexam_lincomb_f32` launches) -> `flexam_mask_blend_f32`."""
        st = self._state
        if isinstance(self.scheduler, FlowMatchEulerDiscreteScheduler):
            hip.cfg_euler_blend(tok_uncond, tok_cond, st["ref_len"], st["guidance"], self.scheduler.sigma_step(i), st["latents"],
                                st["known"], st["mask"])
            return st["latents"]
        if "velocity" not in st:
            st["velocity"] = torch.empty_like(st["latents"])
        v = hip.cfg_velocity(tok_uncond, tok_cond, st["ref_len"], st["guidance"], st["velocity"])
        new = self.scheduler.step(v.unsqueeze(0), self.scheduler.timesteps[i], st["latents"].unsqueeze(0), **st.get("step_kwargs", {}),
                                  return_dict=False)[0]
        st["latents"].copy_(new[0])
        if st["mask"] is not None:
            hip.mask_blend(st["latents"], st["known"], st["mask"])
        return st["latents"]

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """PIPE.py:410-415 (returns a CPU fp32 tensor instead of a numpy array)."""
        frames = self.vae.decode(latents).sample
        return (frames.float() / 2 + 0.5).clamp(0, 1).cpu()

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, height: int = 480, width: int = 720, video=None, mask_video=None,
                 control_video=None, depth_video=None, cos_level: int = 4, cos_control_videos=None, density: float = 1.0,
                 control_camera_video=None, start_image=None, ref_image=None, num_frames: int = 49, num_inference_steps: int = 50,
                 timesteps=None, guidance_scale: float = 6, num_videos_per_prompt: int = 1, eta: float = 0.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type: str = "numpy", return_dict: bool = False,
                 callback_on_step_end: Optional[Callable] = None, attention_kwargs: Optional[Dict[str, Any]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512, boundary: float = 0.875,
                 comfyui_progressbar: bool = False, shift: int = 5, conditioning: Optional[LatentConditioning] = None):
        self.check_inputs(prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds, negative_prompt_embeds)
        self._interrupt = False
        if control_camera_video is not None or start_image is not None:
            raise NotImplementedError("control_camera_video / start_image are unused by FlexAM (pipelines.py:1181)")
        dev = self.transformer.device
        cfg = guidance_scale > 1.0
        ctx_c, ctx_u = self.encode_prompt(prompt, negative_prompt, cfg, prompt_embeds, negative_prompt_embeds, max_sequence_length, dev)
        if cfg and ctx_u is None:
            raise ValueError("classifier-free guidance needs negative_prompt_embeds (or a text encoder)")
        zc = self.vae.config.latent_channels if self.vae is not None else self.transformer.out_dim
        tcr = self.vae.temporal_compression_ratio if self.vae is not None else 4
        scr = self.vae.spatial_compression_ratio if self.vae is not None else 16
        shape = (1, zc, (num_frames - 1) // tcr + 1, height // scr, width // scr)
        if latents is None:
            gen_dev = generator.device if generator is not None else "cpu"
            latents = torch.randn(shape, generator=generator, device=gen_dev, dtype=F32)
        if tuple(latents.shape) != shape:
            raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
        if conditioning is None:
            conditioning = self.encode_conditioning(video, mask_video, control_video, depth_video, cos_control_videos, ref_image,
                                                    height, width, shape)
        self.prepare(latents, conditioning, ctx_c, ctx_u, density, guidance_scale, num_inference_steps, timesteps, shift)
        # PIPE.py:417-433, 836, 931: the generator reaches scheduler.step of the schedulers that take one (the SDE sampler's noise)
        import inspect
        self._state["step_kwargs"] = ({"generator": generator} if "generator" in inspect.signature(self.scheduler.step).parameters else {})
        for i in range(self._num_timesteps):
            if self._interrupt:
                continue
            lat = self.denoise_step(i)
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, self.scheduler.timesteps[i], {"latents": lat.unsqueeze(0)})
                if out and "latents" in out:
                    self._state["latents"].copy_(out["latents"][0])
        final = self._state["latents"].unsqueeze(0)
        if output_type == "latent":
            video_out = final
        else:
            video_out = self.decode_latents(final)
        self.maybe_free_model_hooks()
        return WanPipelineOutput(videos=video_out)

    @staticmethod
    def _preprocess(v: torch.Tensor, height: int, width: int, mask: bool = False) -> torch.Tensor:
        """VaeImageProcessor.preprocess on a [B,C,F,H,W] tensor as PIPE.py:625-627, 657-659 use it: resize
        (torch nearest) when the size differs; pixels in [0,1] -> [-1,1] unless already signed; masks are
        reduced to one channel and binarised at 0.5 (0/255 masks binarise to 0/1)."""
        v = v.to(F32)
        b, c, f, h, w = v.shape
        if (h, w) != (height, width):
            v = F.interpolate(v.transpose(1, 2).reshape(b * f, c, h, w), size=(height, width)).view(b, f, c, height, width).transpose(1, 2)
        if mask:
            return (v[:, :1] >= 0.5).to(F32)
        return v * 2 - 1 if float(v.min()) >= 0 else v

    def encode_conditioning(self, video, mask_video, control_video, depth_video, cos_control_videos, ref_image, height, width, shape):
        """PIPE.py:623-822: pixel-space streams -> LatentConditioning through AutoencoderKLWan3_8.encode
        (posterior mode, PIPE.py:345-403).  video / control / depth / cos / ref: [1,3,F,H,W] in [0,1];
        mask_video [1,1,F,H,W] with 255 (or 1) = regenerate.  An all-255 mask gives zero mask latents,
        zero known latents and mask = 1 (PIPE.py:648-654)."""
        if self.vae is None or not getattr(self.vae, "supports_encode", False):
            raise NotImplementedError("pixel-space conditioning needs a VAE with an encode path; "
                                      "pass conditioning=LatentConditioning(...) with pre-encoded latents")
        if video is None or mask_video is None or control_video is None:
            raise ValueError("video, mask_video and control_video are required (predict_v2v.py:1181 always passes them)")
        if cos_control_videos is None or len(cos_control_videos) == 0:
            raise ValueError("cos_control_videos is mandatory for FlexAM (PIPE.py:744-773, 865-866)")
        dev = self.vae.device
        enc = lambda v: self.vae.encode(v.to(dev))[0].mode().float()
        pre = lambda v: None if v is None else self._preprocess(v, height, width)
        zeros = torch.zeros(shape, device=dev, dtype=F32)
        jobs = [("control", pre(control_video)), ("depth", pre(depth_video)), ("ref", pre(ref_image))]
        jobs += [(("cos", k), pre(cos_control_videos[k])) for k in sorted(cos_control_videos)]
        if bool((mask_video == 255).all()):
            mask_pixels = None
            mask_latents, mask = zeros[:, :1].repeat(1, 4, 1, 1, 1), torch.ones_like(zeros[:, :1])
        else:
            mask_pixels = self._preprocess(mask_video, height, width, mask=True)
            jobs.append(("masked", self._preprocess(video, height, width) * (mask_pixels < 0.5)))
            mask_latents = mask = None
        from .dist import shard_streams
        lat = shard_streams(jobs, enc)                # one process: plain loop; N ranks: stream j on rank j % N, then broadcast
        masked = lat.get("masked", zeros) if mask_pixels is not None else zeros
        cos = [lat[("cos", k)] if lat[("cos", k)] is not None else zeros for k in sorted(cos_control_videos)]
        depth = lat["depth"] if lat["depth"] is not None else zeros
        if lat["ref"] is not None:
            ref = lat["ref"][:, :, 0]
        else:
            ref = zeros[:, :, 0] if getattr(self.transformer, "ref_conv", None) is not None else None
        return LatentConditioning(control_latents=lat["control"], additional_control=torch.cat([depth] + cos, dim=1),
                                  masked_video_latents=masked, ref_latents=ref, mask_pixels=mask_pixels, mask_latents=mask_latents, mask=mask)
