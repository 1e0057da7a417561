"""GPU parity for BASELINE config 4 (foreground_edit): a per-frame blob mask drives the 4 mask
channels, the masked-video latent is zero inside the blob, frame 0 is kept (demo.py:107-111) so the
latent blend stays active.  Tiny DiT, 3 Euler steps, HIP sampler vs the fp32 oracle loop; PSNR >= 40 dB."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O
from oracle import sampler as S

pytestmark = pytest.mark.gpu


def test_foreground_edit_sampler_matches_oracle():
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM, Wan2_2Transformer3DModel_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    cfg = dict(O.DIT_TINY)
    kw = dict(cfg)
    kw.pop("eps")
    sd = C.dit_weights(cfg, 7)
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
    m.load_state_dict(sd, strict=True)
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=m.to("cuda:0"))
    sc = C.sampler_case(cfg, seed=33)
    # blob mask: 1 inside a disc that moves over the frames, frame 0 untouched
    fpix, hp, wp = sc["mask_pixels"].shape[2:]
    yy, xx = torch.meshgrid(torch.arange(hp), torch.arange(wp), indexing="ij")
    mask = torch.zeros(1, 1, fpix, hp, wp)
    for f in range(1, fpix):
        mask[0, 0, f] = (((yy - 100 - 6 * f) ** 2 + (xx - 90 - 9 * f) ** 2) < 60 ** 2).float()
    ml, mk, pinned = S.prepare_masks(mask, sc["latents"])
    assert pinned and 0 < float(ml.mean()) < 1                       # fractional mask-latent values exercise the 4 channels
    masked = sc["masked_video_latents"] * (1 - ml[:, :1])              # "zeroed" inside the blob
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], masked, sc["ref_latents"], mask_pixels=mask)
    out = pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
               num_inference_steps=3, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent")
    ref = S.denoise_loop(lambda **k: O.dit_forward(sd, cfg, **k), S.FlowMatchEulerSchedule(1000, 5.0), 3, sc["latents"], sc["context_uncond"],
                         sc["context_cond"], sc["control_latents"], sc["additional_control"], ml, masked, sc["ref_latents"], mk, pinned,
                         0.1, 6.0)
    p = C.psnr(out.videos.float().cpu(), ref)
    print(f"foreground_edit 3-step sampler: psnr {p:.1f} dB")
    assert p >= 40.0
    torch.testing.assert_close(out.videos[:, :, 0].float().cpu(), masked[:, :, 0])
