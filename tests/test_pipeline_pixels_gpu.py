"""GPU: the pixel-space entry of the sampler (`__call__(video=..., control_video=..., ...)`, PIPE.py:623-822):
eight conditioning streams are VAE-encoded on the HIP path, the denoise loop runs, and the result is decoded.
Checked against the fp32 oracle composition: oracle VAE encode of every stream (pinned by golden G8) ->
restated loop (pinned by G9) -> oracle VAE decode (pinned by G7).  Tolerance: PSNR >= 40 dB on every encoded
stream, >= 35 dB on the final latents (they inherit the encode error through 3 CFG steps), >= 30 dB on pixels."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O
from oracle import sampler as S
from oracle import vae as OV

pytestmark = pytest.mark.gpu
H, W, FRAMES = 64, 96, 9


def make_pipe():
    from flexam_amd import AutoencoderKLWan3_8, Wan2_2FunControlPipeline_FlexAM, Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_TINY)
    kw = dict(cfg)
    kw.pop("eps")
    dit = Wan2_2Transformer3DModel_FlexAM(**kw)
    dsd = C.dit_weights(cfg, 7)
    dit.load_state_dict(dsd, strict=True)
    vae = AutoencoderKLWan3_8(latent_channels=48, c_dim=16, dec_dim=16, spatial_compression_ratio=16)
    vsd = dict(C.vae_weights(C.VAE_SMALL, seed=31))
    vsd.update(C.vae_enc_weights(C.VAE_ENC_SMALL, seed=41))
    vae.load_state_dict(vsd, strict=True)
    return Wan2_2FunControlPipeline_FlexAM(vae=vae.to("cuda:0"), transformer=dit.to("cuda:0")), cfg, dsd, vsd


def pixel_streams(seed=61):
    g = torch.Generator().manual_seed(seed)
    clip = lambda f=FRAMES: torch.rand(1, 3, f, H, W, generator=g)
    mask = torch.full((1, 1, FRAMES, H, W), 255.0)
    mask[:, :, 0] = 0                                            # motion transfer: frame 0 kept
    return dict(video=clip(), control_video=clip(), depth_video=clip(), cos_control_videos={lv: clip() for lv in (3, 0, 2, 1)},
                ref_image=clip(1), mask_video=mask)


def oracle_latents(vsd, px):
    enc = lambda v: OV.vae_encode(vsd, v * 2 - 1, C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN, OV.LATENT_STD)
    mask01 = (px["mask_video"] >= 0.5).float()
    cos = [enc(px["cos_control_videos"][k]) for k in sorted(px["cos_control_videos"])]
    return dict(control=enc(px["control_video"]), add=torch.cat([enc(px["depth_video"])] + cos, dim=1),
                masked=OV.vae_encode(vsd, (px["video"] * 2 - 1) * (mask01 < 0.5), C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN,
                                     OV.LATENT_STD),
                ref=enc(px["ref_image"])[:, :, 0], mask01=mask01)


def test_encode_conditioning_matches_oracle_streams():
    pipe, cfg, dsd, vsd = make_pipe()
    px = pixel_streams()
    shape = (1, 48, 3, H // 16, W // 16)
    cond = pipe.encode_conditioning(px["video"], px["mask_video"], px["control_video"], px["depth_video"], px["cos_control_videos"],
                                    px["ref_image"], H, W, shape)
    want = oracle_latents(vsd, px)
    for name, got, ref in (("control", cond.control_latents, want["control"]), ("additional", cond.additional_control, want["add"]),
                           ("masked", cond.masked_video_latents, want["masked"]), ("ref", cond.ref_latents, want["ref"])):
        p = C.psnr(got.float().cpu(), ref)
        print(f"{name}: shape {tuple(got.shape)} psnr {p:.1f} dB")
        assert got.shape == ref.shape and p >= 40.0
    assert torch.equal(cond.mask_pixels.cpu(), want["mask01"])
    # all-255 mask: zero mask latents / known latents, mask = 1 (PIPE.py:648-654)
    full = pipe.encode_conditioning(px["video"], torch.full_like(px["mask_video"], 255.0), px["control_video"], None,
                                    px["cos_control_videos"], None, H, W, shape)
    assert float(full.masked_video_latents.abs().max()) == 0 and float(full.mask_latents.abs().max()) == 0
    assert bool((full.mask == 1).all()) and float(full.additional_control[:, :48].abs().max()) == 0
    assert full.ref_latents.shape == (1, 48, H // 16, W // 16) and float(full.ref_latents.abs().max()) == 0


def test_pixel_call_matches_oracle_composition():
    pipe, cfg, dsd, vsd = make_pipe()
    px = pixel_streams()
    g = torch.Generator().manual_seed(62)
    latents = torch.randn(1, 48, 3, H // 16, W // 16, generator=g)
    ctx_c, ctx_u = [torch.randn(7, cfg["text_dim"], generator=g) * 0.1], [torch.randn(3, cfg["text_dim"], generator=g) * 0.1]
    common = dict(prompt_embeds=ctx_c, negative_prompt_embeds=ctx_u, height=H, width=W, num_frames=FRAMES, num_inference_steps=3,
                  guidance_scale=6.0, density=0.1, latents=latents, **px)
    lat = pipe(output_type="latent", **common).videos.float().cpu()
    vid = pipe(**common).videos
    assert vid.shape == (1, 3, FRAMES, H, W) and float(vid.min()) >= 0 and float(vid.max()) <= 1
    want = oracle_latents(vsd, px)
    ml, mask, pinned = S.prepare_masks(want["mask01"], latents)
    ref_lat = S.denoise_loop(lambda **kw: O.dit_forward(dsd, cfg, **kw), S.FlowMatchEulerSchedule(1000, 5.0), 3, latents, ctx_u, ctx_c,
                             want["control"], want["add"], ml, want["masked"], want["ref"], mask, pinned, 0.1, 6.0)
    p_lat = C.psnr(lat, ref_lat)
    ref_vid = OV.vae_decode(vsd, ref_lat, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD) / 2 + 0.5
    p_vid = C.psnr(vid.float(), ref_vid.clamp(0, 1), peak=1.0)
    print(f"pixel call: latents psnr {p_lat:.1f} dB, decoded video psnr {p_vid:.1f} dB")
    assert p_lat >= 40.0 and p_vid >= 40.0                 # north_star: PSNR >= 40 dB vs the reference path
