"""GPU parity of the Wan2.2 3D-VAE decode (HIP path through the C ABI) against golden G7 (REFERENCE
module output, small widths, 3 latent frames: first chunk, "Rep" hand-over, cached chunk) and against
the fp32 oracle on another shape (4 chunks).  Tolerance: bf16 conv operands, fp32 accumulate and fp32
residual stream -> PSNR >= 40 dB on the clamped [-1, 1] video (peak 2), rel-RMS <= 2e-2."""
import pytest
import torch

from oracle import cases as C
from oracle import vae as OV

pytestmark = pytest.mark.gpu


def build(seed=31):
    from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
    v = C.VAE_SMALL
    vae = AutoencoderKLWan3_8(latent_channels=v["z_dim"], dec_dim=v["dec_dim"], dim_mult=list(v["dim_mult"]),
                              temperal_downsample=list(v["temporal_up"])[::-1], spatial_compression_ratio=16)
    sd = C.vae_weights(v, seed=seed, prefix="model.")
    missing, unexpected = vae.load_state_dict(sd, strict=True)
    return vae.to("cuda:0"), sd


def check(got, want, what):
    got, want = got.float().cpu(), want.float()
    rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    p = C.psnr(got, want, peak=2.0)
    print(f"{what}: rel-rms {rel:.3e}, psnr {p:.1f} dB")
    assert p >= 40.0 and rel <= 2e-2, f"{what}: rel-rms {rel:.3e}, psnr {p:.1f} dB"


def test_vae_decode_matches_reference_golden(golden):
    fx = golden("g7_vae_decode")
    vae, sd = build()
    z = C.vae_case(h=4, w=6)
    out = vae.decode(z.cuda()).sample
    assert out.shape == (1, 3, 9, 64, 96)
    check(out, fx["out"], "vae decode g7")
    assert float(out.abs().max()) <= 1.0


def test_vae_decode_four_chunks_other_shape_and_repeatable():
    vae, sd = build(seed=77)
    z = C.vae_case(seed=78, frames=4, h=2, w=4)
    want = OV.vae_decode(sd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    out1 = vae.decode(z.cuda()).sample
    assert out1.shape == (1, 3, 13, 32, 64)
    check(out1, want, "vae decode 4 chunks")
    out2 = vae.decode(z.cuda()).sample                      # history must be reset between calls
    torch.testing.assert_close(out1, out2, rtol=0, atol=0)
