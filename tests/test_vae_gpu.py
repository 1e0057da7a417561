"""GPU parity of the Wan2.2 3D-VAE decode (HIP path through the C ABI) against golden G7 (REFERENCE
module output, small widths, 3 latent frames: first chunk, "Rep" hand-over, cached chunk) and against
the fp32 oracle on another shape (4 chunks).  Tolerance: bf16 conv operands, fp32 accumulate and fp32
residual stream -> PSNR >= 40 dB on the clamped [-1, 1] video (peak 2), rel-RMS <= 2e-2."""
import math

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
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("model.encoder.", "model.conv1.")) for k in missing)
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


# ----------------------------------------------------------------------------- encode
def build_encoder(seed=41):
    from flexam_amd.wan_vae3_8 import AutoencoderKLWan3_8
    v = C.VAE_ENC_SMALL
    vae = AutoencoderKLWan3_8(latent_channels=v["z_dim"], c_dim=v["dim"], dec_dim=16, dim_mult=list(v["dim_mult"]),
                              temperal_downsample=list(v["temporal_down"]), spatial_compression_ratio=16)
    sd = C.vae_enc_weights(v, seed=seed, prefix="model.")
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("model.decoder.", "model.conv2.")) for k in missing)
    return vae.to("cuda:0"), sd


def check_latent(got, want, what):
    got, want = got.float().cpu(), want.float()
    rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    p = C.psnr(got, want)
    print(f"{what}: rel-rms {rel:.3e}, psnr {p:.1f} dB")
    assert p >= 40.0 and rel <= 2e-2, f"{what}: rel-rms {rel:.3e}, psnr {p:.1f} dB"


def test_vae_encode_matches_reference_golden(golden):
    """G8: REFERENCE AutoencoderKLWan2_2_.encode outputs (normalised mu) on a 9-frame clip (chunks 1+4+4:
    first-chunk temporal-conv skip, cached strided temporal conv, AvgDown3D front pad) and a single image."""
    fx = golden("g8_vae_encode")
    vae, sd = build_encoder()
    xv, xi = C.vae_enc_case(), C.vae_enc_case(seed=43, frames=1, h=32, w=32)
    mu_v = vae.encode(xv.cuda()).latent_dist.mode()
    assert mu_v.shape == (1, 48, 3, 2, 4)
    check_latent(mu_v, fx["mu_video"], "vae encode g8 video")
    mu_i = vae.encode(xi.cuda())[0].mode()
    assert mu_i.shape == (1, 48, 1, 2, 2)
    check_latent(mu_i, fx["mu_image"], "vae encode g8 image")


def test_vae_encode_five_chunks_other_shape_repeatable_and_logvar():
    vae, sd = build_encoder(seed=91)
    x = C.vae_enc_case(seed=92, frames=17, h=48, w=32)
    want_mu = OV.vae_encode(sd, x, C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN, OV.LATENT_STD)
    post = vae.encode(x.cuda(), return_dict=False)[0]
    assert post.parameters.shape == (1, 96, 5, 3, 2)
    check_latent(post.mode(), want_mu, "vae encode 5 chunks")
    again = vae.encode(x.cuda()).latent_dist.parameters        # caches must be reset between calls
    torch.testing.assert_close(post.parameters, again, rtol=0, atol=0)
    assert torch.isfinite(post.sample(generator=torch.Generator("cuda").manual_seed(0))).all()


def test_vae_tiled_decode_is_exact_at_half_the_clip_size():
    """The 2 x 4 grid of the clip itself on a [16, 28] latent.  At this size the tiles' GEMMs and the full frames' get different tail
    split-K plans -- another fp32 summation order, the only difference between them: in a child process with FLEXAM_GEMM_SPLITK=0 (the
    switch is read once per process) the tiles are bit-identical to the full decode; with the default plans they agree to > 50 dB."""
    import os, subprocess, sys
    env = dict(os.environ, FLEXAM_GEMM_SPLITK="0")
    code = ("import sys; sys.path.insert(0, %r); import test_vae_gpu as T; T.test_vae_tiled_decode_is_exact(8, 16, 28); print('exact')"
            % os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "exact" in r.stdout, r.stderr[-2000:]
    vae, sd = build(seed=55)
    z = C.vae_case(seed=56, frames=3, h=16, w=28)
    full = vae.decode(z.cuda()).sample[0]
    eng = vae.engine()
    got = eng.assemble_tiles([eng.decode(z[0].cuda(), stripe=(r_, 8)) for r_ in range(8)], eng.band_grid(16, 28, 8))
    mse = float((got - full).double().pow(2).mean())
    assert mse == 0 or 10 * math.log10(4.0 / mse) > 50.0


@pytest.mark.parametrize("world,h,w", [(2, 8, 4), (4, 8, 4), (8, 8, 4), (8, 32, 4), (4, 20, 4), (4, 8, 14), (6, 12, 6)])
def test_vae_tiled_decode_is_exact(world, h, w):
    """Parallel decode: each rank decodes one tile of a rows x columns grid with a receptive-field halo, re-cropped at every
    stage (r6); the tiles of all ranks, computed here one after another on one GPU, must tile the full decode BIT-EXACTLY.  Narrow latents
    (w = 4) get row bands -- h = 32 is the 97 x 512 x 896 clip's latent height: the row pattern of every rank of eight at the real size --,
    wide ones a 2-D grid (16 x 28 is the clip's latent at half size: 2 x 4 tiles on 8 ranks, like the clip itself)."""
    vae, sd = build(seed=55)
    z = C.vae_case(seed=56, frames=3, h=h, w=w)
    full = vae.decode(z.cuda()).sample[0]
    eng = vae.engine()
    gr, gc = eng.band_grid(h, w, world)
    assert gr * gc == world
    tiles = [eng.decode(z[0].cuda(), stripe=(r, world)) for r in range(world)]
    assert all(tl.shape == (3, 9, 16 * h // gr, 16 * w // gc) for tl in tiles)
    torch.testing.assert_close(eng.assemble_tiles(tiles, (gr, gc)), full, rtol=0, atol=0)
    plans = [eng.stripe_plan(h, w, r, world) for r in range(world)]
    for crops, (lo, hi, clo, chi), grid in plans:
        assert grid == (gr, gc) and hi - lo == 16 * h // gr and chi - clo == 16 * w // gc
        assert all(0 <= a < b and 0 <= ca < cb for a, b, ca, cb in crops.values())
    if w == 4:
        assert gc == 1                                               # nothing to win from cutting 64 columns
        if world > 2:                                                # the last stage works on less than the frame's 8 h rows
            assert all(max(crops) == 3 and crops[3][1] - crops[3][0] < 8 * h for crops, _, _ in plans)
        if h == 32:
            assert {k: v[:2] for k, v in plans[4][0].items()} == {1: (20, 52), 2: (14, 50), 3: (13, 59)} and 0 in plans[0][0] and 0 in plans[7][0]
    if (world, h, w) == (8, 16, 28):
        assert (gr, gc) == (2, 4)


def test_vae_tile_grid_of_the_clip():
    """The grid the 97 x 512 x 896 clip gets (latent 32 x 56): 2 x 4 tiles on 8 ranks, 2 x 2 on 4, two row bands on 2 (host logic of the real
    decoder widths; no decode)."""
    from flexam_amd import AutoencoderKLWan3_8
    with torch.device("cuda"):
        vae = AutoencoderKLWan3_8(spatial_compression_ratio=16).to(torch.bfloat16)
    eng = vae.engine()
    assert eng.band_grid(32, 56, 8) == (2, 4) and eng.band_grid(32, 56, 4) == (2, 2) and eng.band_grid(32, 56, 2)[0] * eng.band_grid(32, 56, 2)[1] == 2
    crops, (lo, hi, clo, chi), grid = eng.stripe_plan(32, 56, 5, 8)            # second row of tiles, second column
    assert (hi - lo, chi - clo) == (256, 224) and max(crops) == 3


@pytest.mark.parametrize("frames", [9, 12])
def test_vae_decode_ring_wraps_are_consumed(frames, monkeypatch):
    """The causal convs keep their input frames in a ring of 4 chunks; the wrap copy fires after chunk 3, 7, ... and the chunk
    AFTER a wrap reads the copied history.  9 latent frames = chunks 0..8 (two wraps consumed, first chunk of 1 frame so the
    window start is de-aligned like the 25-chunk config-2 decode); 12 = three wraps.  Against the fp32 oracle, and the ring
    must not change a bit: FLEXAM_VAE_RING = 1 (copy after every chunk, the r1 form) decodes the identical video."""
    import flexam_amd.wan_vae3_8 as V
    vae, sd = build(seed=83)
    z = C.vae_case(seed=84, frames=frames, h=2, w=4)
    want = OV.vae_decode(sd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    assert V._Conv.RING == 4
    out = vae.decode(z.cuda()).sample
    assert out.shape == (1, 3, 4 * (frames - 1) + 1, 32, 64)
    check(out, want, f"vae decode {frames} chunks (ring of 4)")
    again = vae.decode(z.cuda()).sample                       # second call starts from a reset ring
    torch.testing.assert_close(out, again, rtol=0, atol=0)
    monkeypatch.setattr(V._Conv, "RING", 1)
    vae1, _ = build(seed=83)
    out1 = vae1.decode(z.cuda()).sample
    torch.testing.assert_close(out, out1, rtol=0, atol=0)
    # last frames alone: an error in a consumed wrap would show there even if the clip-level PSNR hid it
    check(out[:, :, -8:], want[:, :, -8:], "last two chunks")


# ----------------------------------------------------------------------------- r5: chunk length is a schedule, not a result
def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def test_vae_decode_does_not_depend_on_the_chunk_length(monkeypatch):
    """The reference decodes ONE latent frame per chunk (VAE.py:1046-1052); the engine's default is two (FLEXAM_VAE_DEC_CHUNK), any length
    gives the same causal convolutions.  8 latent frames: chunk lengths 1 (the reference's walk), 2 (1 + 2 + 2 + 2 + 1: a short last
    chunk), 3 and 7 (everything after frame 0 at once) against the fp32 oracle and against each other (different GEMM shapes -> different
    split-K sums: fp32 rounding only)."""
    z = C.vae_case(seed=86, frames=8, h=2, w=4)
    outs = {}
    for n in (1, 2, 3, 7):
        monkeypatch.setenv("FLEXAM_VAE_DEC_CHUNK", str(n))
        vae, sd = build(seed=85)
        assert vae.engine().chunk == n
        outs[n] = vae.decode(z.cuda()).sample
        assert outs[n].shape == (1, 3, 29, 32, 64)
    want = OV.vae_decode(sd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    for n, o in outs.items():
        check(o, want, f"vae decode, {n} latent frame(s) per chunk")
        r = _rel(o, outs[1])
        print(f"chunk {n} vs chunk 1: rel-rms {r:.2e}")
        assert r <= 2e-3


def test_vae_encode_does_not_depend_on_the_chunk_length(monkeypatch):
    """Encoder: the reference walks 1 + 4 + 4 + ... frames (VAE.py:1029-1037); default here 1 + 24 + ... (FLEXAM_VAE_ENC_CHUNK).  25 frames:
    chunk lengths 4, 8 (1 + 8 + 8 + 8), 16 (1 + 16 + 8) and 24 against the oracle and each other; trailing frames past 1 + 4k are dropped
    like the reference's `iter_ = 1 + (t - 1) // 4`."""
    x = C.vae_enc_case(seed=94, frames=25, h=32, w=32)
    outs = {}
    for n in (4, 8, 16, 24):
        monkeypatch.setenv("FLEXAM_VAE_ENC_CHUNK", str(n))
        vae, sd = build_encoder(seed=93)
        assert vae.encoder_engine().chunk == n
        outs[n] = vae.encode(x.cuda()).latent_dist.mode()
        assert outs[n].shape == (1, 48, 7, 2, 2)
    want = OV.vae_encode(sd, x, C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN, OV.LATENT_STD)
    for n, o in outs.items():
        check_latent(o, want, f"vae encode, {n} frames per chunk")
        r = _rel(o, outs[4])
        print(f"chunk {n} vs chunk 4: rel-rms {r:.2e}")
        assert r <= 2e-3
    x27 = torch.cat([x, x[:, :, -2:]], dim=2)                 # 27 = 1 + 4 * 6 + 2: the last two frames are not encoded
    torch.testing.assert_close(vae.encode(x27.cuda()).latent_dist.mode(), outs[24], rtol=0, atol=0)


def test_phase_decomposed_upsample_convolution_equals_the_upsampled_image_form(monkeypatch):
    """Resample upsample2d / upsample3d (VAE.py:76-99,153-160): "nearest-exact 2x upsample, then Conv2d 3x3" runs as four 2x2 phase
    convolutions of the LOW-resolution frames with pre-summed taps (_ConvUp2x: 16 instead of 36 tap products per low-resolution pixel)
    and flexam_phase_dupup_cl interleaves the phases into the residual stream.  FLEXAM_VAE_UPCONV=image keeps the earlier form (upsampled
    image + one 3x3 convolution): same video up to the bf16 rounding of the summed taps; both against the fp32 oracle, 4 latent frames
    (first-chunk branch without the time convolution, de-interleaved temporal upsample after it)."""
    z = C.vae_case(seed=88, frames=4, h=4, w=6)
    outs = {}
    for form in ("phase", "image"):
        monkeypatch.setenv("FLEXAM_VAE_UPCONV", form)
        vae, sd = build(seed=87)
        assert vae.engine().phase_up == (form == "phase")
        outs[form] = vae.decode(z.cuda()).sample
    want = OV.vae_decode(sd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    for form, o in outs.items():
        check(o, want, f"vae decode, upsample convolutions in the {form} form")
    r = _rel(outs["phase"], outs["image"])
    print(f"phase form vs image form: rel-rms {r:.2e}")
    assert r <= 1.2e-2          # measured 7.1e-3: each form is ~1e-2 from the oracle (bf16 operands), their roundings are independent


def test_decoder_head_as_per_tap_products_and_gather_equals_the_implicit_gemm(monkeypatch):
    """The decoder head (3x3x3 causal convolution to 12 channels, VAE.py:668-672) runs as ONE plain GEMM over the input pixels (per-tap
    products, N = 27 x 12) + flexam_tapsum_cl (_ConvFold); FLEXAM_VAE_HEADCONV=implicit keeps the implicit GEMM.  Same operands, fp32
    sums in another order: the two decodes agree to fp32 rounding; 5 latent frames so that history frames carry over chunks."""
    z = C.vae_case(seed=90, frames=5, h=4, w=4)
    outs = {}
    for form in ("fold", "implicit"):
        monkeypatch.setenv("FLEXAM_VAE_HEADCONV", form)
        vae, sd = build(seed=89)
        assert type(vae.engine().head_conv).__name__ == ("_ConvFold" if form == "fold" else "_Conv")
        outs[form] = vae.decode(z.cuda()).sample
    want = OV.vae_decode(sd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD)
    check(outs["fold"], want, "vae decode, head convolution as per-tap products + gather")
    r = _rel(outs["fold"], outs["implicit"])
    print(f"fold vs implicit head: rel-rms {r:.2e}")
    assert r <= 1e-4
