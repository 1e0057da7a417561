"""GPU parity of the sampler hot loop (DiT x CFG pair + fused CFG/Euler/blend in HIP) against golden
G9: per-step latents produced by the REFERENCE DiT module inside the restated loop (BASELINE
config 1: 9x256x256 -> latent [1,48,3,16,16], 4 Euler steps, CFG 6, motion_transfer mask).
Tolerance: PSNR >= 40 dB on every step's latents (north_star), rel-RMS <= 2e-2."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O

pytestmark = pytest.mark.gpu


def make_pipe(cfg, seed):
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM, Wan2_2Transformer3DModel_FlexAM
    kw = dict(cfg)
    kw.pop("eps")
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
    m.load_state_dict(C.dit_weights(cfg, seed), strict=True)
    return Wan2_2FunControlPipeline_FlexAM(transformer=m.to("cuda:0"))


def test_sampler_trace_matches_reference_golden(golden):
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    fx = golden("g9_sampler")
    cfg = dict(O.DIT_TINY)
    pipe = make_pipe(cfg, 7)
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(control_latents=sc["control_latents"], additional_control=sc["additional_control"],
                              masked_video_latents=sc["masked_video_latents"], ref_latents=sc["ref_latents"],
                              mask_pixels=sc["mask_pixels"])
    trace = []
    out = pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
               num_inference_steps=sc["num_steps"], guidance_scale=sc["guidance_scale"], density=sc["density"], latents=sc["latents"],
               conditioning=cond, output_type="latent",
               callback_on_step_end=lambda p, i, t, kw: trace.append(kw["latents"].float().cpu().clone()))
    torch.testing.assert_close(pipe.scheduler.sigmas, fx["sigmas"])
    assert len(trace) == 4
    for i, lat in enumerate(trace):
        want = fx["trace"][i]
        rel = ((lat - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        p = C.psnr(lat, want)
        print(f"step {i}: rel-rms {rel:.3e} psnr {p:.1f} dB")
        assert p >= 40.0 and rel <= 2e-2
        # frame 0 stays pinned to the known latent (PIPE.py:933-934): exact in fp32
        torch.testing.assert_close(lat[:, :, 0], sc["masked_video_latents"][:, :, 0])
    torch.testing.assert_close(out.videos.float().cpu(), trace[-1])


def test_sampler_without_cfg_and_input_checks():
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    cfg = dict(O.DIT_TINY)
    pipe = make_pipe(cfg, 7)
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
    out = pipe(prompt_embeds=sc["context_cond"], height=256, width=256, num_frames=9, num_inference_steps=2, guidance_scale=1.0,
               density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent")
    assert out.videos.shape == (1, 48, 3, 16, 16) and bool(torch.isfinite(out.videos).all())
    with pytest.raises(ValueError):
        pipe(height=250, width=256, prompt_embeds=sc["context_cond"])
    with pytest.raises(ValueError):
        pipe(height=256, width=256)


def test_sampler_cfg_skip_uses_conditional_row_only():
    """cfg_skip (cfg_optimization.py:5-37) in the fused loop: for the last half of the steps only the
    conditional row runs; the result must equal an oracle loop that does v = cond on those steps."""
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from oracle import sampler as S
    cfg = dict(O.DIT_TINY)
    pipe = make_pipe(cfg, 7)
    sd = C.dit_weights(cfg, 7)
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
    pipe.transformer.enable_cfg_skip(0.5, 4)
    out = pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
               num_inference_steps=4, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent")
    pipe.transformer.disable_cfg_skip()
    ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    calls = {"n": 0}

    def model(**kw):
        out_ = O.dit_forward(sd, cfg, **kw)
        step = calls["n"]
        calls["n"] += 1
        return torch.cat([out_[1:], out_[1:]]) if step >= 2 else out_        # duplicated cond row on skipped steps
    ref = S.denoise_loop(model, S.FlowMatchEulerSchedule(1000, 5.0), 4, sc["latents"], sc["context_uncond"], sc["context_cond"],
                         sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"], sc["ref_latents"], mask, pinned,
                         0.1, 6.0)
    p = C.psnr(out.videos.float().cpu(), ref)
    print(f"cfg_skip sampler: psnr {p:.1f} dB")
    assert p >= 40.0


@pytest.mark.parametrize("thresh,want_calc", [(1.5, [True, False, True, False, True, False, True, True]),
                                              (2.5, [True, False, False, True, False, False, True, True])])
def test_sampler_teacache_together_with_cfg_skip(thresh, want_calc):
    """TeaCache + cfg_skip in one run: the reference's forward makes the TeaCache decision and counts on EVERY conditional
    forward, the cfg-skipped B = 1 ones included, and adds `previous_residual[-x.size(0):]` (FX.py:977-1006, 1119-1122); with
    thresh 2.5 step 4 is a skipped B = 1 step that re-uses the conditional row of a B = 2 residual.  The oracle loop below is
    the reference's composition: cfg_skip decorator (cfg_optimization.py:5-37) around dit_forward with its TeaCache state."""
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from oracle import sampler as S
    cfg = dict(O.DIT_TINY)
    pipe = make_pipe(cfg, 7)
    sd = C.dit_weights(cfg, 7)
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
    n = 8
    tr = pipe.transformer
    tr.enable_teacache([1.0, 0.0], n, rel_l1_thresh=thresh, num_skip_start_steps=1)
    tr.enable_cfg_skip(0.5, n)
    seen = []
    out = pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
               num_inference_steps=n, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent",
               callback_on_step_end=lambda p, i, t, kw: seen.append(bool(tr.teacache.should_calc)))
    assert tr.teacache.cnt == 0 and tr.teacache.previous_modulated_input is None       # the reset fired at the clip's end
    assert seen == want_calc
    tr.disable_teacache()
    tr.disable_cfg_skip()
    ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    tc = O.teacache_state([1.0, 0.0], n, thresh, 1)
    calls, dec = {"n": 0}, []

    def model(**kw):
        step = calls["n"]
        calls["n"] += 1
        if step >= n // 2:
            cut = lambda v: v[1:] if isinstance(v, (torch.Tensor, list, tuple)) else v
            o = O.dit_forward(sd, cfg, teacache=tc, **{k: cut(v) for k, v in kw.items()})
            o = torch.cat([o, o])
        else:
            o = O.dit_forward(sd, cfg, teacache=tc, **kw)
        dec.append(bool(tc["should_calc"]))
        return o
    ref = S.denoise_loop(model, S.FlowMatchEulerSchedule(1000, 5.0), n, sc["latents"], sc["context_uncond"], sc["context_cond"],
                         sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"], sc["ref_latents"], mask, pinned,
                         0.1, 6.0)
    assert dec == want_calc
    p = C.psnr(out.videos.float().cpu(), ref)
    print(f"teacache + cfg_skip sampler (thresh {thresh}): psnr {p:.1f} dB")
    assert p >= 40.0


def test_fractional_mask_many_timestep_rows_and_bounded_table():
    """A mask whose frame 0 is not all-known is NOT pinned (PIPE.py:688-690): the trilinear latent mask then has many distinct
    values, i.e. many distinct per-token timesteps (foreground edits with soft edges).  The AdaLN table is then rebuilt per
    layer from one bounded buffer instead of [layers, R, 6, C]; results must not depend on which form is used."""
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from oracle import sampler as S
    cfg = dict(O.DIT_TINY)
    pipe = make_pipe(cfg, 7)
    sd = C.dit_weights(cfg, 7)
    sc = C.sampler_case(cfg)
    g = torch.Generator().manual_seed(77)
    mp = torch.rand(sc["mask_pixels"].shape, generator=g)                          # soft mask values, frame 0 included
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], mp)
    kw = dict(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
              num_inference_steps=2, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent")
    eng = pipe.transformer.engine()
    out_a = pipe(**kw).videos.float().cpu().clone()
    assert pipe._state["U"] > 50                            # many distinct timesteps per sample
    eng.table_limit = 0                                     # force the per-layer table
    out_b = pipe(**kw).videos.float().cpu().clone()
    torch.testing.assert_close(out_a, out_b, rtol=0, atol=0)
    ml, mask, pinned = S.prepare_masks(mp, sc["latents"])
    assert not pinned
    ref = S.denoise_loop(lambda **k: O.dit_forward(sd, cfg, **k), S.FlowMatchEulerSchedule(1000, 5.0), 2, sc["latents"], sc["context_uncond"],
                         sc["context_cond"], sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"],
                         sc["ref_latents"], mask, pinned, 0.1, 6.0)
    p = C.psnr(out_b, ref)
    print(f"fractional mask, U = {pipe._state['U']}: psnr {p:.1f} dB")
    assert p >= 40.0


def test_block0_shared_self_attention_half_equals_per_sample_run(monkeypatch):
    """The sampler's CFG pair is one latent with two prompts: until block 0's cross-attention both samples are the same tensor, so
    the engine runs block 0's LayerNorm / q|k|v / RoPE / self-attention / output projection once and copies the residual stream
    (DiTEngine.run, `share0`).  Same arithmetic per row as the per-sample run (FLEXAM_SHARE_BLOCK0=0): the two must agree to the
    rounding of the split-KV merge (the attention launch splits a different set of work units), and both match the oracle loop.
    A model.forward() call with a batch-2 latent (no shared-latent promise) never takes the shortcut."""
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from oracle import sampler as S
    cfg = dict(O.DIT_TINY, num_layers=3)
    sd = C.dit_weights(cfg, 7)
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
    kw = dict(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
              num_inference_steps=3, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent")
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FLEXAM_SHARE_BLOCK0", flag)
        outs[flag] = make_pipe(cfg, 7)(**kw).videos.float().cpu()
    rel = ((outs["1"] - outs["0"]).pow(2).mean().sqrt() / outs["0"].pow(2).mean().sqrt()).item()
    print(f"shared block-0 half vs per-sample: rel-rms {rel:.2e}")
    assert rel <= 1e-3
    ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    ref = S.denoise_loop(lambda **k: O.dit_forward(sd, cfg, **k), S.FlowMatchEulerSchedule(1000, 5.0), 3, sc["latents"], sc["context_uncond"],
                         sc["context_cond"], sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"],
                         sc["ref_latents"], mask, pinned, 0.1, 6.0)
    for flag, o in outs.items():
        p = C.psnr(o, ref)
        print(f"FLEXAM_SHARE_BLOCK0={flag}: psnr vs oracle {p:.1f} dB")
        assert p >= 40.0
