"""CPU: the oracle (oracle/*.py) against golden vectors produced by the reference modules
(oracle/make_golden.py).  This is what pins the oracle; fp32 vs fp32, tight tolerances."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O
from oracle import sampler as S
from oracle import vae as OV

TOL = dict(rtol=2e-4, atol=2e-5)


def _same_problem(fx, key, tensors):
    torch.testing.assert_close(C.checksum(tensors), fx[key], rtol=1e-9, atol=1e-9,
                               msg="seeded inputs/weights differ from the ones the fixture was generated with")


def test_g1_rope(golden):
    fx = golden("g1_rope")
    g = torch.Generator().manual_seed(101)
    x = C.randn(g, 1, 4 * 8 * 8 + 7, 2, 128)
    _same_problem(fx, "in_sum", dict(x=x))
    out = O.rope_apply(x, (4, 8, 8), O.rope_angles(1024, 128))
    torch.testing.assert_close(out, fx["out"], **TOL)
    torch.testing.assert_close(out[:, -7:], x[:, -7:])          # pass-through tail (FX.py:160)


def test_g2_norms_sinusoid(golden):
    fx = golden("g2_norms")
    g = torch.Generator().manual_seed(102)
    x = C.randn(g, 2, 9, 256) * 3.0
    w = 1.0 + 0.1 * C.randn(g, 256)
    sc, sh, dn = C.randn(g, 2, 9, 256) * 0.3, C.randn(g, 2, 9, 256) * 0.3, C.randn(g, 2, 1, 256) * 0.3
    _same_problem(fx, "in_sum", dict(x=x, w=w, sc=sc, sh=sh, dn=dn))
    torch.testing.assert_close(O.rms_norm(x, w, 1e-6), fx["rms"], **TOL)
    torch.testing.assert_close(O.layer_norm(x, 1e-6) * (1 + sc) + sh + dn, fx["ln_mod"], **TOL)
    sin = O.sinusoidal_embedding_1d(256, torch.tensor([0.0, 24.4, 500.0, 1000.0]))
    torch.testing.assert_close(sin, fx["sinus"], rtol=1e-12, atol=1e-12)            # fp64 like the reference


def test_g3_block(golden):
    fx = golden("g3_block")
    bc = C.block_case()
    bw = C.block_weights(bc["dim"], bc["ffn"])
    _same_problem(fx, "in_sum", {k: v for k, v in bc.items() if torch.is_tensor(v)})
    _same_problem(fx, "w_sum", bw)
    sd = {"b." + k: v for k, v in bw.items()}
    out = O.block_forward(sd, "b", bc["x"], bc["e0"], bc["dens0"], bc["grid"], O.rope_angles(1024, 128), bc["context"], bc["heads"])
    torch.testing.assert_close(out, fx["out"], **TOL)


def _dit(golden, name, per_tok, h=16, w=16):
    fx = golden(name)
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 7)
    _same_problem(fx, "w_sum", sd)
    case = C.dit_case(cfg, 41, per_token_t=per_tok, h=h, w=w)
    flat = {k: v for k, v in case.items() if torch.is_tensor(v)}
    flat.update({f"ctx{i}": u for i, u in enumerate(case["context"])})
    _same_problem(fx, "in_sum", flat)
    out = O.dit_forward(sd, cfg, **case)
    torch.testing.assert_close(out, fx["out"], **TOL)


def test_g4_dit_per_token_t(golden):
    _dit(golden, "g4_dit_tokent", True)


def test_g4b_dit_nonsquare(golden):
    _dit(golden, "g4b_dit_nonsquare", True, 8, 24)


def test_g5_dit_scalar_t(golden):
    _dit(golden, "g5_dit_scalart", False)


def test_g7_vae_decode(golden):
    fx = golden("g7_vae_decode")
    vsd = C.vae_weights(C.VAE_SMALL, prefix="")
    _same_problem(fx, "w_sum", vsd)
    z = C.vae_case(h=4, w=6)
    _same_problem(fx, "in_sum", dict(z=z))
    taps = []
    out = OV.vae_decode(vsd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD, prefix="", taps=taps)
    assert out.shape == (1, 3, 9, 64, 96)
    torch.testing.assert_close(torch.cat([t["middle"] for t in taps], dim=2), fx["middle"], **TOL)
    torch.testing.assert_close(taps[-1]["up1"], fx["up1_last"], **TOL)
    torch.testing.assert_close(out, fx["out"], **TOL)


def test_scheduler_known_answers():
    """Closed-form values of the double-shifted flow-match Euler schedule; the
    scheduler itself is third-party and absent: PARITY UNPINNED (oracle/sampler.py header)."""
    s = S.FlowMatchEulerSchedule(1000, 5.0)
    assert abs(s.sigma_max - 1.0) < 1e-7 and abs(s.sigma_min - 0.0049801) < 1e-6
    s.set_timesteps(50)
    sig = s.sigmas
    assert sig.shape == (51,) and sig[-1] == 0
    ref_head = torch.tensor([1.0, 0.995872, 0.991605])
    ref_tail = torch.tensor([0.192804, 0.114819, 0.024414])
    torch.testing.assert_close(sig[:3], ref_head, rtol=0, atol=2e-6)
    torch.testing.assert_close(sig[47:50], ref_tail, rtol=0, atol=2e-6)
    torch.testing.assert_close(s.timesteps, sig[:-1] * 1000)
    x = torch.ones(3, dtype=torch.bfloat16)
    v = torch.full((3,), 2.0, dtype=torch.bfloat16)
    out = s.step(v, x)
    assert out.dtype == torch.bfloat16
    torch.testing.assert_close(out.float(), (1.0 + (sig[1] - sig[0]) * 2.0).to(torch.bfloat16).float().expand(3))


def test_g9_sampler_trace(golden):
    """BASELINE config 1 plumbing: 4 Euler steps, CFG 6, latent [1,48,3,16,16], oracle DiT inside
    the restated loop vs the reference DiT module inside the same loop."""
    fx = golden("g9_sampler")
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 7)
    _same_problem(fx, "w_sum", sd)
    sc = C.sampler_case(cfg)
    mask_latents, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    assert pinned and float(mask[:, :, 0].abs().max()) == 0 and float(mask[:, :, 1:].min()) == 1
    torch.testing.assert_close(mask_latents, fx["mask_latents"])
    sched = S.FlowMatchEulerSchedule(1000, 5.0)
    trace = []
    model = lambda **kw: O.dit_forward(sd, cfg, **kw)
    final = S.denoise_loop(model, sched, sc["num_steps"], sc["latents"], sc["context_uncond"], sc["context_cond"],
                           sc["control_latents"], sc["additional_control"], mask_latents, sc["masked_video_latents"],
                           sc["ref_latents"], mask, pinned, sc["density"], sc["guidance_scale"], trace=trace)
    torch.testing.assert_close(torch.stack(trace), fx["trace"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(final, fx["final"], rtol=1e-3, atol=1e-4)
    # frame 0 stays pinned to the known latent after every step (PIPE.py:933-934)
    torch.testing.assert_close(final[:, :, 0], sc["masked_video_latents"][:, :, 0])


def test_g6_teacache(golden):
    """TeaCache (FX.py:977-1051): decisions and outputs of computed AND skipped steps vs the reference."""
    fx = golden("g6_teacache")
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 7)
    _same_problem(fx, "w_sum", sd)
    tcase = C.TEACACHE_CASE
    tc = O.teacache_state(tcase["coefficients"], tcase["num_steps"], tcase["thresh"], tcase["skip_start"])
    for i, tv in enumerate(tcase["t_values"]):
        out = O.dit_forward(sd, cfg, **C.dit_case(cfg, 41, per_token_t=True, t_value=tv), teacache=tc)
        assert float(tc["should_calc"]) == float(fx["should_calc"][i]) or tc["cnt"] == 0
        torch.testing.assert_close(out, fx["outs"][i], **TOL)
    assert tc["cnt"] == 0                                     # reset after num_steps calls (FX.py:1121-1122)


def test_g8_vae_encode(golden):
    fx = golden("g8_vae_encode")
    esd = C.vae_enc_weights(C.VAE_ENC_SMALL, prefix="")
    xv, xi = C.vae_enc_case(), C.vae_enc_case(seed=43, frames=1, h=32, w=32)
    assert torch.equal(C.checksum(esd), fx["w_sum"]) and torch.equal(C.checksum(dict(v=xv, i=xi)), fx["in_sum"])
    for x, key in ((xv, "mu_video"), (xi, "mu_image")):
        mu = OV.vae_encode(esd, x, C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN, OV.LATENT_STD, prefix="")
        assert mu.shape == fx[key].shape
        assert (mu - fx[key]).abs().max().item() < 2e-5 * max(1.0, fx[key].abs().max().item())


@pytest.mark.parametrize("name", sorted(C.SOLVER_CASES))
def test_g10_multistep_solvers(golden, name):
    from oracle import solvers as SV
    fx = golden("g10_solver_" + name)
    kind, steps, shift, kw, x, vs = C.solver_case(name)
    assert torch.equal(C.checksum(dict(x=x, **{f"v{i}": v for i, v in enumerate(vs)})), fx["in_sum"])
    if kind == "unipc":
        sig, ts = SV.flow_sigmas(steps, shift)
        sol = SV.UniPC(sig, **kw)
    else:
        sig, ts = SV.flow_sigmas(steps, 1.0, sigmas=SV.sampling_sigmas(steps, shift))
        sol = SV.DPMSolverPP(sig, **kw)
    assert torch.equal(sig, fx["sigmas"]) and torch.equal(ts, fx["timesteps"])
    cur = x.clone()
    gen = torch.Generator().manual_seed(C.SOLVER_NOISE_SEED)
    for i in range(steps):
        cur = sol.step(vs[i], cur, generator=gen) if kind == "dpm" else sol.step(vs[i], cur)
        want = fx["trace"][i]
        assert (cur - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), (name, i)


def test_g12_t5_encoder(golden):
    from oracle import t5 as OT
    fx = golden("g12_t5")
    cfg = dict(OT.T5_TINY)
    sd = OT.seeded_t5_weights(cfg, 5)
    assert torch.equal(C.checksum(sd), fx["w_sum"])
    ids, mask = OT.t5_case(cfg)
    assert torch.equal(ids.float(), fx["ids"]) and torch.equal(mask.float(), fx["mask"])
    out = OT.t5_encode(sd, cfg, ids, mask)
    torch.testing.assert_close(out, fx["out"], rtol=1e-5, atol=1e-5)


def test_g11b_dit_with_riflex(golden):
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 7)
    case = C.dit_case(cfg, 41)
    out = O.dit_forward(sd, cfg, riflex=(2, 3, 1.0), **case)
    torch.testing.assert_close(out, golden("g11b_dit_riflex")["out"], **TOL)
    assert not torch.allclose(out, golden("g4_dit_tokent")["out"], atol=1e-3)          # the switch does change the output
