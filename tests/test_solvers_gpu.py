"""GPU: the vendored multistep samplers on the HIP path.

  * `FlowUniPCMultistepScheduler` / `FlowDPMSolverMultistepScheduler` step() traces against golden G10 (outputs of the
    REFERENCE classes on seeded model outputs): fp32 linear combinations, tolerance 2e-5 of the trace's range;
  * the sampler loop with each of them (tiny DiT) against the fp32 oracle loop: PSNR >= 40 dB on the final latents;
  * the three element-wise kernels they add (cfg_velocity, lincomb, mask_blend) against torch formulas."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O
from oracle import sampler as S
from oracle import solvers as SV

pytestmark = pytest.mark.gpu


def make_scheduler(kind, kw):
    from flexam_amd import FlowDPMSolverMultistepScheduler, FlowUniPCMultistepScheduler
    cls = FlowUniPCMultistepScheduler if kind == "unipc" else FlowDPMSolverMultistepScheduler
    return cls(num_train_timesteps=1000, shift=1, **kw)


@pytest.mark.parametrize("name", sorted(C.SOLVER_CASES))
def test_solver_traces_match_reference_golden(golden, name):
    from flexam_amd.fm_solvers import get_sampling_sigmas, retrieve_timesteps
    fx = golden("g10_solver_" + name)
    kind, steps, shift, kw, x, vs = C.solver_case(name)
    sch = make_scheduler(kind, kw)
    if kind == "unipc":
        sch.set_timesteps(steps, device="cuda:0", shift=shift)
    else:
        retrieve_timesteps(sch, device="cuda:0", sigmas=get_sampling_sigmas(steps, shift))
    assert torch.equal(sch.sigmas.cpu(), fx["sigmas"]) and torch.equal(sch.timesteps.cpu(), fx["timesteps"])
    cur = x.cuda()
    gen = torch.Generator().manual_seed(C.SOLVER_NOISE_SEED)             # the SDE cases draw their noise from it, on the CPU like the reference
    for i, t in enumerate(sch.timesteps):
        cur = sch.step(vs[i].cuda(), t, cur, generator=gen, return_dict=False)[0]
        want = fx["trace"][i]
        err = (cur.cpu() - want).abs().max().item()
        assert err <= 2e-5 * max(1.0, want.abs().max().item()), (name, i, err)
    assert sch.step_index == steps


def test_solver_kernels_match_torch():
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(8)
    c, f, h, w = 48, 3, 4, 6
    L, ref = f * (h // 2) * (w // 2), 6
    tu, tc = torch.randn(L + ref, 4 * c, generator=g), torch.randn(L + ref, 4 * c, generator=g)
    v = H.cfg_velocity(tu.cuda(), tc.cuda(), ref, 6.0, torch.empty(c, f, h, w, device="cuda:0"))
    comb = tu + 6.0 * (tc - tu)
    want = O.unpatchify(comb[ref:], (f, h // 2, w // 2), (1, 2, 2), c)
    torch.testing.assert_close(v.cpu(), want, rtol=1e-6, atol=1e-6)
    v1 = H.cfg_velocity(tu.cuda(), None, ref, 6.0, torch.empty(c, f, h, w, device="cuda:0"))
    torch.testing.assert_close(v1.cpu(), O.unpatchify(tu[ref:], (f, h // 2, w // 2), (1, 2, 2), c), rtol=0, atol=0)
    ts = [torch.randn(c, f, h, w, generator=g) for _ in range(5)]
    cs = [0.3, -1.7, 2.5, 0.01, -0.4]
    dts = [t.cuda() for t in ts]
    out = H.lincomb(torch.empty_like(dts[0]), list(zip(cs, dts)))
    torch.testing.assert_close(out.cpu(), sum(ci * ti for ci, ti in zip(cs, ts)), rtol=1e-5, atol=1e-6)
    H.lincomb(dts[0], [(2.0, dts[0]), (1.0, dts[1])])                      # in place
    torch.testing.assert_close(dts[0].cpu(), 2.0 * ts[0] + ts[1], rtol=1e-6, atol=1e-6)
    mask = (torch.rand(f, h, w, generator=g) > 0.5).float()
    x, known = ts[2].clone(), ts[3]
    got = H.mask_blend(x.cuda(), known.cuda(), mask.cuda())
    torch.testing.assert_close(got.cpu(), (1 - mask) * known + mask * ts[2], rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        H.lincomb(out, [(1.0, dts[0])] * 9)


@pytest.mark.parametrize("kind,kw", [("unipc", {}), ("dpm", {}), ("dpm", dict(solver_order=3)), ("dpm", dict(algorithm_type="sde-dpmsolver++")),
                                     ("dpm", dict(thresholding=True, sample_max_value=2.0))])
def test_sampler_loop_with_multistep_solver_matches_oracle(kind, kw):
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM, Wan2_2Transformer3DModel_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    cfg = dict(O.DIT_TINY)
    mk = dict(cfg)
    mk.pop("eps")
    m = Wan2_2Transformer3DModel_FlexAM(**mk)
    sd = C.dit_weights(cfg, 7)
    m.load_state_dict(sd, strict=True)
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=m.to("cuda:0"), scheduler=make_scheduler(kind, kw))
    sc = C.sampler_case(cfg)
    cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
    steps = 5
    out = pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
               num_inference_steps=steps, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent",
               shift=5, generator=torch.Generator().manual_seed(C.SOLVER_NOISE_SEED)).videos.float().cpu()
    ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
    ref = S.denoise_loop(lambda **k: O.dit_forward(sd, cfg, **k), SV.MultistepSchedule(kind, 5.0, generator=torch.Generator().manual_seed(C.SOLVER_NOISE_SEED), **kw), steps, sc["latents"],
                         sc["context_uncond"], sc["context_cond"], sc["control_latents"], sc["additional_control"], ml,
                         sc["masked_video_latents"], sc["ref_latents"], mask, pinned, 0.1, 6.0)
    p = C.psnr(out, ref)
    print(f"{kind} {kw}: final latents psnr {p:.1f} dB")
    assert p >= 40.0
    torch.testing.assert_close(out[:, :, 0], sc["masked_video_latents"][:, :, 0])      # frame 0 stays pinned
