"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/*.safetensors from the REFERENCE modules.

Run where /root/reference is mounted:   python -m oracle.make_golden
Each fixture holds reference outputs for a seeded case of oracle/cases.py plus checksums of
the regenerated inputs / weights.  Only data is written: tensors produced by running the
reference classes; no reference source text.

Fixtures:
  g1_rope            rope_apply on [1,L,2,128], grid (4,8,8) + 7 pass-through tail tokens     FX.py:137-164
  g2_norms           WanRMSNorm, WanLayerNorm + modulate, sinusoidal_embedding_1d             FX.py:31-41,173-202
  g3_block           one WanAttentionBlock, dim 256, L 256, per-token e (two distinct rows)    FX.py:422-472
  g4_dit_tokent      tiny Wan2_2Transformer3DModel_FlexAM, per-token t, all FlexAM inputs      FX.py:817-1123
  g4b_dit_nonsquare  same as g4 on a non-square latent [2,48,3,8,24] (catches h/w swaps)
  g5_dit_scalart     same, 1-D t branch                                                         FX.py:941-944
  g6_teacache        6 forwards with TeaCache on (identity rescale, threshold 2.0): computed and skipped steps  FX.py:977-1051
  g8_vae_encode      small AutoencoderKLWan2_2_ encode [1,3,9,32,64] / [1,3,1,32,32] -> normalised mu          VAE.py:788-818
  g10_solver_*       FlowUniPCMultistepScheduler / FlowDPMSolverMultistepScheduler step() traces             fm_solvers_unipc.py:640-724, fm_solvers.py:706-798
  g11_riflex         WanTransformer3DModel_FlexAM.enable_riflex() rope table (cos / sin of the complex freqs)              FX.py:57-113,774-788
  g12_t5             WanT5EncoderModel.forward (tiny umT5 config, per-layer relative position bias, key mask)            wan_text_encoder.py:256-305
  g7_vae_decode      small AutoencoderKLWan2_2_ decode [1,48,3,4,6] -> [1,3,9,64,96] + taps    VAE.py:820-849
  g9_sampler         4-step CFG/Euler/blend trace at latent [1,48,3,16,16] driving the
                     reference DiT module through oracle.sampler.denoise_loop                  PIPE.py:840-949
"""
import os
import sys
import warnings

import torch
from safetensors.torch import save_file

from . import cases as C
from . import dit as O
from . import ref_import
from . import sampler as S
from . import vae as OV

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _save(name, tensors):
    tensors = {k: v.detach().contiguous().to(torch.float64 if v.dtype == torch.float64 else torch.float32)
               for k, v in tensors.items()}
    save_file(tensors, os.path.join(OUT, name + ".safetensors"))
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in tensors.items()))


def _ref_dit(ref, cfg, sd):
    m = ref.dit.Wan2_2Transformer3DModel_FlexAM(
        model_type="ti2v", patch_size=cfg["patch_size"], text_len=cfg["text_len"], in_dim=cfg["in_dim"], dim=cfg["dim"],
        ffn_dim=cfg["ffn_dim"], freq_dim=cfg["freq_dim"], text_dim=cfg["text_dim"], out_dim=cfg["out_dim"],
        num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], add_ref_conv=True, in_dim_ref_conv=cfg["in_dim_ref_conv"],
        add_cnn_block=True, in_dim_cnn_block=cfg["in_dim_cnn_block"], out_dim_cnn_block=cfg["out_dim_cnn_block"]).eval()
    rsd = m.state_dict()
    shapes = O.dit_param_shapes(cfg)
    assert set(rsd) == set(shapes), sorted(set(rsd) ^ set(shapes))
    for k, v in shapes.items():
        assert tuple(rsd[k].shape) == tuple(v), (k, tuple(rsd[k].shape), v)
    m.load_state_dict(sd)
    return m


@torch.no_grad()
def main():
    warnings.filterwarnings("ignore")
    os.makedirs(OUT, exist_ok=True)
    ref = ref_import.load_reference()
    D = ref.dit

    # ---- G1 rope
    g = torch.Generator().manual_seed(101)
    grid = (4, 8, 8)
    l = grid[0] * grid[1] * grid[2] + 7
    x = C.randn(g, 1, l, 2, 128)
    freqs = torch.cat([D.rope_params(1024, 128 - 4 * (128 // 6)), D.rope_params(1024, 2 * (128 // 6)),
                       D.rope_params(1024, 2 * (128 // 6))], dim=1)
    out = D.rope_apply(x, torch.tensor([grid]), freqs)
    _save("g1_rope", dict(out=out, in_sum=C.checksum(dict(x=x))))

    # ---- G2 norms + sinusoid
    g = torch.Generator().manual_seed(102)
    x = C.randn(g, 2, 9, 256) * 3.0
    w = 1.0 + 0.1 * C.randn(g, 256)
    rn = D.WanRMSNorm(256, eps=1e-6)
    rn.weight.data.copy_(w)
    ln = D.WanLayerNorm(256, eps=1e-6)
    sc, sh, dn = C.randn(g, 2, 9, 256) * 0.3, C.randn(g, 2, 9, 256) * 0.3, C.randn(g, 2, 1, 256) * 0.3
    tvals = torch.tensor([0.0, 24.4, 500.0, 1000.0])
    _save("g2_norms", dict(rms=rn(x), ln_mod=ln(x) * (1 + sc) + sh + dn,
                           sinus=D.sinusoidal_embedding_1d(256, tvals),
                           in_sum=C.checksum(dict(x=x, w=w, sc=sc, sh=sh, dn=dn))))

    # ---- G3 block
    bc = C.block_case()
    bw = C.block_weights(bc["dim"], bc["ffn"])
    blk = D.WanAttentionBlock("cross_attn", bc["dim"], bc["ffn"], bc["heads"], cross_attn_norm=True).eval()
    assert set(blk.state_dict()) == set(bw)
    blk.load_state_dict(bw)
    l = bc["x"].shape[1]
    y = blk(bc["x"], bc["e0"], bc["dens0"], torch.tensor([l, l]), torch.tensor([bc["grid"]] * 2), freqs, bc["context"], None,
            dtype=torch.float32)
    _save("g3_block", dict(out=y, in_sum=C.checksum({k: v for k, v in bc.items() if torch.is_tensor(v)}),
                           w_sum=C.checksum(bw)))

    # ---- G4 / G5 tiny DiT
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 7)
    m = _ref_dit(ref, cfg, sd)
    for name, per_tok, hw in (("g4_dit_tokent", True, (16, 16)), ("g5_dit_scalart", False, (16, 16)),
                              ("g4b_dit_nonsquare", True, (8, 24))):
        case = C.dit_case(cfg, 41, per_token_t=per_tok, h=hw[0], w=hw[1])
        out = m(**case)
        flat = {k: v for k, v in case.items() if torch.is_tensor(v)}
        flat.update({f"ctx{i}": u for i, u in enumerate(case["context"])})
        _save(name, dict(out=out, in_sum=C.checksum(flat), w_sum=C.checksum(sd)))

    # ---- G6 TeaCache: 6 calls, identity rescale, threshold 2.0 -> calc / skip / skip / calc / skip / calc
    m.enable_teacache(C.TEACACHE_CASE["coefficients"], C.TEACACHE_CASE["num_steps"], rel_l1_thresh=C.TEACACHE_CASE["thresh"],
                      num_skip_start_steps=C.TEACACHE_CASE["skip_start"], offload=False)
    outs, calcs = [], []
    for tv in C.TEACACHE_CASE["t_values"]:
        outs.append(m(**C.dit_case(cfg, 41, per_token_t=True, t_value=tv)))
        calcs.append(float(m.should_calc))
    m.disable_teacache()
    assert calcs == [1.0, 0.0, 0.0, 1.0, 0.0, 1.0], calcs
    _save("g6_teacache", dict(outs=torch.stack(outs), should_calc=torch.tensor(calcs), w_sum=C.checksum(sd)))

    # ---- G9 sampler trace (reference DiT module inside the restated loop)
    sc_ = C.sampler_case(cfg)
    mask_latents, mask, pinned = S.prepare_masks(sc_["mask_pixels"], sc_["latents"])
    assert pinned
    trace = []
    sched = S.FlowMatchEulerSchedule(1000, 5.0)
    final = S.denoise_loop(lambda **kw: m(**kw), sched, sc_["num_steps"], sc_["latents"], sc_["context_uncond"],
                           sc_["context_cond"], sc_["control_latents"], sc_["additional_control"], mask_latents,
                           sc_["masked_video_latents"], sc_["ref_latents"], mask, pinned, sc_["density"],
                           sc_["guidance_scale"], trace=trace)
    _save("g9_sampler", dict(trace=torch.stack(trace), final=final, mask_latents=mask_latents, mask=mask,
                             sigmas=sched.sigmas, timesteps=sched.timesteps, w_sum=C.checksum(sd)))

    # ---- G7 / G10 VAE decode (small widths; same code path as the 5B VAE)
    V = ref.vae.AutoencoderKLWan2_2_(dim=32, dec_dim=C.VAE_SMALL["dec_dim"], z_dim=48,
                                     temperal_downsample=[False, True, True]).eval()
    vsd = C.vae_weights(C.VAE_SMALL, prefix="")
    ref_keys = {k for k in V.state_dict() if k.startswith("decoder.") or k.startswith("conv2.")}
    assert ref_keys == set(vsd), sorted(ref_keys ^ set(vsd))
    missing, unexpected = V.load_state_dict(vsd, strict=False)
    assert not unexpected
    z = C.vae_case(h=4, w=6)
    mean, std = torch.tensor(OV.LATENT_MEAN), torch.tensor(OV.LATENT_STD)
    taps = {"middle": [], "up1": []}
    hooks = [V.decoder.middle[2].register_forward_hook(lambda mod, i, o: taps["middle"].append(o.clone())),
             V.decoder.upsamples[1].register_forward_hook(lambda mod, i, o: taps["up1"].append(o.clone()))]
    out = V.decode(z, [mean, 1.0 / std]).clamp(-1, 1)
    for h in hooks:
        h.remove()
    _save("g7_vae_decode", dict(out=out, middle=torch.cat(taps["middle"], dim=2), up1_last=taps["up1"][-1], in_sum=C.checksum(dict(z=z)), w_sum=C.checksum(vsd)))

    # ---- G8: VAE encode (posterior mode, normalised) on a 9-frame clip (chunks 1+4+4) and a single image
    VE = ref.vae.AutoencoderKLWan2_2_(dim=C.VAE_ENC_SMALL["dim"], dec_dim=16, z_dim=48,
                                      temperal_downsample=list(C.VAE_ENC_SMALL["temporal_down"])).eval()
    esd = C.vae_enc_weights(C.VAE_ENC_SMALL, prefix="")
    missing, unexpected = VE.load_state_dict(esd, strict=False)
    assert not unexpected and all(k.startswith(("decoder.", "conv2.")) for k in missing), (missing, unexpected)
    mean, std = torch.tensor(OV.LATENT_MEAN), torch.tensor(OV.LATENT_STD)
    xv, xi = C.vae_enc_case(), C.vae_enc_case(seed=43, frames=1, h=32, w=32)
    with torch.no_grad():
        mu_v = VE.encode(xv, [mean, 1.0 / std])[:, :48]
        mu_i = VE.encode(xi, [mean, 1.0 / std])[:, :48]
    _save("g8_vae_encode", dict(mu_video=mu_v, mu_image=mu_i, in_sum=C.checksum(dict(v=xv, i=xi)), w_sum=C.checksum(esd)))

    # ---- G10: vendored multistep samplers (UniPC, DPM-Solver++): sigma schedule + trace of step() outputs
    for name in sorted(C.SOLVER_CASES):
        kind, steps, shift, kw, x, vs = C.solver_case(name)
        if kind == "unipc":
            sch = ref.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, **kw)
            sch.set_timesteps(steps, device="cpu", shift=shift)                          # PIPE.py:606-608
        else:
            sch = ref.dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, **kw)
            ref.dpm.retrieve_timesteps(sch, device="cpu", sigmas=ref.dpm.get_sampling_sigmas(steps, shift))   # PIPE.py:609-614
        trace, cur = [], x.clone()
        gen = torch.Generator().manual_seed(C.SOLVER_NOISE_SEED)         # consumed by the SDE cases only
        for i, t in enumerate(sch.timesteps):
            cur = sch.step(vs[i], t, cur, generator=gen, return_dict=False)[0]
            trace.append(cur.clone())
        _save("g10_solver_" + name, dict(sigmas=sch.sigmas.clone(), timesteps=sch.timesteps.clone(), trace=torch.stack(trace),
                                         in_sum=C.checksum(dict(x=x, **{f"v{i}": v for i, v in enumerate(vs)}))))

    # ---- G11: RIFLEx rope table (enable_riflex defaults k=6, L_test=66, L_test_scale=4.886; FX.py:57-113, 774-788)
    tiny = _ref_dit(ref, dict(O.DIT_TINY), C.dit_weights(dict(O.DIT_TINY), 7))
    tiny.enable_riflex()
    riflex = torch.angle(tiny.freqs.to(torch.complex128)) if tiny.freqs.is_complex() else tiny.freqs.double()
    tiny.disable_riflex()
    base = torch.angle(tiny.freqs.to(torch.complex128))
    tiny.enable_riflex(k=2, L_test=3, L_test_scale=1.0)           # a setting that visibly moves a 3-frame clip
    rcase = C.dit_case(dict(O.DIT_TINY), 41)
    with torch.no_grad():
        rout = tiny(**rcase)
    tiny.disable_riflex()
    _save("g11b_dit_riflex", dict(out=rout))
    _save("g11_riflex", dict(cis_real=tiny.freqs.real.float()[:64].contiguous(), riflex_cos=torch.cos(riflex).float().contiguous(),
                             riflex_sin=torch.sin(riflex).float().contiguous(), base_cos=torch.cos(base).float().contiguous()))

    # ---- G12: umT5 text encoder (WanT5EncoderModel.forward) on a tiny config, right-padded prompts of 24 and 9 tokens
    from . import t5 as OT
    tcfg = dict(OT.T5_TINY)
    T5 = ref.t5.WanT5EncoderModel(**tcfg).eval()
    tsd = OT.seeded_t5_weights(tcfg, 5)
    T5.load_state_dict(tsd)
    ids, amask = OT.t5_case(tcfg)
    with torch.no_grad():
        emb = T5(ids, amask)[0]
    _save("g12_t5", dict(out=emb, ids=ids.float(), mask=amask.float(), w_sum=C.checksum(tsd)))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    sys.exit(main())
