"""GPU: recorded launch plans (flexam_amd/hip.py record / Plan, csrc/replay.hip flexam_replay).  The engine records the blocks' and the
head's launches on the first denoise step and re-issues them from ONE C call per segment afterwards; a replayed step must be the SAME
step -- same kernels, same arguments, same order -- so every comparison here is bit for bit against the same run with FLEXAM_REPLAY=0
(every launch through its Python wrapper)."""
import os

import pytest
import torch

from oracle import cases as C
from oracle import dit as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def _model(cfg, seed=7):
    from flexam_amd import Wan2_2Transformer3DModel_FlexAM
    kw = dict(cfg)
    kw.pop("eps")
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
    m.load_state_dict(C.dit_weights(cfg, seed), strict=True)
    return m.to(DEV)


def test_plan_of_plain_launches_replays_on_new_contents():
    """hip.record() around a GEMM (+ tail split-K scratch), a LayerNorm and a gated residual; the plan re-run after the INPUTS changed
    gives what the direct calls give on the new inputs (addresses are recorded, contents are not), in ONE flexam_replay call."""
    from flexam_amd import hip
    g = torch.Generator().manual_seed(1)
    m, k, n = 2912, 512, 768                                  # 13 x 3 = 39 tiles of 224 x 256: a tail split-K launch + finish ride along
    a = torch.randn(m, k, generator=g).to(DEV, BF)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV, BF)
    b = torch.randn(n, generator=g).to(DEV)
    x = torch.randn(m, n, generator=g).to(DEV)
    w2 = (torch.randn(n, n, generator=g) / n ** 0.5).to(DEV, BF)
    gate = torch.randn(2, n, generator=g).to(DEV)
    y, h = torch.empty(m, n, device=DEV, dtype=BF), torch.empty(m, n, device=DEV, dtype=BF)

    def direct():
        hip.gemm(a, w, b, out=y, epilogue=hip.EPI_GELU_TANH)
        hip.ln_modulate(x, out=h)
        hip.gemm_gate_residual(h, w2, b, x, gate=gate, rows_per_batch=m // 2)
    x0 = x.clone()
    with hip.record() as plan:
        direct()
    assert plan.launches >= 3 and [kd for kd, _, _ in plan.items] == ["c"]
    first = (y.clone(), x.clone())
    x.copy_(x0)
    plan.run()
    assert torch.equal(y, first[0]) and torch.equal(x, first[1])
    a.copy_(torch.randn(m, k, generator=g).to(DEV, BF))      # new contents, same addresses
    x.copy_(torch.randn(m, n, generator=g).to(DEV))
    x1 = x.clone()
    plan.run()
    got = (y.clone(), x.clone())
    x.copy_(x1)
    direct()
    assert torch.equal(got[0], y) and torch.equal(got[1], x) and not torch.equal(y, first[0])
    # a failing command is reported with its index and the entry point's own message
    bad = hip.Plan()
    with hip.record() as bad:
        hip.ln_modulate(x, out=h)
    bad.items[0][1][0].a[0].p = None                         # corrupt the recorded x pointer
    with pytest.raises(RuntimeError, match=r"command 0 \(flexam_ln_modulate\)"):
        bad.run()


@pytest.mark.parametrize("variant", ["bf16", "fp8", "sage", "cfg_skip"])
def test_replayed_sampler_steps_equal_launch_by_launch_steps(variant, monkeypatch):
    """Four sampler steps of the tiny DiT (CFG pair: block 0's shared self-attention half = a torch copy inside the plan) with and
    without replay: latents after every step bit-identical; steps 2.. are replays (engine.replay_taken) of > 30 recorded launches."""
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    cfg = dict(O.DIT_TINY, num_layers=3)
    if variant == "sage":
        monkeypatch.setenv("VIDEOX_ATTENTION_TYPE", "SAGE_ATTENTION")
    traces, taken = {}, {}
    for replay in ("0", "1"):
        monkeypatch.setenv("FLEXAM_REPLAY", replay)
        m = _model(cfg)
        if variant == "fp8":
            m.enable_fp8_gemm(True)
        if variant == "cfg_skip":
            m.enable_cfg_skip(0.5, 4) if hasattr(m, "enable_cfg_skip") else None
        pipe = Wan2_2FunControlPipeline_FlexAM(transformer=m)
        sc = C.sampler_case(cfg)
        cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
        tr, tk = [], []

        def cb(p, i, t, k):
            tr.append(k["latents"].float().cpu().clone())
            tk.append((bool(getattr(m.engine(), "replay_taken", False)), getattr(m.engine(), "plan_launches", 0)))
        pipe(prompt_embeds=sc["context_cond"], negative_prompt_embeds=sc["context_uncond"], height=256, width=256, num_frames=9,
             num_inference_steps=4, guidance_scale=6.0, density=0.1, latents=sc["latents"], conditioning=cond, output_type="latent",
             callback_on_step_end=cb)
        traces[replay], taken[replay] = tr, tk
    assert len(traces["0"]) == len(traces["1"]) == 4
    for a, b in zip(traces["0"], traces["1"]):
        assert torch.equal(a, b)
    assert not any(t for t, _ in taken["0"])
    assert [t for t, _ in taken["1"]][1:] == [True] * 3 or variant == "cfg_skip"       # (cfg_skip alternates two row sets: each has its own plan)
    assert any(t for t, _ in taken["1"]) and max(n for _, n in taken["1"]) > 30


def test_new_conditioning_and_a_new_shape_get_new_plans():
    """Plans live in the per-clip state: a second clip (other conditioning CONTENTS in the same buffers' places, then another latent
    shape) must not run the first clip's launches.  forward() through the reference's call signature, replay on: equal to replay off."""
    cfg = dict(O.DIT_TINY)
    m = _model(cfg)
    outs = {}
    for replay in ("0", "1"):
        os.environ["FLEXAM_REPLAY"] = replay
        try:
            res = []
            for seed, (f, h, w) in ((41, (3, 16, 16)), (43, (3, 16, 16)), (41, (3, 16, 16)), (47, (5, 16, 24))):
                case = C.dit_case(cfg, seed, frames=f, h=h, w=w)
                d = {k: ([u.to(DEV) for u in v] if isinstance(v, list) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in case.items()}
                res.append(m(**d).float().cpu())
                assert not m.engine().replay_taken           # new conditioning (or a new shape): recorded afresh, never the old clip's launches
                res.append(m(**d).float().cpu())             # the same call again (a new per-token timestep tensor, as the reference's loop builds one per step)
                assert bool(m.engine().replay_taken) == (replay == "1")
            outs[replay] = res
        finally:
            os.environ.pop("FLEXAM_REPLAY", None)
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    assert torch.equal(outs["1"][0], outs["1"][1]) and not torch.equal(outs["1"][0], outs["1"][2])


def test_emulated_rank_replays_with_collectives_as_host_steps():
    """One emulated rank of four (LoopbackGroup: collectives = device copies on a side stream) in the three exchange forms: the plan
    interleaves C segments with host steps (issue / wait), and replayed steps equal launch-by-launch steps bit for bit."""
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from benchlib.emulate import set_emulated_layout
    cfg = dict(O.DIT_TINY, dim=512, num_heads=4, num_layers=2)
    sc = C.sampler_case(cfg)
    saved = {k: os.environ.get(k) for k in ("FLEXAM_SP_MODE", "FLEXAM_SP_OVERLAP", "FLEXAM_SP_PIECES", "FLEXAM_REPLAY")}
    try:
        for mode, cfgp, overlap, pieces in (("allgather", True, "1", "2"), ("allgather", False, "0", "1"), ("ulysses", False, "1", "1"), ("ulysses", False, "2", "1")):
            os.environ.update(FLEXAM_SP_MODE=mode, FLEXAM_SP_OVERLAP=overlap, FLEXAM_SP_PIECES=pieces)
            lat = {}
            for replay in ("0", "1"):
                os.environ["FLEXAM_REPLAY"] = replay
                m = _model(cfg)
                set_emulated_layout(m, 4, cfgp, 1)
                pipe = Wan2_2FunControlPipeline_FlexAM(transformer=m)
                cond = LatentConditioning(sc["control_latents"], sc["additional_control"], sc["masked_video_latents"], sc["ref_latents"], sc["mask_pixels"])
                pipe.prepare(sc["latents"], cond, sc["context_cond"], sc["context_uncond"], density=0.1, guidance_scale=6.0, num_inference_steps=4)
                steps = []
                for i in range(3):
                    pipe.denoise_step(i)
                    torch.cuda.synchronize()
                    steps.append((pipe._state["latents"].float().cpu().clone(), bool(m.engine().replay_taken)))
                lat[replay] = steps
                if replay == "1":
                    plan = next(iter(m.engine().cond["_plans"].values()))
                    kinds = [k for k, _, _ in plan.items]
                    assert "py" in kinds and "c" in kinds and [t for _, t in steps] == [False, True, True], (mode, kinds[:8])
            for (a, _), (b, _) in zip(lat["0"], lat["1"]):
                assert torch.equal(a, b), (mode, cfgp, overlap)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_delay_kernel_and_the_link_model_of_the_loopback_group():
    """flexam_delay_us holds a stream for the asked time on the constant 100 MHz counter (the emulated ranks' stand-in for transfer time),
    and a LoopbackGroup with an assumed link rate holds its side stream for bytes-per-link / rate + latency: an all-gather's time is ONE
    chunk over one link (every peer's chunk on its own link), an all-to-all's ONE block; it is a host step of a recorded plan, not part
    of a C segment."""
    from flexam_amd import hip
    from flexam_amd.dist import LoopbackGroup, all_gather_into_tensor, all_to_all_blocks
    for us in (200.0, 1000.0):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hip.delay_us(50.0)
        torch.cuda.synchronize()
        s.record(); hip.delay_us(us); e.record()
        torch.cuda.synchronize()
        got = s.elapsed_time(e) * 1e3
        assert us <= got <= us * 1.5 + 300.0, (us, got)                # (the lower bound is the contract; the upper one only catches a runaway wait)
    g = LoopbackGroup(4, 1, link_gbps=50.0, latency_us=10.0)
    chunk = torch.zeros(5 * 1000 * 1000, dtype=torch.uint8, device=DEV)          # 5 MB per peer link -> 100 us at 50 GB/s
    out = torch.empty(4 * chunk.numel(), dtype=torch.uint8, device=DEV)
    all_gather_into_tensor(out, chunk, group=g)                                   # warm (stream creation)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    w = all_gather_into_tensor(out, chunk, group=g, async_op=True)
    w.wait()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    assert 0.105 <= ms <= 1.5, ms                                                 # at least 100 us of "link" + 10 us latency; + the copies themselves and launch gaps
    blocks_in = [torch.zeros(1000 * 1000, dtype=torch.uint8, device=DEV) for _ in range(4)]      # 1 MB per link: 20 us + 10
    blocks_out = [torch.empty_like(b) for b in blocks_in]
    with hip.record() as plan:
        hip.host_op(lambda: all_to_all_blocks(blocks_out, blocks_in, group=g))
    assert [k for k, _, _ in plan.items] == ["py"] and plan.launches == 0          # the delay kernel inside the host step was not recorded
