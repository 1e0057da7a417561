#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec of the FlexAM hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N rank processes, see launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --mask blob                           (BASELINE configs[3]: foreground_edit masks, its own JSON line)
  python bench.py --logit-scale 6                       (peaked softmax rows: prices the attention kernel's rescale path inside the step)
  python bench.py --emulate-rank 8                      (ONE process runs one rank's share of an 8-GPU step; collectives = device copies)

Workload (BASELINE config 2): Wan2.2-Fun-5B-FLEXAM DiT (dim 3072, 24 heads x 128, ffn 14336,
30 layers, 5.03 B parameters, random-init bf16), 97x512x896 -> latent [1,48,25,32,56], L = 11648 tokens
(448 reference-image tokens + 11200 video tokens), CFG pair (B = 2), flow-match Euler, 50-step schedule,
synthetic seeded conditioning.  One *step* = one iteration of the reference loop
(pipeline_wan2_2_fun_control_FlexAM.py:844-949): two DiT sample-forwards + CFG + Euler + masked blend.
N > 1: one process per GPU (strong scaling of ONE clip).  Default layout: the two CFG rows are split
first (independent until the guidance combine: no per-block traffic), then the token sequence into N/2 contiguous chunks whose
post-norm K|V are all-gathered per block over RCCL/xGMI ("allgather", the collective BASELINE.json's north_star names; since r6 ONE
gather per block, waited for, then ONE attention call -- FLEXAM_SP_OVERLAP=1 selects the local-chunk-first form that attends to the
local chunk under the gather).  FLEXAM_SP_MODE=ulysses selects the all-to-all-over-heads exchange instead (then N >= 4 runs N-way
token chunks with the CFG pair batched), FLEXAM_CFG_PARALLEL=0/1 overrides the CFG split; for N >= 4 with nothing pinned the layout
probe times two steps of every candidate on the node and runs the fastest; `config.parallelism` names what ran.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the self-attention flash kernel, timed
live with events on the launch stream) and `cpu_baseline` (the fp32 oracle on the host cores, one of the
60 block-forwards of a step, extrapolated -- a baseline, not a target).

The pieces live in benchlib/ (inputs, kernels, cpu_baseline, vae_clip, launch, probe, emulate); this file is the argument parser, the
timed region and the JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from benchlib.inputs import (PEAK_BF16_TFLOPS, PEAK_FP8_TFLOPS, PEAK_HBM_GBS, blob_mask_pixels, block_flops, build_model,  # noqa: E402,F401
                             set_logit_scale, synthetic_inputs)
from benchlib.launch import launch_ranks, multi_gpu_check, visible_gpus_without_hip  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=97)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=896)
    ap.add_argument("--layers", type=int, default=30, help="debug only: fewer layers makes the number INVALID")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: QKV / FFN GEMMs on fp8 MFMA; a SEPARATE line (dtype fp8), never the headline")
    ap.add_argument("--sage", action="store_true", help="self-attention on MXFP8 operands (the reference's VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION switch); "
                    "a SEPARATE line (dtype says so), never the headline")
    ap.add_argument("--fp8-oproj", action="store_true", help="with --fp8: the two output projections of a block on the fp8 pipe as well (FLEXAM_FP8_OPROJ=1; "
                    "beyond configs[4]'s 'QKV/FFN', its own line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-legs", action="store_true", help="skip the config-1 and VAE-chunk CPU baseline legs (keep the one-block leg)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="skip the end-to-end clip (pixels -> encode -> 50 steps -> decode -> frames)")
    ap.add_argument("--mask", choices=["motion", "blob", "blob-open", "soft"], default="motion",
                    help="motion: BASELINE configs[1] (motion_transfer, frame 0 known); blob: configs[3] foreground_edit as demo.py "
                         "builds it; blob-open / soft: foreground masks that are not pinned (many per-token timesteps)")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the cross-rank / single-GPU self-check after the timed region")
    ap.add_argument("--logit-scale", type=float, default=None, metavar="S",
                    help="multiply every self_attn.norm_q / norm_k weight by sqrt(S): q.k / sqrt(128) then has std ~S instead of ~1 (peaked softmax rows, "
                         "the attention kernel's deferred-rescale branch taken); a SEPARATE line that says so, never the headline")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="N",
                    help="after the single-GPU measurement: one rank's share of an N-GPU step per layout, collectives replaced by same-size device "
                         "copies on a side stream -> `emulated_ranks` / `predicted_scaling_no_comm` (benchlib/emulate.py); default: 8 on the headline "
                         "workload (a few seconds, after the timed region), off on the variant lines; 0 = off")
    ap.add_argument("--emulate-which", type=int, default=None, help="which rank of the sequence-parallel group to emulate (default: a middle chunk)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args, os.path.abspath(__file__))      # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # FLEXAM_BENCH_ONE_DEVICE=1 + FLEXAM_BENCH_BACKEND=gloo: validation of the multi-rank code path on a box with ONE GPU
    # (all ranks share cuda:0, collectives go through the host); such a line is marked invalid below.
    one_device = os.environ.get("FLEXAM_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("FLEXAM_BENCH_BACKEND", "nccl")
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective that one rank never joins must end the attempt long before the launcher's limit (600 s), so that the
        # conservative second attempt still fits the driver's window
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("FLEXAM_BENCH_PG_TIMEOUT", "150")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
    from flexam_amd.configs import WAN22_FUN_5B_FLEXAM
    cfg = dict(WAN22_FUN_5B_FLEXAM, num_layers=args.layers)
    model = build_model(cfg, device)
    if args.fp8_oproj:
        if not args.fp8:
            raise SystemExit("--fp8-oproj needs --fp8")
        os.environ["FLEXAM_FP8_OPROJ"] = "1"                        # read by the engine at every forward
    if args.fp8:
        model.enable_fp8_gemm(True)
    if args.logit_scale is not None:
        set_logit_scale(model, args.logit_scale)
    if args.sage:
        os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"      # read by the engine at every forward, as the reference's attention() does
    if world > 1:
        cp = os.environ.get("FLEXAM_CFG_PARALLEL")
        model.enable_multi_gpus_inference(cfg_parallel=None if cp is None else cp == "1")
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model)
    def conditioning(mode):
        i = synthetic_inputs(args.frames, args.height, args.width, cfg["text_dim"], mode)
        return i, LatentConditioning(control_latents=i["control"], additional_control=i["additional"], masked_video_latents=i["masked"],
                                     ref_latents=i["ref"], mask_latents=i["mask_latents"], mask=i["mask"], mask_pixels=i["mask_pixels"])
    inp, cond = conditioning(args.mask)
    total_steps = 50
    layout_probe = None
    from benchlib.probe import PROBE_ENV, probe_layouts
    pinned = [k for k in PROBE_ENV if k in os.environ]
    if world >= 4 and not pinned and os.environ.get("FLEXAM_BENCH_LAYOUT_PROBE", "1") != "0":
        layout_probe = probe_layouts(model, pipe, inp, cond, cfg, world, device, total_steps)
    pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    torch.cuda.synchronize()
    tp0 = time.perf_counter()                       # second call: buffers exist, this is the per-clip cost of the step-invariant work
    pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    torch.cuda.synchronize()
    prepare_sec = time.perf_counter() - tp0
    eng = model.engine()
    L = eng.cond["L"]
    B = 2

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        pipe.denoise_step(i % total_steps)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pipe.denoise_step((args.warmup + i) % total_steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(pipe._state["latents"]).all())
    rows_u = int(pipe._state["U"])
    # host side of a step: how long Python + ctypes need to ENQUEUE one step's ~420 launches (the GPU is idle-free while this
    # stays below the step's GPU time; at N ranks the GPU time shrinks N-fold, the enqueue time does not)
    sync()
    th0 = time.perf_counter()
    for i in range(2):
        pipe.denoise_step((args.warmup + args.steps + i) % total_steps)
    host_enqueue = (time.perf_counter() - th0) / 2
    sync()

    def timed(p):
        for i in range(args.warmup):
            p.denoise_step(i % total_steps)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            p.denoise_step((args.warmup + i) % total_steps)
        sync()
        return time.perf_counter() - t0
    motion_elapsed = None
    if args.mask != "motion" and world == 1:
        # configs[3] next to configs[1]: the SAME steps on the motion_transfer conditioning, in this process (the delta is what
        # the foreground masks cost: per-token timestep rows, AdaLN table size)
        inp_m, cond_m = conditioning("motion")
        pipe.prepare(inp_m["latents"], cond_m, inp_m["ctx_c"], inp_m["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
        motion_elapsed = timed(pipe)
        pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
        del inp_m, cond_m


    from benchlib.kernels import kernel_rooflines, kernels_object, newest_profile, roofline_object, sampler_step_roofline, self_attention_in_step
    lc = L // eng.sp_size
    b_local = 1 if eng.cfg_size == 2 else B
    kern = None if args.no_kernel_timing else kernel_rooflines(eng, b_local, L, lc)
    attn_in_step = None
    if kern is not None and world == 1:
        attn_in_step = self_attention_in_step(pipe, (args.warmup + args.steps + 2) % total_steps, b_local)
    if kern is not None and pipe._state.get("known") is not None:
        kern["cfg_euler_blend"] = sampler_step_roofline(pipe)
    check = None
    inproc_fallback = None
    if world > 1 and not args.no_check:
        check = multi_gpu_check(pipe, model, inp, cond, (args.warmup + args.steps) % total_steps, total_steps, world, rank)
        # Started by someone else's torch.distributed.run (not by launch_ranks, which has its own second attempt): a failed check of
        # the default exchange gets ONE more measurement in this process on the conservative form of the same exchange -- one K|V
        # all-gather per block, waited for before attention.  Every rank takes the same decision (check.ok is built from all-gathered
        # values); the engine is rebuilt on its next use (multi_gpu_check leaves model._engine = None) and reads the switches again.
        if not check["ok"] and os.environ.get("FLEXAM_BENCH_SPAWNED") != "1" and eng.sp_size > 1 and not pinned:
            inproc_fallback = {"attempt": "default", "check": check, "ms_per_step": elapsed / args.steps * 1e3}
            os.environ["FLEXAM_SP_PIECES"], os.environ["FLEXAM_SP_OVERLAP"] = "1", "0"
            os.environ.pop("FLEXAM_SP_MODE", None)        # the conservative form IS the K|V all-gather, whatever the probe had picked
            model.enable_multi_gpus_inference(cfg_parallel=None)
            model._engine = None
            pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
            eng = model.engine()
            elapsed = timed(pipe)
            tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            finite = bool(torch.isfinite(pipe._state["latents"]).all())
            check = multi_gpu_check(pipe, model, inp, cond, (args.warmup + args.steps) % total_steps, total_steps, world, rank)
    eng_sage = bool(getattr(eng, "sage_taken", False))          # what the engine DID (SAGE_ATTENTION is ignored under sequence parallelism)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from benchlib.cpu_baseline import cpu_baseline, cpu_baseline_legs
        base = cpu_baseline(L, cfg)
        if not args.no_cpu_legs:
            base["legs"] = cpu_baseline_legs(cfg)
    eng_cfg, eng_sp, eng_mode = eng.cfg_size, eng.sp_size, getattr(eng, "sp_mode", "-")
    eng_cross_lk = eng.cond.get("cross_lk")
    eng_share0 = bool(getattr(eng, "share0_taken", False))      # what the engine DID in its last run, not what a switch asked for
    if eng_mode == "allgather":
        pcs = getattr(eng, "sp_pieces", 1)
        eng_mode = (f"K|V all-gather per block in {pcs} head-group piece(s), " +
                    ("each consumed as it lands: Q projection + local-chunk attention under piece 0 (partial softmaxes merged), piece g+1 under the attention of group g"
                     if getattr(eng, "sp_overlap", False) else "waited for before attention (overlaps the Q projection only)"))
    elif eng_mode == "ulysses":
        lvl = getattr(eng, "sp_overlap_level", 0)
        eng_mode = "all-to-all over heads (q|k|v out, attention output back)" + (
            (", a sample's blocks leave under the other sample's projection, one attention call for the pair" if lvl == 1 else
             ", the samples of the CFG pair as pipeline stages: a sample's blocks travel under the other's projection / attention")
            if (b_local > 1 and lvl > 0) else "")
    vae_sec = enc_sec = enc_stream_sec = None
    emulated = raster = None
    headline = not (args.fp8 or args.sage or args.logit_scale is not None or args.mask != "motion" or args.layers != 30
                    or (args.frames, args.height, args.width) != (97, 512, 896))
    n_emulate = args.emulate_rank if args.emulate_rank is not None else (8 if headline else 0)
    if n_emulate and rank == 0 and world == 1:
        from benchlib.emulate import emulate
        pipe = eng = None
        torch.cuda.empty_cache()
        try:                                             # an add-on after the timed region: it must not cost the steps/s line if it fails
            emulated = emulate(model, lambda: Wan2_2FunControlPipeline_FlexAM(transformer=model), inp, cond, n_emulate, args.steps, args.warmup,
                               total_steps, rank=args.emulate_which)
        except Exception as e:                           # noqa: BLE001
            emulated = [{"error": f"{type(e).__name__}: {e}"}]
            model._parallel, model._engine = None, None
        pipe = eng = None
    clip = None
    if rank == 0 and world == 1 and not args.no_vae:
        from benchlib.vae_clip import time_clip, time_vae
        pipe = eng = None
        torch.cuda.empty_cache()
        vae_sec, enc_stream_sec, enc_image_sec, vae_finite, vae = time_vae(device, args.frames, args.height, args.width)
        enc_sec = 7 * enc_stream_sec + enc_image_sec     # control, depth, 4 cos levels, masked video + the reference image
        band_sec = None
        if n_emulate and n_emulate > 1:
            try:                                         # the slowest rank's tile of the N-rank parallel decode (for the predicted sec/clip below)
                from benchlib.vae_clip import time_decode_band
                band_sec = time_decode_band(vae, device, args.frames, args.height, args.width, n_emulate)
            except Exception as e:                       # noqa: BLE001
                band_sec = None
        finite = finite and vae_finite
        try:                                             # the step before the encode: tracks -> conditioning videos
            from benchlib.vae_clip import time_raster
            raster = time_raster(device, args.frames, args.height, args.width, cpu_leg=not args.no_cpu_baseline)
        except Exception as e:                           # noqa: BLE001
            raster = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_clip and args.mask == "motion":
            try:                                         # the measured clip must not cost the steps/s line if it fails
                sec, shape, ok = time_clip(model, vae, inp, args.frames, args.height, args.width, total_steps, device)
                clip = {"sec": sec, "steps": total_steps, "output_shape": list(shape), "finite": ok,
                        "what": "Wan2_2FunControlPipeline_FlexAM.__call__ end to end: 8 pixel-space conditioning streams -> VAE encode -> "
                                "50 denoise steps -> VAE decode -> frames on the host (synthetic pixels, prompt embeddings given)"}
                finite = finite and ok
            except Exception as e:                       # noqa: BLE001
                clip = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        steps_per_sec = args.steps / elapsed
        blk = block_flops(L, cfg["dim"], cfg["ffn_dim"], cfg["text_len"])
        step_block_flops = blk * B * cfg["num_layers"]
        # Work the build removes is not counted as achieved FLOPs: with the CFG pair on one latent, block 0's q|k|v / o
        # projections and self-attention run once, not per sample (DiTEngine.run, share0)
        shared0 = eng_share0
        removed_flops = (8 * L * cfg["dim"] ** 2 + 4 * L * L * cfg["dim"]) if shared0 else 0
        # ... and the identical padded text rows are attended to as one weighted key (DiTEngine.set_conditioning, cross_lk)
        cross_lk = eng_cross_lk if eng_cross_lk else cfg["text_len"]
        removed_flops += 4 * L * (cfg["text_len"] - cross_lk) * cfg["dim"] * B * cfg["num_layers"]
        executed_block_flops = step_block_flops - removed_flops
        variant = args.fp8 or eng_sage or args.logit_scale is not None
        headline_shape = (args.frames, args.height, args.width) == (97, 512, 896)
        if headline_shape and not variant:
            tag = ("(BASELINE configs[1])" if args.mask == "motion" else
                   f"(BASELINE configs[3]: foreground_edit, mask '{args.mask}', {rows_u} distinct per-token timesteps per sample)")
        else:
            tag = ("(BASELINE configs[4] shape family: " + ("fp8 e4m3 QKV/FFN GEMMs with per-row / per-channel scales, everything else bf16/fp32" if args.fp8 else "bf16")
                   + (", self-attention on MXFP8 operands (VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION)" if eng_sage else "")
                   + (", o-projections on fp8 too (FLEXAM_FP8_OPROJ=1)" if args.fp8_oproj else "")
                   + (f", self-attention logits scaled to std ~{args.logit_scale:g} (norm_q / norm_k x sqrt of it: peaked softmax rows)" if args.logit_scale is not None else "")
                   + "; not the headline)")
        if eng_cfg == 2:
            par = (f"cfg{eng_cfg} x sp{eng_sp}: one CFG row per rank" + (", no per-block traffic" if eng_sp == 1 else
                   f", token-chunk sequence parallel inside each half; exchange around self-attention (RCCL): {eng_mode}"))
        else:
            par = (f"cfg1 x sp{eng_sp}: CFG pair batched on every rank, token-chunk sequence parallel over all ranks; "
                   f"exchange around self-attention (RCCL): {eng_mode}")
        result = {
            "metric": "denoise-steps/sec", "value": steps_per_sec, "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("fp8" if args.fp8 else "bf16") + (" + mxfp8 self-attention" if eng_sage else "") + (" + fp8 output projections" if args.fp8_oproj else ""),
            "data": "synthetic",
            "config": {"workload": f"Wan2.2-Fun-5B-FLEXAM DiT denoise step, {args.frames}x{args.height}x{args.width}, "
                                   f"L={L} tokens, CFG pair B=2, {cfg['num_layers']} layers, flow-match Euler (50-step schedule), "
                                   f"random-init bf16 weights, synthetic conditioning " + tag,
                       "parallelism": par if world > 1 else "single GPU", "layers": cfg["num_layers"]},
            "sec_per_clip_50_steps_denoise_only": total_steps / steps_per_sec,
            "vae_decode_sec": vae_sec, "vae_encode_sec_per_stream": enc_stream_sec, "conditioning_encode_sec_8_streams": enc_sec,
            "prepare_sec": prepare_sec,
            "conditioning_raster": raster,
            "clip_end_to_end": clip,
            "sec_per_clip": (enc_sec + prepare_sec + total_steps / steps_per_sec + vae_sec) if vae_sec is not None else None,
            "sec_per_clip_from_tracks": ((enc_sec + prepare_sec + total_steps / steps_per_sec + vae_sec + raster["sec"])
                                         if vae_sec is not None and raster and raster.get("sec") else None),
            "dit_block_executed_tflops": executed_block_flops * steps_per_sec / 1e12,
            "dit_block_executed_mfma_frac": executed_block_flops * steps_per_sec / 1e12 / (PEAK_BF16_TFLOPS * world),
            "dit_block_algorithmic_tflops": step_block_flops * steps_per_sec / 1e12,
            "dit_block_flops_note": ("executed = algorithmic block FLOPs (%.3f TF per block and sample x %d) minus what the build removes -- the block-0 self-attention "
                                     "half the CFG pair shares (taken: %s) and the identical padded text keys of cross-attention (keys attended: %d of %d): %.3f of "
                                     "%.1f TF per step; the fraction is executed FLOPs against the 2.5 PFLOP/s bf16 peak.  The algorithmic figure counts work that "
                                     "did not run: it is a throughput in the reference's units, not a roofline fraction"
                                     % (blk / 1e12, B * cfg["num_layers"], shared0, cross_lk, cfg["text_len"], removed_flops / 1e12, step_block_flops / 1e12))
                                    + (" (QKV / FFN ran on the 5 PFLOP/s fp8 pipe: not a roofline fraction)" if args.fp8 else ""),
            "finite": finite,
            "mask": args.mask, "timestep_rows_per_sample": rows_u,
            "host_enqueue_ms_per_step": host_enqueue * 1e3,
        }
        if args.sage and not eng_sage:
            result["sage_ignored"] = "--sage / VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION was asked for but the engine ran the bf16 attention kernel (sequence parallelism)"
        if args.logit_scale is not None:
            result["logit_scale"] = {"std_asked": args.logit_scale,
                                     "what": "every blocks.*.self_attn.norm_q / norm_k weight multiplied by sqrt(S): the scores q.k / sqrt(128) of the random-init "
                                             "model have std ~S instead of ~1, so rows carry maxima tens of exp2 units above their mean and the kernel's deferred "
                                             "rescale branch (any lane maximum > 2^8 above the running reference) is taken; compare ms_per_step and "
                                             "roofline.launch_ms with the default line of the same box"}
        if emulated is not None:
            from benchlib.emulate import HOST_NOTE
            clip_pred = None
            if vae_sec is not None and band_sec is not None:
                # the clip on N ranks before any link time: the 8 conditioning streams encoded round-robin (dist.shard_streams: ceil(8 / N) per rank),
                # prepare, 50 rank steps of the layout, the slowest rank's tile of the decode (tiles all-gathered: not included)
                jobs = [enc_stream_sec] * 7 + [enc_image_sec]                     # stream j on rank j % N: the slowest rank's share
                streams = max(sum(jobs[r::n_emulate]) for r in range(n_emulate))
                clip_pred = {"vae_encode_sec": streams, "vae_decode_tile_sec": band_sec, "prepare_sec": prepare_sec,
                             "per_layout": {r["layout"]: round(streams + prepare_sec + total_steps * r["ms_per_step"] * 1e-3 + band_sec, 3) for r in emulated if "ms_per_step" in r},
                             "per_layout_at_link_GBps": {r["layout"]: {k: round(streams + prepare_sec + total_steps * v * 1e-3 + band_sec, 3) for k, v in r["ms_per_step_at_link_GBps"].items()}
                                                         for r in emulated if "ms_per_step_at_link_GBps" in r},
                             "single_gpu_sec_per_clip": enc_sec + prepare_sec + total_steps / steps_per_sec + vae_sec,
                             "note": "sum of one rank's parts (7 videos + the reference image over the ranks, 50 rank steps, the slowest tile of the tiled decode) measured on "
                                     "this GPU; no collective is timed: a ceiling, NOT a multi-GPU measurement.  per_layout_at_link_GBps: the same sum with the rank steps "
                                     "of the link MODEL (link_time_note); the latent broadcasts and the all-gather of the tiles (67 MB per link) are not in it"}
            result["emulated_ranks"] = {"world": n_emulate, "layouts": emulated, "host_enqueue_note": HOST_NOTE,
                                        "predicted_scaling_no_comm": {r["layout"]: round(elapsed / args.steps * 1e3 / r["ms_per_step"], 3) for r in emulated if "ms_per_step" in r},
                                        "predicted_scaling_compute_only": {r["layout"]: round(elapsed / args.steps * 1e3 / r["ms_per_step_compute_only"], 3)
                                                                           for r in emulated if "ms_per_step_compute_only" in r},
                                        "predicted_scaling_at_link_GBps": {r["layout"]: {k: round(elapsed / args.steps * 1e3 / v, 3) for k, v in r["ms_per_step_at_link_GBps"].items()}
                                                                           for r in emulated if "ms_per_step_at_link_GBps" in r},
                                        "predicted_sec_per_clip_no_comm": clip_pred,
                                        "link_time_note": __import__("benchlib.emulate", fromlist=["LINK_NOTE"]).LINK_NOTE,
                                        "compute_only_note": "the same rank step with collectives that move nothing (stream plumbing only): predicted_scaling_no_comm keeps "
                                                             "the same-size device copies as a stand-in for the bytes a rank receives; where a gather is waited for, "
                                                             "those copies sit in front of the attention call and are counted as if they were compute",
                                        "what": "ONE process ran one rank's share of an N-GPU step per layout at full size: the real engine on that rank's token chunk / "
                                                "CFG row with every launch, piece, partial attention and merge of the multi-GPU path, collectives replaced by device "
                                                "copies of the same sizes on a side stream (flexam_amd.dist.LoopbackGroup).  predicted_scaling_no_comm = this run's "
                                                "single-GPU ms_per_step / the emulated rank's: the ceiling of N-GPU scaling before any xGMI time; NOT a multi-GPU measurement"}
        if motion_elapsed is not None:
            result["configs1_same_process"] = {"ms_per_step": motion_elapsed / args.steps * 1e3, "value": args.steps / motion_elapsed,
                                               "delta_pct": (elapsed / motion_elapsed - 1.0) * 100.0,
                                               "note": "the same steps on the motion_transfer conditioning (BASELINE configs[1]) in this process"}
        if layout_probe is not None:
            result["layout_probe"] = layout_probe
        if inproc_fallback is not None:
            result["launch"] = {"spawned_by": "the caller's torch.distributed.run", "attempt": "FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0 (second measurement in the same processes)",
                                "earlier_attempts": [inproc_fallback], "fallback": True}
        if check is not None:
            result["check"] = check
            result["rccl_ranks"] = world if backend == "nccl" else 0
            result["ranks_agree"] = check["ranks_agree"]
            if not check["ok"]:
                result["invalid"] = "multi-GPU self-check failed: " + json.dumps(check)
        if one_device or backend != "nccl":
            result["invalid"] = f"code-path validation only: {world} ranks on one device / backend {backend}"
        if kern is not None:
            result["roofline"] = roofline_object(ROOT, kern, attn_in_step, [b_local, L, cfg["num_heads"], cfg["dim"] // cfg["num_heads"]], world, eng_sage)
            result["kernels"] = kernels_object(kern)
            pmc = newest_profile(ROOT, "block_kernels_pmc.txt")
            result["kernels_note"] = ("live per-launch timing at this run's shapes; mfma rows: algorithmic FLOPs / 2.5 PFLOP/s, hbm rows: "
                                      "algorithmic bytes (SURVEY 8d) / 8 TB/s; counter-side traffic and MFMA-busy: "
                                      + (f"profiles/{pmc}" if pmc else "no PMC table committed"))
        if base is not None:
            result["cpu_baseline"] = base
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
