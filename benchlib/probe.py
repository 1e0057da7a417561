"""Layout probe of an N >= 4 run: which exchange around self-attention is faster depends on what the links of THIS node deliver (the
K|V all-gather moves N/2 times the bytes of the all-to-all over heads but hides part of them; the all-to-all uses every link of the
mesh when the CFG pair is batched).  Two steps of each candidate layout on the real fabric decide; the line says so."""
import os
import time

import torch

PROBE_ENV = ("FLEXAM_SP_MODE", "FLEXAM_CFG_PARALLEL", "FLEXAM_SP_OVERLAP", "FLEXAM_SP_PIECES")


def candidates(world: int, num_heads: int):
    """(name, FLEXAM_SP_MODE, cfg_parallel, FLEXAM_SP_OVERLAP, FLEXAM_SP_PIECES).  The first one is the default layout (what runs when
    nothing is measured): one K|V gather per block, waited for, ONE attention call -- the fastest form on compute (r5: 41.4 against 48.1
    ms per rank step at 8 GPUs for the overlapped form, whose partial-softmax machinery pays only on slow links: measured here)."""
    cands = [(f"cfg2 x sp{world // 2}, K|V all-gather in one piece, waited for (default)", "allgather", True, "0", "1"),
             (f"cfg2 x sp{world // 2}, K|V all-gather, local-chunk-first attention under the gather", "allgather", True, "1", None)]
    if world // 2 >= 4:
        cands.append((f"cfg2 x sp{world // 2}, K|V all-gather in one piece, local-chunk-first attention under it", "allgather", True, "1", "1"))
    if num_heads % world == 0:
        cands.append((f"cfg1 x sp{world}, all-to-all over heads, a sample's blocks leave under the other's projection", "ulysses", False, "1", None))
        cands.append((f"cfg1 x sp{world}, all-to-all over heads, samples fully pipelined (attention per sample)", "ulysses", False, "2", None))
    if num_heads % (world // 2) == 0:
        cands.append((f"cfg2 x sp{world // 2}, all-to-all over heads", "ulysses", True, "1", None))
    return cands


def _warm_groups(model, device):
    """One tiny collective on every process group the layout uses, BEFORE anything is timed: RCCL builds a communicator lazily on
    its first collective (seconds per communicator on a cold node) and that one-off cost says nothing about the layout's step."""
    import torch.distributed as dist
    par = getattr(model, "_parallel", None) or {}
    groups = {id(g): g for g in (par.get("world_group"), par.get("sp_group")) if g is not None}
    t = torch.zeros(8, device=device)
    dist.all_reduce(t)                                         # the world group (the head gather / latent agreement use it)
    for g in groups.values():
        dist.all_reduce(t, group=g)
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size(g))]
        dist.all_gather(outs, t, group=g)
    torch.cuda.synchronize()


def probe_layouts(model, pipe, inp, cond, cfg, world, device, total_steps):
    """Measures every candidate layout (2 steps each) and leaves the fastest one enabled on `model`.  Returns the `layout_probe`
    object of the JSON line.

    Per-candidate guard: a candidate that raises (every rank the same way: configuration errors) is recorded in `skipped` and the
    probe goes on; all ranks decide on all-reduced values.  The first-step budget covers `denoise_step(0)` ONLY -- communicator
    set-up (a warm-up collective on the layout's groups), the engine rebuild and the per-clip `prepare()` come before the clock
    starts (round-4 advice: on a cold 8-GPU node those one-off costs could push the first candidate that touches a communicator over
    the budget for reasons unrelated to its step time).  A candidate that still exceeds the budget is tried ONCE more at the end,
    when everything it touches is warm.  A rank that dies or hangs alone cannot be skipped over in-process: the process-group
    timeout (FLEXAM_BENCH_PG_TIMEOUT) ends the attempt and the launcher's second attempt runs the conservative exchange without a
    probe."""
    import torch.distributed as dist
    budget = float(os.environ.get("FLEXAM_BENCH_PROBE_BUDGET", "20"))       # seconds for a candidate's first step (single GPU: 0.27 s)
    probe = {"candidates": [], "steps": 2, "first_step_budget_sec": budget,
             "budget_covers": "denoise_step(0) only (communicators warmed, engine rebuilt and prepare() run before the clock starts)"}
    test_hooks = os.environ.get("FLEXAM_BENCH_TEST_HOOKS") == "1"
    cands = candidates(world, cfg["num_heads"])

    def agreed_max(x):
        tt = torch.tensor([x], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def measure(idx, retry):
        name, mode, cfgp, ovl, pcs = cands[idx]
        os.environ["FLEXAM_SP_MODE"], os.environ["FLEXAM_SP_OVERLAP"] = mode, ovl
        os.environ.pop("FLEXAM_SP_PIECES", None)
        if pcs is not None:
            os.environ["FLEXAM_SP_PIECES"] = pcs
        t_c = time.perf_counter()
        try:
            if test_hooks and os.environ.get("FLEXAM_BENCH_PROBE_RAISE") == str(idx):
                raise RuntimeError("test hook: this candidate raises on every rank")
            model.enable_multi_gpus_inference(cfg_parallel=cfgp)
            model._engine = None                      # the engine (buffers, per-clip state) is rebuilt for the layout on its next use
            _warm_groups(model, device)
            pipe.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
            torch.cuda.synchronize()
            dist.barrier()
            t_first = time.perf_counter()
            if test_hooks and not retry and os.environ.get("FLEXAM_BENCH_PROBE_SLOW_FIRST") == str(idx):
                time.sleep(budget + 0.5)              # test hook: the first touch of this candidate is over budget, the retry is not
            pipe.denoise_step(0)
            torch.cuda.synchronize()
            first = agreed_max(time.perf_counter() - t_first)
            if first > budget:
                return {"layout": name, "error": f"first step took {first:.1f} s (> {budget:.0f} s budget)", "wall_sec": round(time.perf_counter() - t_c, 2),
                        "over_budget": True}
            dist.barrier(); torch.cuda.synchronize()
            tq = time.perf_counter()
            for i in range(2):
                pipe.denoise_step(1 + i)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            ms = agreed_max(time.perf_counter() - tq) / 2 * 1e3
            probe["candidates"].append({"layout": name, "ms_per_step": ms, "mode": mode, "cfg_parallel": cfgp, "overlap": ovl, "pieces": pcs,
                                        "first_step_sec": round(first, 3), "wall_sec": round(time.perf_counter() - t_c, 2), "retried": retry})
            return None
        except Exception as e:                        # noqa: BLE001  (raised identically on every rank, or the PG timeout ends the attempt)
            return {"layout": name, "error": f"{type(e).__name__}: {e}", "wall_sec": round(time.perf_counter() - t_c, 2)}

    again = []
    for idx in range(len(cands)):
        skipped = measure(idx, False)
        if skipped is not None:
            if skipped.pop("over_budget", False):
                again.append((idx, skipped))
            else:
                probe.setdefault("skipped", []).append(skipped)
    for idx, first_try in again:                      # once more, now that every communicator and buffer it touches exists
        skipped = measure(idx, True)
        if skipped is not None:
            skipped.pop("over_budget", None)
            skipped["first_try"] = first_try["error"]
            probe.setdefault("skipped", []).append(skipped)
    if probe["candidates"]:
        best = min(probe["candidates"], key=lambda c: c["ms_per_step"])       # identical on every rank (all-reduced times)
        if test_hooks and os.environ.get("FLEXAM_BENCH_LAYOUT_FORCE"):        # test hook: run candidate i whatever the probe measured
            best = probe["candidates"][int(os.environ["FLEXAM_BENCH_LAYOUT_FORCE"])]
        probe["chosen"] = best["layout"]
        os.environ["FLEXAM_SP_MODE"] = best["mode"]
        os.environ["FLEXAM_SP_OVERLAP"] = best["overlap"]  # always pinned: the engine's default differs per exchange (gather 0, all-to-all 1)
        os.environ.pop("FLEXAM_SP_PIECES", None)
        if best["pieces"] is not None:
            os.environ["FLEXAM_SP_PIECES"] = best["pieces"]
        model.enable_multi_gpus_inference(cfg_parallel=best["cfg_parallel"])
    else:                                             # every candidate refused or over budget: keep the default layout
        probe["chosen"] = "none measured: default layout"
        for k in ("FLEXAM_SP_MODE", "FLEXAM_SP_OVERLAP", "FLEXAM_SP_PIECES"):
            os.environ.pop(k, None)
        model.enable_multi_gpus_inference(cfg_parallel=None)
    model._engine = None
    from flexam_amd.dist import live_subgroups
    probe["communicators"] = 1 + live_subgroups()     # the world group + the cached CFG-half groups (created once per member set)
    return probe
