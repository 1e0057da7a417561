"""N > 1: starting the ranks without touching the GPU in the parent, and the N-rank self-check after the timed region."""
import json
import math
import os
import sys


def multi_gpu_check(pipe, model, inp, cond, step_index, total_steps, world, rank):
    """Self-validation of an N-rank run, AFTER the timed region.  Every rank applies the identical sampler update to the gathered
    head output, so "all ranks hold the same latents" alone would also pass with a wrong K|V exchange.  Two checks:
      (1) ranks_agree: checksum of every rank's latents == rank 0's (they are bit-identical by construction);
      (2) rel_rms_vs_single_gpu: one more denoise step from the current latents on the N-rank layout, and the SAME step on this
          rank alone (weights are replicated: a second engine with no parallel layout, full CFG pair, no collective); the DiT's
          head outputs of the two runs -- per CFG row, BEFORE the guidance combine, which multiplies any difference by ~8 at
          guidance 6 -- must agree to the rounding of two summation orders: relative RMS <= 1.5e-2, the library's stated bf16
          tolerance (typical 2-6e-3; a wrong or missing remote chunk shows up as O(1)).
    Returns the `check` object of the JSON line (rank 0's view + the worst rank)."""
    import torch.distributed as dist
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM, hip
    tol = 1.5e-2

    def step_and_grab(p):
        got = {}
        orig = p._sampler_update

        def grab(i, tok_u, tok_c):
            got["rows"] = [t.double().clone() for t in (tok_u, tok_c) if t is not None]
            return orig(i, tok_u, tok_c)
        p._sampler_update = grab
        try:
            p.denoise_step(step_index)
        finally:
            p._sampler_update = orig
        return got["rows"]
    st = pipe._state
    lat0 = st["latents"].clone()
    rows_multi = step_and_grab(pipe)
    sums = [None] * world
    dist.all_gather_object(sums, hip.checksum(st["latents"]))
    agree = all(tuple(c) == tuple(sums[0]) for c in sums)
    layout = model._parallel
    model._parallel, model._engine = None, None                 # a fresh engine: one GPU, no collective
    solo = Wan2_2FunControlPipeline_FlexAM(transformer=model)
    solo.prepare(inp["latents"], cond, inp["ctx_c"], inp["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=total_steps)
    solo._state["latents"].copy_(lat0)
    rows_solo = step_and_grab(solo)
    rel = max(float(((m - s_).pow(2).mean().sqrt() / s_.pow(2).mean().sqrt().clamp_min(1e-30)).item()) for m, s_ in zip(rows_multi, rows_solo))
    model._parallel, model._engine = layout, None
    rels = [None] * world
    dist.all_gather_object(rels, rel)
    worst = max(rels)
    ok = bool(agree and worst <= tol and math.isfinite(worst))
    if (os.environ.get("FLEXAM_BENCH_TEST_HOOKS") == "1" and os.environ.get("FLEXAM_BENCH_FORCE_CHECK_FAIL") == "1"
            and os.environ.get("FLEXAM_SP_OVERLAP") != "0"):
        ok = False                       # test hook (tests/test_bench_launch.py): exercises the launcher's fallback attempt
    return {"ok": ok, "ranks": world, "ranks_agree": bool(agree), "rel_rms_vs_single_gpu": rel, "worst_rank_rel_rms": worst, "tolerance": tol,
            "what": "DiT head output of one denoise step per CFG row (before the guidance combine): N-rank layout vs the same step on "
                    "one GPU (no collective), every rank; checksums of the N ranks' latents"}


def visible_gpus_without_hip():
    """GPUs this process may use, counted from the KFD topology in sysfs (nodes with SIMDs), cut by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when they are plain index lists.  No HIP / HSA call: the launcher must not initialise the runtime before
    it starts the ranks (torch.cuda.device_count() only avoids HIP while its amdsmi path works).  None when sysfs says nothing."""
    import glob
    n = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0                                     # no KFD driver: no AMD GPU on this host
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            if all(x.strip().isdigit() for x in ids):
                n = min(n, len(ids))
    return n


def launch_ranks(args, script):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks ourselves (`script`: bench.py's path).

    This parent never touches the GPU (no HIP call, no torch.cuda.is_available()): the ranks are FRESH child processes of
    `python -m torch.distributed.run`, never an exec of a process that has initialised the device.  Rank 0's JSON line and the
    launcher's exit code are forwarded.  The first attempt runs the default exchange; when its
    self-check (`check.ok`, see multi_gpu_check) fails, it crashes or it hangs, ONE more attempt runs the conservative form of the
    same exchange (FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0: one K|V all-gather per block, waited for before attention) and the line
    says so in `launch.fallback` -- a wrong or dead overlap path must not cost the scaling measurement."""
    import signal
    import socket
    import subprocess

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    n_dev = visible_gpus_without_hip()               # None: cannot tell without touching HIP -> the rank children report it
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev is not None and n_dev < args.gpus and env0.get("FLEXAM_BENCH_ONE_DEVICE") != "1":
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible (FLEXAM_BENCH_ONE_DEVICE=1 FLEXAM_BENCH_BACKEND=gloo "
                         f"runs the rank code path on one device for validation; such a line is marked invalid)")
    attempts = [("default", {})]
    if "FLEXAM_SP_OVERLAP" not in os.environ and "FLEXAM_SP_PIECES" not in os.environ and args.gpus > 2:
        attempts.append(("FLEXAM_SP_PIECES=1 FLEXAM_SP_OVERLAP=0", {"FLEXAM_SP_PIECES": "1", "FLEXAM_SP_OVERLAP": "0"}))
    limit = float(os.environ.get("FLEXAM_BENCH_ATTEMPT_TIMEOUT", "600"))
    last_rc, last_line, notes = 1, None, []
    for name, extra in attempts:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), script, *sys.argv[1:]]
        proc = subprocess.Popen(cmd, env={**env0, **extra, "FLEXAM_BENCH_SPAWNED": "1"}, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=limit)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the process group this launcher created, nothing else
            out, _ = proc.communicate()
            rc = 124
        line = None
        for ln in (out or "").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            else:
                print(ln, file=sys.stderr)
        ok = rc == 0 and line is not None
        if ok:
            res = json.loads(line)
            chk = res.get("check")
            ok = chk is None or bool(chk.get("ok"))
            res["launch"] = {"spawned_by": "bench.py (parent made no GPU call)", "attempt": name, "earlier_attempts": notes or None,
                             "fallback": name != "default"}
            line = json.dumps(res)
        last_rc, last_line = rc, line
        if ok:
            break
        failed = json.loads(line) if line else {}
        notes.append({"attempt": name, "rc": rc, "check": failed.get("check"), "parallelism": failed.get("config", {}).get("parallelism"),
                      "layout_probe": failed.get("layout_probe")})
    if last_line is not None:
        print(last_line, flush=True)
    sys.exit(last_rc if last_rc != 0 else (0 if last_line is not None else 1))

