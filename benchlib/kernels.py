"""Live per-kernel timing on the engine's own buffers, the dominant kernel timed inside the step, and the `roofline` object of the JSON line."""
import json
import os

import torch

from .inputs import PEAK_BF16_TFLOPS, PEAK_FP8_TFLOPS, PEAK_HBM_GBS


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()                 # recorded on torch's current stream = the stream every flexam_* call launches on
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def self_attention_in_step(pipe, step_index, B):
    """The dominant kernel timed where it runs: HIP events (torch's current stream = the stream every flexam_* call launches on)
    around every self-attention call of ONE more denoise step, outside the timed region.  Returns the mean over the calls that run
    the whole batch (29 of 30: block 0's call covers one sample when the CFG pair shares its self-attention half) in seconds --
    the figure rocprofv3 --kernel-trace reports as that kernel's average inside the step.  The isolated back-to-back timing
    (kernel_rooflines) runs at another clock: inside the step the clock is set by the GEMMs around the call."""
    from flexam_amd import hip
    real, marks = hip.attn_fwd, []

    def timed_attn(q, k, v, *a, **kw):
        if k.shape[1] <= 1024 or q.shape[1] != k.shape[1]:          # text cross-attention / partial calls: not the kernel in question
            return real(q, k, v, *a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = real(q, k, v, *a, **kw)
        e.record()
        marks.append((q.shape[0], s, e))
        return out
    real8 = hip.attn_fwd_fp8

    def timed_attn8(bufs, L, *a, **kw):                 # --sage: the MXFP8 kernel in the same place (its pack launch is not part of this figure)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = real8(bufs, L, *a, **kw)
        e.record()
        marks.append((kw["out"].shape[0] if kw.get("out") is not None else bufs[0].shape[0], s, e))
        return out
    hip.attn_fwd, hip.attn_fwd_fp8 = timed_attn, timed_attn8
    saved = os.environ.get("FLEXAM_REPLAY")
    os.environ["FLEXAM_REPLAY"] = "0"            # this one step goes through the Python wrappers (the events sit there); same launches, same stream
    try:
        pipe.denoise_step(step_index)
        torch.cuda.synchronize()
    finally:
        hip.attn_fwd, hip.attn_fwd_fp8 = real, real8
        if saved is None:
            os.environ.pop("FLEXAM_REPLAY", None)
        else:
            os.environ["FLEXAM_REPLAY"] = saved
    full = [s.elapsed_time(e) * 1e-3 for b, s, e in marks if b == B]
    every = [s.elapsed_time(e) * 1e-3 for b, s, e in marks]
    return {"sec": sum(full) / len(full), "calls": len(full), "calls_other_batch": len(marks) - len(full),
            "sec_all_calls": sum(every) / len(every)} if full else None


def kernel_rooflines(eng, B, L, lc):
    """Live per-launch timing of the hot kernels at this run's shapes, on the engine's own buffers."""
    from flexam_amd import hip
    d, f, nh, hd = eng.dim, eng.ffn, eng.nh, eng.hd
    ws = eng._workspace(B, lc)             # the step's own buffers (made here if the step ran another layout, e.g. the dual-stream mode)
    p = eng.blocks[0]
    qkv, ao, hbuf, ffn = ws["qkv"], ws["ao"], ws["h"], ws["ffn"]
    M = B * lc
    q4 = qkv.view(B, lc, 3 * d)[:, :, 0:d].unflatten(2, (nh, hd))
    out = {}
    if eng.sp_size > 1 and eng.sp_mode == "ulysses":
        # this rank's attention: all L tokens of nh / sp heads (the q|k|v it received in the last block's all-to-all)
        hg = nh // eng.sp_size
        full = ws["a2a_recv"].view(B, L, 3, hg, hd)
        t = time_kernel(lambda: hip.attn_fwd(full[:, :, 0], full[:, :, 1], full[:, :, 2], out=ws["a2a_out"], prescaled=True), iters=8)
        out["attn_self"] = dict(flops=4.0 * B * L * L * hg * hd, sec=t)
    else:
        if eng.sp_size == 1:
            k4 = qkv.view(B, lc, 3 * d)[:, :, d:2 * d].unflatten(2, (nh, hd))
            v4 = qkv.view(B, lc, 3 * d)[:, :, 2 * d:].unflatten(2, (nh, hd))
            t = time_kernel(lambda: hip.attn_fwd(q4, k4, v4, out=ao.view(B, lc, nh, hd), prescaled=True), iters=8)
        else:
            kv = ws["kv_cat"]                  # gathered K|V pieces of the last block [G, B, L, 2C/G] (same shape in every block)
            G = kv.shape[0]
            cb, hg = d // G, nh // G
            ao4 = ao.view(B, lc, nh, hd)

            def all_groups():
                for g in range(G):
                    hip.attn_fwd(q4[:, :, g * hg:(g + 1) * hg], kv[g, :, :, 0:cb].unflatten(2, (hg, hd)), kv[g, :, :, cb:].unflatten(2, (hg, hd)),
                                 out=ao4[:, :, g * hg:(g + 1) * hg], prescaled=True)
            t = time_kernel(all_groups, iters=8)
        out["attn_self"] = dict(flops=4.0 * B * lc * L * d, sec=t)
    t = time_kernel(lambda: hip.gemm(hbuf, p["wqkv"], p["bqkv"], out=qkv))
    out["gemm_qkv"] = dict(flops=2.0 * M * 3 * d * d, sec=t)
    t = time_kernel(lambda: hip.gemm(hbuf, p["w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH))
    out["gemm_ffn1_gelu"] = dict(flops=2.0 * M * f * d, sec=t)
    xs = torch.zeros(M, d, device=qkv.device, dtype=torch.float32)
    t = time_kernel(lambda: hip.gemm_gate_residual(ffn, p["w2"], p["b2"], xs))
    out["gemm_ffn2_residual"] = dict(flops=2.0 * M * d * f, sec=t)
    t = time_kernel(lambda: hip.gemm_gate_residual(ao, p["wo"], p["bo"], xs))
    out["gemm_oproj_residual"] = dict(flops=2.0 * M * d * d, sec=t)
    if getattr(eng, "fp8", False):
        w8 = eng._fp8_w[0]
        fused = d % 512 == 0 and d <= 4096            # the LN launch writes e4m3 + row scales + FFN1's output scales (DiTEngine._ln_fp8)
        if fused:
            a8, sa = hip.ln_modulate_fp8(ws["x"], ws["a8d"], ws["sa"], next_scale=ws["so"], next_wnorm=w8["w1_norm"], next_bias=w8["b1_max"])
        else:
            a8, sa = hip.quantize_rows_fp8(hbuf, ws["a8d"], ws["sa"])
        t = time_kernel(lambda: hip.gemm_fp8(a8, sa, w8["wqkv"], w8["s_wqkv"], p["bqkv"], out=qkv))
        out["gemm_fp8_qkv"] = dict(flops=2.0 * M * 3 * d * d, sec=t)
        if fused:                                      # FFN1 writes FFN2's e4m3 operand itself: no quantise pass in between
            t = time_kernel(lambda: hip.gemm_fp8_gelu_q(a8, sa, w8["w1"], w8["s_w1"], p["b1"], ws["so"], ws["a8"]))
            out["gemm_fp8_ffn1_gelu_e4m3_out"] = dict(flops=2.0 * M * f * d, sec=t)
            a8f, saf = ws["a8"], ws["so"]
            t = time_kernel(lambda: hip.ln_modulate_fp8(ws["x"], ws["a8d"], ws["sa"], next_scale=ws["so"], next_wnorm=w8["w1_norm"], next_bias=w8["b1_max"]))
            out["ln_modulate_fp8"] = dict(bytes=M * d * 5.0, sec=t)
        else:
            t = time_kernel(lambda: hip.gemm_fp8(a8, sa, w8["w1"], w8["s_w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH))
            out["gemm_fp8_ffn1_gelu"] = dict(flops=2.0 * M * f * d, sec=t)
            a8f, saf = hip.quantize_rows_fp8(ffn, ws["a8"], ws["sa"])
            t = time_kernel(lambda: hip.quantize_rows_fp8(ffn, ws["a8"], ws["sa"]))
            out["quantize_rows_fp8_ffn"] = dict(bytes=M * f * 3.0, sec=t)
        t = time_kernel(lambda: hip.gemm_fp8_gate_residual(a8f, saf, w8["w2"], w8["s_w2"], p["b2"], xs))
        out["gemm_fp8_ffn2_residual"] = dict(flops=2.0 * M * d * f, sec=t)
    for v in out.values():
        if "flops" in v:
            v["tflops"] = v["flops"] / v["sec"] / 1e12
    # bandwidth-bound kernels: ALGORITHMIC bytes / live time, against the 8 TB/s HBM3E peak
    T = torch.randn(4, 6, d, device=qkv.device)
    rows = (torch.arange(M, device=qkv.device) % 2).to(torch.int32)
    t = time_kernel(lambda: hip.ln_modulate(xs, out=hbuf, shift=T[:, 0], scale=T[:, 1], row_index=rows))
    out["ln_modulate"] = dict(bytes=M * d * 6.0, sec=t)                       # read fp32 x, write bf16
    cd = eng.cond
    if eng.sp_size == 1:
        t = time_kernel(lambda: hip.rmsnorm_rope(qkv[:, 0:d], p["nq"], qkv[:, d:2 * d], p["nk"], rope_cos=cd["cos"], rope_sin=cd["sin"],
                                                 tokens_per_batch=lc, head_dim=hd))
        out["rmsnorm_rope_qk"] = dict(bytes=M * d * 8.0, sec=t)               # q and k: read + write bf16
    return out


def sampler_step_roofline(pipe):
    """The fused CFG + Euler + blend launch on the clip's latents (26 MB algorithmic: two head-token rows, latents r/w, known, mask)."""
    from flexam_amd import hip
    st = pipe._state
    c, f, h, w = st["shape"]
    L = st["ref_len"] + f * (h // 2) * (w // 2)
    tok = torch.randn(2, L, 4 * c, device=st["latents"].device)
    lat = st["latents"].clone()
    t = time_kernel(lambda: hip.cfg_euler_blend(tok[0], tok[1], st["ref_len"], 6.0, -0.01, lat, st["known"], st["mask"]))
    n = c * f * h * w
    return dict(bytes=4.0 * (2 * n + 2 * n + n + n / c), sec=t)


# ----------------------------------------------------------------------------- the `roofline` / `kernels` objects of the JSON line
ATTN_SOURCES = ("flexam_amd/csrc/attn.hip", "flexam_amd/csrc/attn_fp8.inc", "flexam_amd/csrc/common.h")
ATTN_TRAFFIC_FILE = "profiles/head_attn_traffic.json"


# compile-time switches of DIAGNOSTIC builds (tools/build_attn_variants.py, -DFLEXAM_DIAGNOSTIC_BUILD): never defined in the product
# library, so the text they guard is not part of what the counter record was measured on
ATTN_DIAGNOSTIC_MACROS = ("FLEXAM_ATTN_STAMPS", "A32_NOMAX_ABLATE", "A32_VALU", "FLEXAM_ATTN_BODY16", "A32_RESCALE_THR", "A32_HALF_FRAG_ABLATE", "FLEXAM_DIAGNOSTIC_BUILD")


def product_text(src: str, undefined=ATTN_DIAGNOSTIC_MACROS) -> str:
    """The source text the PRODUCT build compiles: `#ifdef X ... [#else ...] #endif` / `#ifndef X` / `#if defined(X) ...` blocks of the
    diagnostic macros resolved as "X is not defined", comments and blank lines dropped, runs of white space folded -- so that adding a
    diagnostic switch or editing a comment does not orphan a committed counter record, while any change to compiled code does."""
    import re
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out, stack = [], []                # stack of [kind, keep_now, parent_keep]; kind "diag" = a block of ours, "other" = any other conditional
    for line in src.split("\n"):
        code = re.sub(r"//.*$", "", line).rstrip()
        st = code.strip()
        keep = all(k for _, k, _ in stack)
        m = re.match(r"#\s*(ifdef|ifndef)\s+(\w+)", st)
        m2 = re.match(r"#\s*if\b(.*)", st)
        if m and m.group(2) in undefined:
            stack.append(["diag", m.group(1) == "ifndef", keep])
            continue
        if m2 and not m and any(re.search(r"defined\s*\(?\s*" + u + r"\b", m2.group(1)) for u in undefined):
            stack.append(["diag", False, keep])      # `#if defined(DIAG) ...` guards (the #error check): false in the product
            continue
        if m or m2:
            stack.append(["other", True, keep])
        elif re.match(r"#\s*else\b", st) and stack and stack[-1][0] == "diag":
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r"#\s*endif\b", st) and stack:
            kind = stack.pop()[0]
            if kind == "diag":
                continue
        if keep and st:
            out.append(re.sub(r"\s+", " ", st))
    return "\n".join(out)


def attn_source_sha(root):
    """sha256 (16 hex digits) over the PRODUCT text (product_text) of the attention kernel's sources: what a committed counter record
    must have been taken from."""
    import hashlib
    h = hashlib.sha256()
    for rel in ATTN_SOURCES:
        with open(os.path.join(root, rel), "r") as f:
            h.update(product_text(f.read()).encode())
    return h.hexdigest()[:16]


def attn_traffic(root, shape):
    """HBM bytes per self-attention launch from the counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, the
    guide's gfx950 correction; tools/pmc_table.py --attn-traffic writes the record in the same gpurun call as the kernel trace) -- only
    when the record was taken from THIS tree's attention sources at THIS shape; anything else is null, never an older round's number."""
    path = os.path.join(root, ATTN_TRAFFIC_FILE)
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, "no counter record in this tree (" + ATTN_TRAFFIC_FILE + ")"
    if rec.get("source_sha16") != attn_source_sha(root):
        return None, f"{ATTN_TRAFFIC_FILE} was taken from other attention sources ({rec.get('source_sha16')}): not quoted"
    if list(rec.get("shape", [])) != list(shape):
        return None, f"{ATTN_TRAFFIC_FILE} is for shape {rec.get('shape')}, this run is {list(shape)}: not quoted"
    return rec["hbm_bytes_per_launch"], ("bytes/launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over the launches of one call, taken from this tree's "
                                         f"attention sources (sha {rec['source_sha16']}, {rec.get('record', ATTN_TRAFFIC_FILE)})")


def roofline_object(root, kern, attn_in_step, shape, world, sage_taken):
    a = kern["attn_self"]
    traffic, note = (attn_traffic(root, shape) if world == 1 else (None, "single-GPU record only"))
    sec_live = attn_in_step["sec"] if attn_in_step else a["sec"]
    r = {"bound": "mfma", "kernel": "attn_fwd_kernel<0, true, true> (self-attention, head_dim 128, q pre-scaled by its RMSNorm weight; the one-basic-block-step instance)",
         "achieved": a["flops"] / sec_live / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": a["flops"] / sec_live / 1e12 / PEAK_BF16_TFLOPS,
         "traffic": traffic, "traffic_note": note,
         "launch_ms": sec_live * 1e3, "flops_per_launch": a["flops"],
         "timed": ("IN the step: HIP events around the %d whole-batch self-attention calls of one denoise step (the other %d call(s) run one "
                   "sample: block 0 shared by the CFG pair); this is what rocprofv3 --kernel-trace --stats of the same command averages for "
                   "the kernel (+ the 20 us merge)" % (attn_in_step["calls"], attn_in_step["calls_other_batch"])) if attn_in_step
                  else "isolated back-to-back launches (no in-step timing in this run)",
         "launch_ms_all_calls": attn_in_step["sec_all_calls"] * 1e3 if attn_in_step else None,
         "launch_ms_all_calls_note": "mean over ALL self-attention calls of that step (the one-sample call of block 0 included) + the merge: "
                                     "compare with the kernel's AverageNs in rocprofv3 --kernel-trace --stats of `bench.py --no-kernel-timing`",
         "isolated_launch_ms": a["sec"] * 1e3, "isolated_frac": a["tflops"] / PEAK_BF16_TFLOPS,
         "isolated_note": "8 back-to-back launches on the step's own buffers: runs at the clock the kernel holds alone, not the step's",
         "launch_note": "one self-attention call = ONE attn_fwd_kernel<0, true, true> launch (the full rounds of work units and, on the same "
                        "XCDs behind them, the last partial round with its keys cut in 3) + attn_merge_kernel; launch_ms is the whole "
                        "call = its AverageNs in rocprofv3 + the merge"}
    if sage_taken and attn_in_step:               # the dominant kernel of THIS line is the MXFP8 one: priced against the fp8 pipe
        r.update(kernel="attn8_fwd_kernel<0> (self-attention on MXFP8 operands, csrc/attn_fp8.inc)", peak=PEAK_FP8_TFLOPS,
                 frac=r["achieved"] / PEAK_FP8_TFLOPS, traffic=None, traffic_note="not collected for the MXFP8 kernel", isolated_launch_ms=None,
                 isolated_frac=None, isolated_note="not timed alone in this run",
                 launch_note="one call = ONE attn8_fwd_kernel launch + attn_merge_kernel; the attn8_pack_kernel launch in front of it (0.12 ms) is not part of launch_ms")
    return r


def kernels_object(kern):
    return {k: ({"ms": round(v["sec"] * 1e3, 4), "tflops": round(v["tflops"], 1), "bound": "mfma",
                 "peak": PEAK_FP8_TFLOPS if k.startswith("gemm_fp8") else PEAK_BF16_TFLOPS,
                 "frac": round(v["tflops"] / (PEAK_FP8_TFLOPS if k.startswith("gemm_fp8") else PEAK_BF16_TFLOPS), 4)} if "flops" in v else
                {"ms": round(v["sec"] * 1e3, 4), "gbs": round(v["bytes"] / v["sec"] / 1e9, 1), "bound": "hbm",
                 "frac": round(v["bytes"] / v["sec"] / 1e9 / PEAK_HBM_GBS, 4)}) for k, v in kern.items()}


def newest_profile(root, suffix):
    """Name of the newest committed record under profiles/ that ends in `suffix` (by round letter order), or None."""
    import glob
    import re
    cands = [os.path.basename(p) for p in glob.glob(os.path.join(root, "profiles", "r*_" + suffix))]
    key = lambda n: (int(re.match(r"r(\d+)", n).group(1)), len(re.match(r"r\d+([a-z]*)", n).group(1)), n)
    return max(cands, key=key) if cands else None
