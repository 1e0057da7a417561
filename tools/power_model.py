"""Dev tool: the energy budget of one denoise step (round-4 verdict item 4: "a power model that lands within 2 % of the measured step and
names the floor").  Every hot kernel of a DiT block is looped ALONE for a few seconds at the config-2 shapes while hwmon socket power (PPT)
and sclk are sampled at 50 Hz; with the per-step launch counts this gives joules per step by kernel class, the step the socket power cap
allows for that energy, and what the same step would cost if the only energy were the matrix pipe's (pure-MFMA probe figure, r3f).
usage: power_model.py [seconds-per-kernel] [measured_ms_per_step]   -> text table + JSON line"""
import glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import kernel_driver as KD

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
MEASURED_MS = float(sys.argv[2]) if len(sys.argv) > 2 else None
L, B, d, f, T = KD.L, KD.B, KD.d, KD.f, KD.T
M = B * L
# (kernel_driver name, launches per step, algorithmic FLOPs per launch, algorithmic HBM bytes per launch)
PER_STEP = [
    ("attn_self", 29.5, 4.0 * B * L * L * d, 4 * M * d * 2),                                   # block 0 runs one sample (share0): 29 + 0.5
    ("gemm_qkv", 29.5, 2.0 * M * 3 * d * d, (M * d + 3 * d * d + M * 3 * d) * 2),
    ("gemm_oproj_residual", 59.5, 2.0 * M * d * d, (M * d + d * d) * 2 + M * d * 8),              # o-proj + cross-o
    ("gemm_crossq", 30, 2.0 * M * d * d, (2 * M * d + d * d) * 2),
    ("attn_cross", 30, 4.0 * B * L * 127 * d, 2 * M * d * 2),                                     # 127 keys after the padded-row fold (driver runs 512: scaled below)
    ("gemm_ffn1_gelu", 30, 2.0 * M * f * d, (M * d + f * d + M * f) * 2),
    ("gemm_ffn2_residual", 30, 2.0 * M * d * f, (M * f + f * d) * 2 + M * d * 8),
    ("ln_modulate", 59.5, 0.0, M * d * 6),
    ("ln_affine", 30, 0.0, M * d * 6),
    ("rmsnorm_rope_qk", 29.5, 0.0, M * d * 8),
    ("rmsnorm_q", 30, 0.0, M * d * 4),
]


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
cap = None
for dd in hw:
    v = read(os.path.join(dd, "power1_cap"))
    if v:
        cap = max(cap or 0.0, float(v) / 1e6)


def sample(stop, rows):
    while not stop.is_set():
        rows.append([(read(os.path.join(dd, "power1_input")) or read(os.path.join(dd, "power1_average")), read(os.path.join(dd, "freq1_input"))) for dd in hw])
        time.sleep(0.02)


def loop(fn, secs):
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, rows)); th.start()
    t0 = time.perf_counter(); n = 0
    if fn is None:
        time.sleep(secs)
    else:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        s.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(10):
                fn()
            e.record(); torch.cuda.synchronize(); n += 10
    stop.set(); th.join()
    tail = rows[len(rows) // 2:]

    def avg(c, i, scale):
        vals = [float(r[c][i]) / scale for r in tail if r[c][i] not in (None, "")]
        return sum(vals) / len(vals) if vals else float("nan")
    c = max(range(len(hw)), key=lambda j: avg(j, 0, 1e6)) if hw else 0
    us = (s.elapsed_time(e) * 1e3 / n) if fn is not None and n else float("nan")
    return us, (avg(c, 0, 1e6) if hw else float("nan")), (avg(c, 1, 1e6) if hw else float("nan"))


K = KD.build_kernels()
for fn in K.values():            # every buffer a later kernel reads holds what the step would put there (an unwritten A operand = zeros = a GEMM at 1700 TF/s)
    fn()
torch.cuda.synchronize()
_, p_idle, f_idle = loop(None, 2.0)
print(f"hwmon dirs: {len(hw)}, power cap {cap} W, idle {p_idle:.0f} W at {f_idle:.0f} MHz")
table, e_step, t_sum = [], 0.0, 0.0
for name, count, flops, nbytes in PER_STEP:
    us, watts, mhz = loop(K[name], SECS)
    if name == "attn_cross":
        us *= 127.0 / 512.0                      # the driver attends to all 512 text rows; the step to 127 (a fair share of a launch-bound 0.1 ms kernel)
    joules = watts * us * 1e-6
    e_step += joules * count
    t_sum += us * 1e-3 * count
    table.append(dict(kernel=name, per_step=count, us_alone=round(us, 1), watts=round(watts, 1), mhz=round(mhz), joules_per_launch=round(joules, 4),
                      joules_per_step=round(joules * count, 2), tflops=round(flops / us / 1e6, 1) if flops else None,
                      gbs=round(nbytes / us / 1e3, 1), pj_per_flop=round(joules / flops * 1e12, 3) if flops else None))
    print(f"{name:22s} x{count:5.1f}  {us:8.1f} us  {watts:7.1f} W  {mhz:6.0f} MHz  {joules:7.4f} J/launch  {joules * count:7.2f} J/step"
          + (f"  {flops / us / 1e6:7.1f} TF/s  {joules / flops * 1e12:.3f} pJ/FLOP" if flops else f"  {nbytes / us / 1e3:7.1f} GB/s  {(watts - p_idle) * us * 1e-6 / nbytes * 1e12:.1f} pJ/B above idle"), flush=True)
cap_w = cap or 1400.0
pred_ms = e_step / cap_w * 1e3
print(f"\nsum of the kernels' isolated times: {t_sum:.1f} ms per step; energy {e_step:.1f} J per step; at the {cap_w:.0f} W cap that energy takes {pred_ms:.1f} ms"
      + (f"; measured step {MEASURED_MS:.1f} ms -> model / measured = {pred_ms / MEASURED_MS:.3f} (energy), {t_sum / MEASURED_MS:.3f} (sum of isolated times)" if MEASURED_MS else ""))
flops_step = sum(c * fl for _, c, fl, _ in PER_STEP)
# the matrix pipe alone (tools/mfma_toggle_probe.py, profiles/r3f): random bf16 operands, nothing but MFMAs, on the same cap
E_MFMA = {"16x16x32": (cap_w - p_idle) / 1.95e15, "32x32x16": (cap_w - p_idle) / 1.835e15}
e_gemm = sum(c * fl for n, c, fl, _ in PER_STEP if n.startswith("gemm")) * E_MFMA["16x16x32"]
e_attn = sum(c * fl for n, c, fl, _ in PER_STEP if n.startswith("attn")) * E_MFMA["32x32x16"]
bw_j = sum(r["joules_per_step"] for r in table if r["tflops"] is None)
bw_ms = sum(r["us_alone"] * r["per_step"] for r in table if r["tflops"] is None) * 1e-3
floor_ms = (e_gemm + e_attn) / (cap_w - p_idle) * 1e3 + bw_ms
print(f"matrix pipe alone on this data (r3f: 1950 / 1835 TF/s at the cap): {flops_step / 1e12:.1f} TF per step -> {(e_gemm + e_attn):.1f} J above idle = "
      f"{(e_gemm + e_attn) / (cap_w - p_idle) * 1e3:.1f} ms; + the bandwidth kernels as they are ({bw_ms:.1f} ms, {bw_j:.1f} J) -> floor {floor_ms:.1f} ms per step "
      f"for THIS instruction mix on THIS data under THIS cap" + (f" = {floor_ms / MEASURED_MS:.3f} of the measured step" if MEASURED_MS else ""))
print(json.dumps(dict(cap_w=cap_w, idle_w=round(p_idle, 1), table=table, energy_j_per_step=round(e_step, 2), isolated_sum_ms=round(t_sum, 2),
                      energy_at_cap_ms=round(pred_ms, 2), measured_ms=MEASURED_MS, mfma_only_floor_ms=round(floor_ms, 2))))
