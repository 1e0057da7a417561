"""Dev tool: socket power / clocks (sysfs hwmon + rocm-smi, whatever the box lets an ordinary user read) sampled while ONE hot kernel runs in a
loop: pure-MFMA probe, self-attention, QKV GEMM, LayerNorm.  Says whether a kernel's clock is the power cap's or something else's."""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


hw = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")]
print("hwmon dirs:", hw)
for d in hw[:1]:
    for f in sorted(os.listdir(d)):
        v = read(os.path.join(d, f))
        if v is not None and len(v) < 40:
            print("  ", f, v)
try:
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True, timeout=30).stdout
    print(out[-2500:])
except Exception as e:
    print("rocm-smi:", e)


def sample(stop, rows):
    # the box shows all of the node's cards in sysfs but only one to HIP: sample every card, report the busiest
    while not stop.is_set():
        rows.append([(read(os.path.join(d, "power1_input")) or read(os.path.join(d, "power1_average")), read(os.path.join(d, "freq1_input")),
                      read(os.path.join(d, "temp2_input"))) for d in hw])
        time.sleep(0.05)


g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
x = qkv.view(2 * L, 3 * d)[:, :d].contiguous()
w = (torch.randn(3 * d, d, generator=g) * 0.05).to(BF).to(dev)
out = torch.empty(2 * L, 3 * d, dtype=BF, device=dev)
xf = torch.randn(2 * L, d, device=dev)
h = torch.empty(2 * L, d, dtype=BF, device=dev)
z = torch.zeros_like(x); wz = torch.zeros_like(w)
cases = {"idle": None,
         "self-attention": lambda: H.attn_fwd(q, k, v, out=o, prescaled=True),
         "qkv gemm": lambda: H.gemm(x, w, None, out=out),
         "qkv gemm, zero operands": lambda: H.gemm(z, wz, None, out=out),
         "ln_modulate": lambda: H.ln_modulate(xf, out=h)}
for name, fn in cases.items():
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, rows)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        if fn is None:
            time.sleep(0.1)
        else:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); n += 10
    el = time.perf_counter() - t0
    stop.set(); th.join()
    tail = rows[len(rows) // 2:]
    def avg(c, i, scale):
        vals = [float(r[c][i]) / scale for r in tail if r[c][i] not in (None, "")]
        return sum(vals) / len(vals) if vals else float("nan")
    c = max(range(len(hw)), key=lambda j: avg(j, 0, 1e6)) if hw else 0
    print(f"{name:26s} {el / max(n, 1) * 1e6:9.1f} us/call   busiest card {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(hw[c]))))}: power {avg(c, 0, 1e6):7.1f} W   "
          f"sclk {avg(c, 1, 1e6):7.1f} MHz   junction {avg(c, 2, 1e3):5.1f} C   ({len(tail)} samples; all cards W: {[round(avg(j, 0, 1e6)) for j in range(len(hw))]})", flush=True)
