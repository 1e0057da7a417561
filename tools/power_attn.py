"""Dev tool: socket power and shader clock (sysfs hwmon) while ONE arm of the self-attention call loops for 3 s; arms as in
tools/ab_step.py (VAR=value or lib=NAME).  Energy per call = power x time."""
import glob, os, sys, threading, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")


def _read(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return float("nan")


def sampler(stop, rows):
    while not stop.is_set():
        rows.append([(_read(os.path.join(d, "power1_input")), _read(os.path.join(d, "freq1_input"))) for d in hw])
        time.sleep(0.02)


g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
for rnd in range(2):
    for arm in sys.argv[1:]:
        for a2 in sys.argv[1:]:
            for kv in a2.split(","):
                if not kv.startswith("lib="):
                    os.environ.pop(kv.partition("=")[0], None)
        libname = "tree"
        for kv in arm.split(","):
            k_, _, v_ = kv.partition("=")
            if k_ == "lib":
                libname = v_
            else:
                os.environ[k_] = v_
        H.load_library(H.LIB_PATH if libname == "tree" else os.path.join(root, "tools", "probes", f"libflexam_var_{libname}.so"))
        for _ in range(200):
            H.attn_fwd(q, k, v, out=o, prescaled=True)
        torch.cuda.synchronize()
        rows, stop = [], threading.Event()
        th = threading.Thread(target=sampler, args=(stop, rows)); th.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(600):
            H.attn_fwd(q, k, v, out=o, prescaled=True)
        e.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        t = s.elapsed_time(e) / 600 * 1e3
        c = max(range(len(hw)), key=lambda j: sum(x[j][0] for x in rows)) if hw else 0
        pw = sum(x[c][0] for x in rows) / len(rows) / 1e6 if hw else float("nan")
        ck = sum(x[c][1] for x in rows) / len(rows) / 1e6 if hw else float("nan")
        print(f"{arm:28s} {t:8.1f} us/call   socket {pw:6.0f} W   sclk {ck:5.0f} MHz   {pw * t * 1e-6:6.3f} J/call  ({len(rows)} samples)", flush=True)
