"""Dev tool: ENERGY-side A/B of attention builds / switches.  The chip is power-capped and its power controller is slow: arms that
alternate every few launches share one clock, so a short round-robin ranks CYCLES; inside a denoise step what counts is the energy a
call needs.  Here every arm runs alone for `--dwell` seconds (default 1.5: hundreds of launches back to back) before it is timed
over the last `--timed` seconds; arms alternate over `--rounds` rounds.  Arms as in tools/ab_step.py: VAR=value or lib=NAME.
Also prints the short round-robin (5 launches per arm) for the same arms."""
import os, sys, time, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from flexam_amd import hip as H
args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--")}
dwell, timed, rounds = opt.get("dwell", 1.5), opt.get("timed", 1.0), int(opt.get("rounds", 3))
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
out = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
fl = 4.0 * 2 * 24 * L * L * 128


ALL_VARS = sorted({kv.partition("=")[0] for a in args for kv in a.split(",") if not kv.startswith("lib=")})


def select(arm):
    for k_ in ALL_VARS:
        os.environ.pop(k_, None)
    libname = "tree"
    for kv in arm.split(","):
        k_, _, v_ = kv.partition("=")
        if k_ == "lib":
            libname = v_
        else:
            os.environ[k_] = v_
    H.load_library(H.LIB_PATH if libname == "tree" else os.path.join(root, "tools", "probes", f"libflexam_var_{libname}.so"))


def run(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        H.attn_fwd(q, k, v, out=out, prescaled=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / n


res, short = {a: [] for a in args}, {a: [] for a in args}
for r in range(rounds):
    for a in (args if r % 2 == 0 else args[::-1]):
        select(a)
        t1 = run(3)
        run(max(1, int((dwell - timed) / t1)))
        res[a].append(run(max(1, int(timed / t1))))
for r in range(7):
    for a in (args if r % 2 == 0 else args[::-1]):
        select(a)
        run(1)
        short[a].append(run(5))
b0, s0 = statistics.median(res[args[0]]), statistics.median(short[args[0]])
for a in args:
    m, s = statistics.median(res[a]), statistics.median(short[a])
    print(f"{a:28s} dwell {m * 1e6:8.1f} us {fl / m / 1e12:6.0f} TF/s ({100 * (b0 / m - 1):+5.1f} %)   short round-robin {s * 1e6:8.1f} us ({100 * (s0 / s - 1):+5.1f} %)", flush=True)
