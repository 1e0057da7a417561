"""Dev tool: IN-STEP same-process A/B.  Builds bench.py's pipeline once (97x512x896, 30 layers, CFG pair) and alternates groups of
denoise steps between arms; an arm is VAR=value (an environment switch the library reads per call) or lib=NAME
(tools/probes/libflexam_var_NAME.so, `tree` = the in-tree library; the kernels are stateless, so the library can be swapped between
steps).  usage: ab_step.py ARM ARM ... [--steps=3] [--rounds=5] [--fp8=1] [--logit=6]; prints the median ms per step of every arm."""
import os, sys, time, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench
from flexam_amd import hip as H
from flexam_amd import Wan2_2FunControlPipeline_FlexAM
from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
from flexam_amd.configs import WAN22_FUN_5B_FLEXAM

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--")}
steps, rounds = opt.get("steps", 3), opt.get("rounds", 5)
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = dict(WAN22_FUN_5B_FLEXAM)
model = bench.build_model(cfg, dev)
if opt.get("logit", 0):
    bench.set_logit_scale(model, float(opt["logit"]))          # --logit=6: peaked softmax rows (bench.py --logit-scale), for every arm
if opt.get("fp8", 0):
    model.enable_fp8_gemm(True)                 # --fp8=1: the QKV / FFN GEMMs on the fp8 pipe (BASELINE configs[4] variant) for every arm
pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model)
i = bench.synthetic_inputs(97, 512, 896, cfg["text_dim"], "motion")
cond = LatentConditioning(control_latents=i["control"], additional_control=i["additional"], masked_video_latents=i["masked"],
                          ref_latents=i["ref"], mask_latents=i["mask_latents"], mask=i["mask"], mask_pixels=i["mask_pixels"])
pipe.prepare(i["latents"], cond, i["ctx_c"], i["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=50)


ALL_VARS = sorted({kv.partition("=")[0] for a in args for kv in a.split(",") if not kv.startswith("lib=")})


def select(arm):
    """arm = comma-separated settings: lib=NAME and / or VAR=value; variables another arm sets are cleared, the library defaults to the tree's"""
    for k in ALL_VARS:
        os.environ.pop(k, None)
    libname = "tree"
    for kv in arm.split(","):
        k, _, v = kv.partition("=")
        if k == "lib":
            libname = v
        else:
            os.environ[k] = v
    H.load_library(H.LIB_PATH if libname == "tree" else os.path.join(root, "tools", "probes", f"libflexam_var_{libname}.so"))


import glob, threading
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")


def _read(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return float("nan")


def sampler(stop, rows):          # socket power and shader clock of every card the box shows (only one runs)
    while not stop.is_set():
        rows.append([(_read(os.path.join(d, "power1_input")), _read(os.path.join(d, "freq1_input"))) for d in hw])
        time.sleep(0.02)


res = {a: [] for a in args}
tele = {a: [] for a in args}
n = 0
for r in range(rounds + 1):                      # round 0 = warm-up, dropped
    for a in (args if r % 2 == 0 else args[::-1]):
        select(a)
        pipe.denoise_step(n % 50); n += 1        # one untimed step on this arm
        torch.cuda.synchronize()
        rows, stop = [], threading.Event()
        th = threading.Thread(target=sampler, args=(stop, rows)); th.start()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.denoise_step(n % 50); n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        stop.set(); th.join()
        if r:
            res[a].append(el / steps * 1e3)
            if rows and hw:
                c = max(range(len(hw)), key=lambda j: sum(x[j][0] for x in rows))
                tele[a].append((sum(x[c][0] for x in rows) / len(rows) / 1e6, sum(x[c][1] for x in rows) / len(rows) / 1e6))
base = statistics.median(res[args[0]])
for a in args:
    m = statistics.median(res[a])
    pw = statistics.median(t[0] for t in tele[a]) if tele[a] else float("nan")
    ck = statistics.median(t[1] for t in tele[a]) if tele[a] else float("nan")
    print(f"{a:28s} {m:8.2f} ms/step  ({100 * (m / base - 1):+.2f} %)   min {min(res[a]):.2f} max {max(res[a]):.2f}   socket {pw:6.0f} W  sclk {ck:5.0f} MHz  -> {pw * m * 1e-3:6.1f} J/step", flush=True)
