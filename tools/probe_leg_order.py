"""Dev probe (r6u): the emulated overlapped K|V gather of a rank of 8 (cfg2 x sp4, FLEXAM_SP_OVERLAP=1) in ONE process, legs with the links modelled at
the given rates in the given order (`none` = no link model) -- found that a fresh side stream per loopback group aliased the compute stream's hardware
queue every few groups (two timing regimes).  usage: probe_leg_order.py none 50 75 75 50 75   [FLEXAM_CU_BUDGET=248 to leave 8 CUs free]"""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import torch
from benchlib.emulate import layouts, set_emulated_layout, measure_rank_step
from benchlib.inputs import build_model, synthetic_inputs
from flexam_amd import Wan2_2FunControlPipeline_FlexAM
from flexam_amd.configs import WAN22_FUN_5B_FLEXAM
from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
cfg = dict(WAN22_FUN_5B_FLEXAM)
model = build_model(cfg, dev)
i = synthetic_inputs(97, 512, 896, cfg["text_dim"], "motion")
cond = LatentConditioning(control_latents=i["control"], additional_control=i["additional"], masked_video_latents=i["masked"], ref_latents=i["ref"],
                          mask_latents=i["mask_latents"], mask=i["mask"], mask_pixels=i["mask_pixels"])
name, mode, cfgp, pieces, overlap = layouts(8, model.num_heads)[3]
os.environ["FLEXAM_SP_MODE"] = mode; os.environ["FLEXAM_SP_OVERLAP"] = overlap
for rate in [float(x) if x != "none" else None for x in sys.argv[1:]]:
    set_emulated_layout(model, 8, cfgp, 2, link_gbps=rate)
    pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model)
    pipe.prepare(i["latents"], cond, i["ctx_c"], i["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=50)
    m = measure_rank_step(lambda k: pipe.denoise_step(k % 50), torch.cuda.synchronize, 8, 2)
    print(f"rate {rate}: {m['sec'] * 1e3:.2f} ms per step, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
    del pipe
from flexam_amd import hip
print("num_cus at the end:", hip.num_cus(), "FLEXAM_CU_BUDGET", os.environ.get("FLEXAM_CU_BUDGET"))
