"""Dev tool: ONE emulated rank of an N-GPU denoise step (benchlib/emulate.py: the real engine on that rank's token chunk / CFG row, collectives =
same-size device copies on a side stream), alone in a process -- the thing to wrap in `rocprofv3 --kernel-trace --stats` for the per-kernel
table of a rank's step.  usage: emulate_rank.py <world> <layout index: 0 = the default (cfg2 x spN/2, K|V all-gather), 1 = the next of
benchlib.emulate.layouts()> [steps] [warmup]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchlib.emulate import layouts, set_emulated_layout
from benchlib.inputs import build_model, synthetic_inputs
from flexam_amd import Wan2_2FunControlPipeline_FlexAM
from flexam_amd.configs import WAN22_FUN_5B_FLEXAM
from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import LatentConditioning

world, which = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("FLEXAM_CU_BUDGET"):                  # plan the persistent grids for fewer CUs (the rest stay free for the collectives' kernels)
    from flexam_amd import hip
    hip.set_cu_budget(int(os.environ["FLEXAM_CU_BUDGET"]))
cfg = dict(WAN22_FUN_5B_FLEXAM)
model = build_model(cfg, dev)
name, mode, cfgp, pieces, overlap = layouts(world, model.num_heads)[which]
os.environ["FLEXAM_SP_MODE"] = mode
if overlap is not None:
    os.environ["FLEXAM_SP_OVERLAP"] = str(overlap)
sp = world // 2 if cfgp else world
rank = int(os.environ.get("FLEXAM_EMULATE_WHICH", sp // 2 if sp > 2 else 0))
link = float(os.environ["FLEXAM_EMULATE_LINK_GBPS"]) if os.environ.get("FLEXAM_EMULATE_LINK_GBPS") else None      # assumed GB/s per link and direction
set_emulated_layout(model, world, cfgp, rank, link_gbps=link)
i = synthetic_inputs(97, 512, 896, cfg["text_dim"], "motion")
cond = LatentConditioning(control_latents=i["control"], additional_control=i["additional"], masked_video_latents=i["masked"], ref_latents=i["ref"],
                          mask_latents=i["mask_latents"], mask=i["mask"], mask_pixels=i["mask_pixels"])
pipe = Wan2_2FunControlPipeline_FlexAM(transformer=model)
pipe.prepare(i["latents"], cond, i["ctx_c"], i["ctx_u"], density=0.1, guidance_scale=6.0, num_inference_steps=50)
for k in range(warm):
    pipe.denoise_step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    pipe.denoise_step(warm + k)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / steps
eng = model.engine()
print((f"[links modelled at {link:g} GB/s per direction] " if link else "") + f"{name} [replayed launches: {bool(getattr(eng, 'replay_taken', False))}]: rank {rank} of sp{eng.sp_size} (cfg{eng.cfg_size}), {eng.cond['L'] // eng.sp_size} tokens x {1 if eng.cfg_size == 2 else 2} sample(s): "
      f"{sec * 1e3:.2f} ms per step, host enqueue {t_enq / steps * 1e3:.2f} ms")
