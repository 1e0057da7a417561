"""Dev tool: how many host threads the fp32 oracle wants on the GPU box (128 cores): small problems run SLOWER on all cores.  Decides the
thread counts tests/conftest.py pins for the parity tests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, dit as O, sampler as S

print("cores", os.cpu_count(), "default threads", torch.get_num_threads())
tiny = dict(O.DIT_TINY, num_layers=30)
sd = C.dit_weights(tiny, 29)
sc = C.sampler_case(tiny)
ml, mask, pinned = S.prepare_masks(sc["mask_pixels"], sc["latents"])
for nt in (4, 8, 16, 32, 128):
    torch.set_num_threads(nt)
    t = time.perf_counter()
    with torch.no_grad():
        S.denoise_loop(lambda **k: O.dit_forward(sd, tiny, **k), S.FlowMatchEulerSchedule(1000, 5.0), 3, sc["latents"], sc["context_uncond"], sc["context_cond"],
                       sc["control_latents"], sc["additional_control"], ml, sc["masked_video_latents"], sc["ref_latents"], mask, pinned, 0.1, 6.0)
    print(f"width 256, 30 layers, 3 steps: {nt:4d} threads {time.perf_counter() - t:6.2f} s", flush=True)
big = dict(O.DIT_5B, num_layers=1)
sd = C.dit_weights(big, 5)
for name, kw in (("L=256 B=2", dict(frames=3, h=16, w=16)), ("L=2912 B=2", dict(frames=25, h=16, w=28))):
    case = C.dit_case(big, 16, batch=2, text_lens=(77, 126), **kw)
    for nt in (16, 32, 64, 128):
        torch.set_num_threads(nt)
        t = time.perf_counter()
        with torch.no_grad():
            O.dit_forward(sd, big, **case)
        print(f"d=3072 one layer {name}: {nt:4d} threads {time.perf_counter() - t:6.2f} s", flush=True)
# bench.py's cpu_baseline leg: ONE block at L = 11648, one sample
import bench  # noqa: E402  (benchlib on the path)
from benchlib.cpu_baseline import cpu_baseline
for nt in (32, 64, 128):
    if nt > (os.cpu_count() or 8):
        continue
    torch.set_num_threads(nt)
    r = cpu_baseline(11648, dict(O.DIT_5B))
    print(f"cpu_baseline block L=11648: {nt:4d} threads {r['block_seconds']:6.2f} s", flush=True)
