"""Dev tool: vae_prep (RMS-norm + SiLU + bf16 pack): the span form (<= 256 channels, r5) against the wave-per-position form
(FLEXAM_VAE_PREP_WAVE=1) of the tree, and against diagnostic builds tools/probes/libflexam_hip_prep_pix_<a>_<b>.so (the tree's vae.hip compiled with -DFLEXAM_PREP_PIX1=a
-DFLEXAM_PREP_PIX2=b), round-robin in one process at the VAE's shapes.  usage: ab_prep_pix.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flexam_amd import hip as H

import glob
libs = {"tree": H.lib()}
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "probes", "libflexam_hip_prep_pix_*_*.so"))):
    a, b = os.path.basename(path)[:-3].split("_")[-2:]
    libs[f"{a}/{b}"] = ctypes.CDLL(path)
for l in libs.values():
    l.flexam_vae_prep_cl.restype = ctypes.c_int
    l.flexam_vae_prep_cl.argtypes = H._SIGNATURES["flexam_vae_prep_cl"][0]
# arms: (label, library, FLEXAM_VAE_PREP_WAVE): the tree's span form (<= 256 channels) against its wave-per-position form, + any diagnostic build
arms = [("span", libs["tree"], None), ("wave", libs["tree"], "1")] + [(k, l, "1") for k, l in libs.items() if k != "tree"]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
# (name, frames, h, w, C, src dtype): encoder stage 0 / 1, decoder stages 3 / 2 / low resolution
SHAPES = [("enc 256x448 160ch fp32", 4, 256, 448, 160, torch.float32), ("enc 256x448 160ch bf16", 4, 256, 448, 160, torch.bfloat16),
          ("enc 128x224 320ch fp32", 4, 128, 224, 320, torch.float32), ("dec 256x448 256ch fp32", 4, 256, 448, 256, torch.float32),
          ("dec 256x448 256ch bf16", 4, 256, 448, 256, torch.bfloat16), ("dec 128x224 512ch fp32", 4, 128, 224, 512, torch.float32),
          ("dec 64x112 1024ch fp32", 2, 64, 112, 1024, torch.float32)]
for name, t, h, w, c, dt in SHAPES:
    rows = t * (h + 2) * (w + 2)
    src = torch.randn(rows, c, device=dev).to(dt)
    gamma = torch.ones(c, device=dev)
    cp = (c + 63) // 64 * 64 if c % 64 == 0 else (c + 7) // 8 * 8
    dst = torch.zeros(t, h + 2, w + 2, cp, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(5):
        for k, l, wave in arms:
            os.environ.pop("FLEXAM_VAE_PREP_WAVE", None)
            if wave:
                os.environ["FLEXAM_VAE_PREP_WAVE"] = wave
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                l.flexam_vae_prep_cl(src.data_ptr(), 1 if dt == torch.bfloat16 else 0, c, c, t, h, w, gamma.data_ptr(), 2, dst.data_ptr(), cp, 0, 0, st)
            e.record(); torch.cuda.synchronize()
            res.setdefault(k, []).append(s.elapsed_time(e) * 100.0)
    nbytes = t * h * w * c * (src.element_size() + 2)
    line = "  ".join(f"{k}: {sorted(v)[len(v) // 2]:6.1f} us ({nbytes / sorted(v)[len(v) // 2] / 1e6:4.2f})" for k, v in res.items())
    print(f"{name:26s} {line}", flush=True)
