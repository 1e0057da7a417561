"""Dev tool: the conditioning rasteriser at the clip's size (97 x 512 x 896, a 4-pixel grid of tracked points): wall time of
visualize_tracking_DELTA (host colour tables + HIP) and of its two kernels alone.  usage: raster_bench.py [grid_step]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from flexam_amd import conditioning_raster as P
from flexam_amd import hip as H

step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t_n, h, w = 97, 512, 896
rng = np.random.default_rng(0)
ys, xs = np.meshgrid(np.arange(step // 2, h, step), np.arange(step // 2, w, step), indexing="ij")
base = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.float32)
n = base.shape[0]
pts = np.zeros((t_n, n, 3), np.float32)
drift = rng.normal(0, 0.6, (n, 2)).astype(np.float32)
for t in range(t_n):
    pts[t, :, :2] = base + drift * t
pts[:, :, 2] = rng.uniform(0.5, 9, (t_n, n))
vis = rng.random((t_n, n)) > 0.05
dev = "cuda:0"
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr, cos, dep = P.visualize_tracking_DELTA(pts, vis, False, 4, h, w, 4, device=dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"visualize_tracking_DELTA: {n} points x {t_n} frames -> 6 videos [1, 3, {t_n}, {h}, {w}]: {dt:.3f} s", flush=True)
d_pts, d_vis = torch.from_numpy(pts).to(dev), torch.from_numpy(vis).to(dev)
colors = torch.randint(0, 256, (n, 3), dtype=torch.uint8, device=dev)
keys = H.raster_keys(d_pts, d_vis, h, w, 2, 0)
out = torch.empty(3, t_n, h, w, device=dev)
for name, fn, nbytes in (("raster_keys", lambda: H.raster_keys(d_pts, d_vis, h, w, 2, 0, keys=keys), t_n * h * w * 8),
                         ("raster_resolve (float planes)", lambda: H.raster_resolve(keys, colors, out_f32=out), t_n * h * w * (8 + 12))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt * 1e3:.3f} ms  ({nbytes / dt / 1e9:.0f} GB/s of key / output bytes)")
t0 = time.perf_counter(); P._depth_colors(pts, vis); t1 = time.perf_counter(); P._tracking_colors(pts[0], h, w); t2 = time.perf_counter()
enc = P.apply_cosine_positional_encoding(torch.from_numpy(pts), h, w, 4); t3 = time.perf_counter()
print(f"host: depth colours {t1 - t0:.3f} s, tracking colours {t2 - t1:.4f} s, cosine encodings (CPU tensors) {t3 - t2:.3f} s")
