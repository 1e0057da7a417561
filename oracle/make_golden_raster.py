"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/g13_raster_*.safetensors from the REFERENCE's conditioning rasteriser.

Run where /root/reference is mounted:   python -m oracle.make_golden_raster
Each fixture holds seeded inputs (points [T, N, 3], visibility, optional mask video) and what the reference's own methods
(oracle/ref_raster.py: pipelines.py:1501-1641, 1658-1850, run through PIL) return for them: tracking frames, the four cosine
encodings and their frames, depth frames, all uint8 [T, H, W, 3].  Only data is written.

  g13_raster_plain       random points around a 64 x 48 frame, 10 % invisible                      (full_edit)
  g13_raster_edges       + NaN / inf coordinates, points on the frame's edges and in row 0, negative fractions (truncation towards zero)
  g13_raster_foreground  + a random mask video, generate_type = foreground_edit                     (pipelines.py:1842-1850)
  g13_raster_wide        point_wise = 6 (7 x 7 squares in the tracking and depth videos, 5 x 5 in the cosine ones), all points visible
"""
import os

import numpy as np
import torch
from safetensors.torch import save_file

from . import ref_raster

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
T, N, H, W = 3, 400, 48, 64


def case(name):
    """Seeded inputs of fixture `name` (regenerated identically by the tests)."""
    rng = np.random.default_rng({"plain": 1, "edges": 2, "foreground": 3, "wide": 4}[name])
    pts = np.stack([rng.uniform(-6, W + 6, (T, N)), rng.uniform(-6, H + 6, (T, N)), rng.uniform(0.5, 5, (T, N))], -1).astype(np.float32)
    vis = rng.random((T, N)) > 0.1
    mask, gen, point_wise = None, "full_edit", 4
    if name in ("edges", "foreground"):
        pts[1, 5, 0] = np.nan
        pts[2, 7, 1] = np.inf
        pts[0, 9, :2] = (3.7, 0.4)                      # row 0: drawn in the cosine / depth videos, not in the tracking video
        pts[1, 11, :2] = (-0.5, 10.2)                   # truncates to x = 0: inside
        pts[2, 12, :2] = (W - 0.01, H - 0.01)           # last pixel
        pts[0, 13, :2] = (0.0, 1.0)
        pts[1, 14, :2] = (W, 5.0)                       # x = W: outside
        vis[0, 9] = vis[1, 11] = vis[2, 12] = vis[0, 13] = True
        pts[0, 9, 2] = pts[1, 11, 2] = pts[2, 12, 2] = pts[0, 13, 2] = 0.1       # nearest: on top wherever they are drawn
    if name == "foreground":
        mask, gen = (rng.random((T, H, W)) > 0.5).astype(np.float32), "foreground_edit"
    if name == "wide":
        point_wise, vis = 6, np.ones((T, N), dtype=bool)
    return pts, vis, mask, gen, point_wise


def reference_outputs(ref, pts, vis, mask, gen, point_wise):
    u8 = lambda v: (v[0].permute(1, 2, 3, 0).numpy() * 255).round().astype(np.uint8)          # [1, 3, T, H, W] float -> [T, H, W, 3] bytes
    out = {"tracking": np.stack(ref.fun_visualize_tracking_with_depth(torch.from_numpy(pts), torch.from_numpy(vis), H, W, point_wise=point_wise,
                                                                        mask_video=mask, generate_type=gen))}
    enc = ref.apply_cosine_positional_encoding(torch.from_numpy(pts), H, W, 4)
    cos = ref._visualize_cosine_encoded_tracking(enc, pts, vis, H, W, False, mask_video=mask, generate_type=gen)
    for i in range(4):
        out[f"encoding{i}"] = enc[i].numpy()
        out[f"cos{i}"] = u8(cos[i])
    out["depth"] = u8(ref._visualize_depth_tracking(torch.from_numpy(pts), vis, H, W, point_wise, False, mask_video=mask, generate_type=gen))
    return out


def main():
    import contextlib
    import io
    ref = ref_raster.load()
    for name in ("plain", "edges", "foreground", "wide"):
        pts, vis, mask, gen, point_wise = case(name)
        with contextlib.redirect_stdout(io.StringIO()):                                        # the reference prints per encoding level
            out = reference_outputs(ref, pts, vis, mask, gen, point_wise)
        tensors = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
        tensors["points"], tensors["visible"] = torch.from_numpy(pts), torch.from_numpy(vis.astype(np.uint8))
        if mask is not None:
            tensors["mask"] = torch.from_numpy(mask)
        save_file(tensors, os.path.join(OUT, f"g13_raster_{name}.safetensors"))
        print(f"g13_raster_{name}: painted pixels tracking {int((out['tracking'] > 0).any(-1).sum())}, depth {int((out['depth'] > 0).any(-1).sum())}")


if __name__ == "__main__":
    main()
