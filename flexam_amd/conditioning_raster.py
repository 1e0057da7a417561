"""Conditioning rasteriser: tracked 3-D points -> the conditioning videos the sampler's VAE encode consumes.

Mirrors the rasteriser methods of the reference's `FlexAMPipeline` (/root/reference/pipelines.py) under their own names:

  fun_visualize_tracking_with_depth    pipelines.py:1501-1575    tracking video (colour = first-frame position / inverse depth)
  apply_cosine_positional_encoding     pipelines.py:1577-1641    cos(2^i pi * normalised coordinates), i < L
  _visualize_cosine_encoded_tracking   pipelines.py:1730-1761    one video per encoding level
  _visualize_depth_tracking            pipelines.py:1763-1820    depth video (Spectral colormap of the per-frame depth percentiles)
  visualize_tracking_DELTA             pipelines.py:1852-1902    all six, as [1, 3, T, H, W] float tensors in [0, 1]

The reference draws every visible point of every frame as a PIL rectangle from a Python loop, far to near: ~125 s per 97 x 512 x 896
clip with a 4-pixel grid of points, an order of magnitude more than the 50 denoising steps here.  What that loop computes per pixel is
the colour of the NEAREST point among the squares covering it; csrc/raster.hip computes exactly that (a 64-bit atomic minimum per
covered pixel, then a colour gather) and leaves the frames on the GPU, where `Wan2_2FunControlPipeline_FlexAM` encodes them.

Split of the work: the O(T N) colour tables (percentiles, clips, the colormap) are numpy / torch expressions on the host, written as the
reference writes them so the bytes agree; the O(T N squares) drawing is HIP.  There is no CPU drawing path: without the HIP library
`flexam_amd.hip` raises.

Differences from the reference, all at its undefined corners: points of EQUAL depth are ordered by index (lower index on top) where the
reference's order is numpy's unstable argsort; the random blue channel / random z code it draws when every depth is zero takes an explicit
`generator` (default: numpy's global state, like the reference); `save_tracking=True` (mp4 files through moviepy) and `mask_path`
(a video file) are the caller's business -- pass `mask_video` [T, H, W] instead."""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import hip

# ColorBrewer "Spectral", 11 classes: the anchors of matplotlib's `Spectral` colormap (matplotlib/_cm.py `_Spectral_data`), which the
# reference indexes at pipelines.py:1770,1792.  matplotlib interpolates them linearly into a 256-entry table.
_SPECTRAL_ANCHORS = ((158, 1, 66), (213, 62, 79), (244, 109, 67), (253, 174, 97), (254, 224, 139), (255, 255, 191), (230, 245, 152),
                     (171, 221, 164), (102, 194, 165), (50, 136, 189), (94, 79, 162))
_spectral_table = None


def _spectral_bytes_table() -> np.ndarray:
    """[258, 3] uint8: (colormap LUT * 255) truncated, + the under (= first) and over (= last) entries matplotlib appends."""
    global _spectral_table
    if _spectral_table is None:
        a = np.array(_SPECTRAL_ANCHORS, dtype=np.float64) / 255.0
        xa, xi = np.linspace(0.0, 1.0, len(a)), np.linspace(0.0, 1.0, 256)
        ind = np.searchsorted(xa, xi)[1:-1]
        frac = ((xi[1:-1] - xa[ind - 1]) / (xa[ind] - xa[ind - 1]))[:, None]
        lut = np.clip(np.concatenate([a[:1], frac * (a[ind] - a[ind - 1]) + a[ind - 1], a[-1:]], 0), 0.0, 1.0)
        _spectral_table = (np.concatenate([lut, lut[:1], lut[-1:]], 0) * 255).astype(np.uint8)
    return _spectral_table


def _spectral_bytes(x: np.ndarray) -> np.ndarray:
    """(colormap(x, bytes=False)[:, :3] * 255).astype(uint8) (pipelines.py:1792; matplotlib Colormap.__call__: index int(x * 256) in x's
    own precision, 256 -> 255, negative -> under, above -> over, NaN -> the `bad` colour (0, 0, 0))."""
    with np.errstate(invalid="ignore"):
        xa = np.array(x, copy=True) * 256
        bad = np.isnan(xa)
        xa = np.where(xa < 0, -1, xa)
        xa = np.where(xa == 256, 255, xa)
        idx = np.clip(np.where(bad, 0, xa), -1, 256).astype(int)
    idx = np.where(idx < 0, 256, np.where(idx > 255, 257, idx))
    rgb = _spectral_bytes_table()[idx]
    rgb[bad] = 0
    return rgb


def _as_numpy_points(points) -> np.ndarray:
    if isinstance(points, torch.Tensor):
        points = points.detach().cpu().numpy()
    points = np.asarray(points)
    if points.ndim != 3 or points.shape[2] != 3:
        raise ValueError(f"points must be [T, N, 3] (u, v, depth), got {points.shape}")
    return points


def _prepare_vis_mask(vis_mask, points_shape) -> np.ndarray:
    """pipelines.py:1662-1673 (+ :1509-1515): [T, N] or [T, N, 1], tensor or array; None = everything visible."""
    t_n, n, _ = points_shape
    if vis_mask is None:
        return np.ones((t_n, n), dtype=bool)
    if isinstance(vis_mask, torch.Tensor):
        vis_mask = vis_mask.detach().cpu().numpy()
    vis_mask = np.asarray(vis_mask)
    if vis_mask.ndim == 3 and vis_mask.shape[2] == 1:
        vis_mask = vis_mask.squeeze(-1)
    if vis_mask.shape != (t_n, n):
        raise ValueError(f"vis_mask must be [T, N] = {(t_n, n)}, got {vis_mask.shape}")
    return vis_mask.astype(bool)


def _mask_for(mask_video, generate_type, t_n, height, width, device):
    """_should_draw_point (pipelines.py:1842-1850) filters only in the foreground / background modes."""
    if mask_video is None or generate_type not in ("foreground_edit", "background_edit"):
        return None
    m = torch.as_tensor(np.asarray(mask_video.detach().cpu() if isinstance(mask_video, torch.Tensor) else mask_video), dtype=torch.float32)
    if tuple(m.shape) != (t_n, height, width):
        raise ValueError(f"mask_video must be [T, H, W] = {(t_n, height, width)}, got {tuple(m.shape)}")
    return m.to(device).contiguous()


def _tracking_colors(first_frame_pts: np.ndarray, height: int, width: int, generator=None) -> np.ndarray:
    """pipelines.py:1523-1545: red <- u / W, green <- v / H, blue <- inverse depth between its 2nd and 98th percentile (first frame)."""
    n = first_frame_pts.shape[0]
    colors = np.zeros((n, 3), dtype=np.uint8)
    colors[:, 0] = (np.clip((first_frame_pts[:, 0] - 0) / (width - 0), 0, 1) * 255).astype(np.uint8)
    colors[:, 1] = (np.clip((first_frame_pts[:, 1] - 0) / (height - 0), 0, 1) * 255).astype(np.uint8)
    z_values = first_frame_pts[:, 2]
    if np.all(z_values == 0):
        # the reference draws from numpy's global state (np.random.randint, pipelines.py:1536); an explicit generator of either numpy
        # kind makes the corner reproducible: np.random.Generator (default_rng) has `integers`, np.random.RandomState `randint`
        if generator is not None and hasattr(generator, "integers"):
            colors[:, 2] = generator.integers(0, 256, n, dtype=np.uint8)
        else:
            colors[:, 2] = (generator if generator is not None else np.random).randint(0, 256, n, dtype=np.uint8)
    else:
        inv_z = 1 / (z_values + 1e-10)
        p2, p98 = np.percentile(inv_z, 2), np.percentile(inv_z, 98)
        colors[:, 2] = (np.clip((inv_z - p2) / (p98 - p2 + 1e-10), 0, 1) * 255).astype(np.uint8)
    return colors


def _generate_colors_from_points(first_frame_points: np.ndarray, num_points: int) -> np.ndarray:
    """pipelines.py:1675-1692: every channel <- (cos code + 1) / 2."""
    colors = np.zeros((num_points, 3), dtype=np.uint8)
    for c in range(3):
        colors[:, c] = (np.clip((first_frame_points[:, c] + 1) / 2, 0, 1) * 255).astype(np.uint8)
    return colors


def _depth_colors(points: np.ndarray, vis: np.ndarray) -> np.ndarray:
    """pipelines.py:1775-1792 per frame: depth of the visible points clipped to its 2nd ... 98th percentile through the Spectral colormap.
    [T, N, 3] uint8; rows of invisible points stay 0 (they are never drawn)."""
    t_n, n, _ = points.shape
    out = np.zeros((t_n, n, 3), dtype=np.uint8)
    for t in range(t_n):
        d = points[t, vis[t], 2]
        if d.size == 0:
            continue
        p2, p98 = np.percentile(d, [2, 98])
        norm = (np.clip(d, p2, p98) - p2) / (p98 - p2) if p98 > p2 else np.zeros_like(d)
        out[t, vis[t]] = _spectral_bytes(norm)
    return out


def apply_cosine_positional_encoding(pred_tracks_with_depth, height: int, width: int, L: int = 4, generator: Optional[torch.Generator] = None):
    """pipelines.py:1577-1641, the reference's own torch expressions on whatever device the tracks live on: list of L tensors [T, N, 3]."""
    pts = pred_tracks_with_depth if isinstance(pred_tracks_with_depth, torch.Tensor) else torch.as_tensor(np.asarray(pred_tracks_with_depth))
    x_n = torch.clamp((pts[:, :, 0] - 0) / (width - 0), 0, 1)
    y_n = torch.clamp((pts[:, :, 1] - 0) / (height - 0), 0, 1)
    z = pts[:, :, 2]
    if torch.all(z == 0):
        z_n = torch.rand_like(z) if generator is None else torch.rand(z.shape, generator=generator, dtype=z.dtype, device=generator.device).to(z.device)
    else:
        inv_z = 1 / (z + 1e-10)
        inv_np = inv_z.detach().cpu().numpy()
        p2, p98 = np.percentile(inv_np, 2), np.percentile(inv_np, 98)
        p2_t = torch.tensor(p2, device=inv_z.device, dtype=inv_z.dtype)
        p98_t = torch.tensor(p98, device=inv_z.device, dtype=inv_z.dtype)
        z_n = torch.clamp((inv_z - p2_t) / (p98_t - p2_t + 1e-10), 0, 1)
    norm = torch.zeros_like(pts)
    norm[:, :, 0], norm[:, :, 1], norm[:, :, 2] = x_n, y_n, z_n
    return [torch.cos(((2 ** i) * np.pi) * norm) for i in range(L)]


class _Frames:
    """The key images of one clip: which point every pixel shows.  Two selections exist -- the tracking video's (frame test y > 0,
    pipelines.py:1211) and the one the cosine and depth videos share (y >= 0) -- and with equal square sizes they differ only in the
    points of image row 0, so each is rasterised once and resolved with as many colour tables as there are videos."""

    def __init__(self, points: np.ndarray, vis: np.ndarray, height: int, width: int, mask, device):
        self.t_n, self.n = points.shape[:2]
        self.h, self.w, self.device = height, width, device
        self.points = torch.from_numpy(self._kernel_points(points)).to(device)
        self.vis = torch.from_numpy(np.ascontiguousarray(vis).view(np.uint8)).to(device)
        self.mask = mask
        self._keys: Dict[Tuple[int, int], torch.Tensor] = {}

    @staticmethod
    def _kernel_points(points: np.ndarray) -> np.ndarray:
        """float32 [T, N, 3] for flexam_raster_keys, which truncates (u, v), tests the frame and orders depths in float32.  The
        reference does all three in the INPUT's precision (`pixels.astype(int)`, `depths.argsort()`: pipelines.py:1556-1560, 1227), so
        for tracks wider than float32 the decisions are taken here, in that precision, and handed over in a form float32 holds exactly:
        (u, v) -> their truncation towards zero (63.99999999 must stay pixel 63, not round up to 64.0f and leave a 64-wide frame),
        depth -> its dense rank within the frame (equal depths equal ranks, NaN stays NaN: the kernel's order and tie rule see exactly
        the float64 order)."""
        if points.dtype == np.float32 or points.dtype.itemsize < 4 or not np.issubdtype(points.dtype, np.floating):
            return np.ascontiguousarray(points, dtype=np.float32)          # float32 itself, and everything float32 holds exactly
        out = np.empty(points.shape, dtype=np.float32)
        with np.errstate(invalid="ignore"):
            xy = np.trunc(points[:, :, :2])
            xy = np.where(np.isfinite(xy), np.clip(xy, -(1 << 24), 1 << 24), np.nan)      # non-finite stays rejected, far outside stays outside
        out[:, :, :2] = xy
        for t in range(points.shape[0]):
            z = points[t, :, 2]
            nan = np.isnan(z)
            _, inv = np.unique(np.where(nan, 0.0, z), return_inverse=True)     # ascending distinct values; -0.0 == 0.0
            out[t, :, 2] = np.where(nan, np.nan, inv.reshape(z.shape).astype(np.float32))
        return out

    def keys(self, half: int, y_min: int) -> torch.Tensor:
        k = (half, y_min)
        if k not in self._keys:
            self._keys[k] = hip.raster_keys(self.points, self.vis, self.h, self.w, half, y_min, self.mask)
        return self._keys[k]

    def video(self, colors: np.ndarray, half: int, y_min: int, as_bytes: bool = False) -> torch.Tensor:
        c = torch.from_numpy(np.ascontiguousarray(colors)).to(self.device)
        u8, f32 = hip.raster_resolve(self.keys(half, y_min), c, want_u8=as_bytes, want_f32=not as_bytes)
        return u8 if as_bytes else f32.unsqueeze(0)          # [T, H, W, 3] bytes, or [1, 3, T, H, W] = _convert_frames_to_tensor(..).unsqueeze(0)


def _device(device):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def fun_visualize_tracking_with_depth(pred_tracks_with_depth, pred_visibility, height, width, point_wise=4, mask_video=None,
                                      generate_type="full_edit", device=None, generator=None) -> torch.Tensor:
    """pipelines.py:1501-1575.  Returns the frames as ONE uint8 tensor [T, H, W, 3] on the GPU (the reference: a list of T numpy frames)."""
    points = _as_numpy_points(pred_tracks_with_depth)
    vis = _prepare_vis_mask(pred_visibility, points.shape)
    dev = _device(device)
    fr = _Frames(points, vis, height, width, _mask_for(mask_video, generate_type, points.shape[0], height, width, dev), dev)
    return fr.video(_tracking_colors(points[0], height, width, generator), point_wise // 2, 1, as_bytes=True)


def _visualize_cosine_encoded_tracking(encoded_tracks_list, original_points, vis_mask, height, width, save_tracking=False, mask_video=None,
                                       generate_type="full_edit", device=None, _frames: Optional[_Frames] = None) -> Dict[int, torch.Tensor]:
    """pipelines.py:1730-1761: {level: [1, 3, T, H, W]}.  Positions are the ORIGINAL points; the squares are +-2 whatever point_wise is."""
    if save_tracking:
        raise NotImplementedError("save_tracking=True writes mp4 files through moviepy in the reference (pipelines.py:1755-1757): not part of this build")
    points = _as_numpy_points(original_points)
    vis = _prepare_vis_mask(vis_mask, points.shape)
    dev = _device(device)
    fr = _frames or _Frames(points, vis, height, width, _mask_for(mask_video, generate_type, points.shape[0], height, width, dev), dev)
    out = {}
    for i, enc in enumerate(encoded_tracks_list):
        first = enc[0].detach().cpu().numpy() if isinstance(enc, torch.Tensor) else np.asarray(enc[0])
        out[i] = fr.video(_generate_colors_from_points(first, points.shape[1]), 2, 0)
    return out


def _visualize_depth_tracking(points, vis_mask, height, width, point_wise=4, save_tracking=False, mask_video=None, generate_type="full_edit",
                              device=None, _frames: Optional[_Frames] = None) -> torch.Tensor:
    """pipelines.py:1763-1820: [1, 3, T, H, W]."""
    if save_tracking:
        raise NotImplementedError("save_tracking=True writes mp4 files through moviepy in the reference (pipelines.py:1814-1818): not part of this build")
    pts = _as_numpy_points(points)
    vis = _prepare_vis_mask(vis_mask, pts.shape)
    dev = _device(device)
    fr = _frames or _Frames(pts, vis, height, width, _mask_for(mask_video, generate_type, pts.shape[0], height, width, dev), dev)
    return fr.video(_depth_colors(pts, vis), point_wise // 2, 0)


def visualize_tracking_DELTA(points, vis_mask=None, save_tracking=False, point_wise=4, height=480, width=720, cos_level=4,
                             generate_type="full_edit", mask_path=None, mask_video=None, device=None, generator=None, torch_generator=None):
    """pipelines.py:1852-1902: (tracking_video [1, 3, T, H, W], {level: [1, 3, T, H, W]}, depth_video [1, 3, T, H, W]), float32 in [0, 1],
    on the GPU.  `mask_video` [T, H, W] replaces the reference's `mask_path` (a video file it decodes itself, pipelines.py:1822-1840).
    All-zero depths are the one place the reference draws random numbers (pipelines.py:1536 numpy, :1611 torch, both from global state):
    `generator` (np.random.Generator or RandomState) seeds the tracking video's blue channel, `torch_generator` the z code of the cosine
    videos; None = the global states, as the reference."""
    if save_tracking:
        raise NotImplementedError("save_tracking=True writes mp4 files through moviepy in the reference (pipelines.py:1882-1885): not part of this build")
    if mask_path is not None:
        raise NotImplementedError("mask_path: decode the mask video yourself and pass mask_video [T, H, W] (1 = keep; invert it for background_edit "
                                  "as pipelines.py:1835-1837 does)")
    pts = _as_numpy_points(points)
    vis = _prepare_vis_mask(vis_mask, pts.shape)
    dev = _device(device)
    fr = _Frames(pts, vis, height, width, _mask_for(mask_video, generate_type, pts.shape[0], height, width, dev), dev)
    tracking_video = fr.video(_tracking_colors(pts[0], height, width, generator), point_wise // 2, 1)
    src = points if isinstance(points, torch.Tensor) else torch.from_numpy(pts)
    encoded = apply_cosine_positional_encoding(src, height, width, cos_level, generator=torch_generator)
    cos_video_dict = _visualize_cosine_encoded_tracking(encoded, pts, vis, height, width, False, device=dev, _frames=fr)
    depth_video = _visualize_depth_tracking(pts, vis, height, width, point_wise, False, device=dev, _frames=fr)
    return tracking_video, cos_video_dict, depth_video
