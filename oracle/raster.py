"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's conditioning rasteriser: tracked 3-D points ->
the six conditioning videos the sampler's VAE encode consumes (tracking colours, four cosine-encoded levels, depth colours).

Follows /root/reference/pipelines.py:
  tracking_frames      fun_visualize_tracking_with_depth   pipelines.py:1501-1575   (helpers :1200-1253)
  cosine_encodings     apply_cosine_positional_encoding    pipelines.py:1577-1641
  cosine_frames        _generate_colors_from_points / _render_cosine_encoded_frame / _visualize_cosine_encoded_tracking  :1675-1761
  depth_frames         _visualize_depth_tracking           pipelines.py:1763-1820
  should_draw          _should_draw_point                  pipelines.py:1842-1850
  frames_to_tensor     _convert_frames_to_tensor           pipelines.py:1658-1660
Pinned by tests/golden/g13_raster*.safetensors (outputs of the reference's own methods, oracle/make_golden_raster.py) and, where
/root/reference is mounted, by a live comparison (tests/test_raster_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

What the reference draws: per frame, the visible points with finite pixel coordinates, truncated to integers, inside the frame, in
order of DESCENDING depth, each as a filled axis-aligned square [x - h, x + h] x [y - h, y + h] (both ends inclusive: PIL's
rectangle; h = point_wise // 2, so 5 x 5 pixels for point_wise = 4), clipped by the image; later squares overwrite earlier ones,
so a pixel shows the NEAREST point covering it.  Here that is computed as a per-pixel maximum over draw ranks instead of by drawing.

Equal depths: the reference's order among them is whatever numpy's default (unstable, on x86 SIMD-dispatched) argsort returns -- not
reproducible across machines.  This restatement and the HIP kernels use ONE rule: of two points with equal depth the one with the
LOWER index is drawn last (a stable ascending sort, reversed).  NaN depths sort last in numpy, so they are drawn first."""
import numpy as np

# ColorBrewer "Spectral", 11 classes (matplotlib's `Spectral` colormap is the piecewise-linear interpolation of these anchors into a
# 256-entry table: matplotlib/_cm.py `_Spectral_data`, colors.LinearSegmentedColormap; third-party, pinned against matplotlib 3.10.8 in
# tests/test_raster_cpu.py)
_SPECTRAL_11 = ((158, 1, 66), (213, 62, 79), (244, 109, 67), (253, 174, 97), (254, 224, 139), (255, 255, 191), (230, 245, 152),
                (171, 221, 164), (102, 194, 165), (50, 136, 189), (94, 79, 162))


def spectral_lut_float():
    """[256, 3] float64: what matplotlib.colormaps["Spectral"] holds (LinearSegmentedColormap.from_list -> _create_lookup_table, N = 256)."""
    anchors = np.array(_SPECTRAL_11, dtype=np.float64) / 255.0
    xa = np.linspace(0.0, 1.0, len(anchors))
    xi = np.linspace(0.0, 1.0, 256)
    lut = np.empty((256, 3))
    for c in range(3):                                  # matplotlib: np.interp-like piecewise linear between the anchors, clipped to [0, 1]
        ind = np.searchsorted(xa, xi)[1:-1]
        dist = (xi[1:-1] - xa[ind - 1]) / (xa[ind] - xa[ind - 1])
        lut[:, c] = np.concatenate([[anchors[0, c]], dist * (anchors[ind, c] - anchors[ind - 1, c]) + anchors[ind - 1, c], [anchors[-1, c]]])
    return np.clip(lut, 0.0, 1.0)


def spectral_bytes(x):
    """(colormap(x, bytes=False)[:, :3] * 255).astype(np.uint8) for x in [0, 1] (pipelines.py:1792): index int(x * 256), 256 -> 255;
    NaN -> the colormap's `bad` colour (0, 0, 0)."""
    x = np.asarray(x)
    xa = np.array(x, copy=True)
    with np.errstate(invalid="ignore"):
        xa = xa * 256                                   # in x's own precision, as Colormap.__call__ does (xa *= self.N)
        bad = np.isnan(xa)
        xa = np.where(xa < 0, -1, xa)
        xa = np.where(xa == 256, 255, xa)
        idx = np.clip(np.where(bad, 0, xa), -1, 256).astype(int)
    lut = spectral_lut_float()
    table = np.concatenate([lut, lut[:1], lut[-1:]], 0)  # [256] = under -> first colour, [257] = over -> last colour
    idx = np.where(idx < 0, 256, np.where(idx > 255, 257, idx))
    rgb = table[idx]
    rgb = np.where(bad[:, None], 0.0, rgb)
    return (rgb * 255).astype(np.uint8)


def should_draw(x, y, mask_video, frame_idx, generate_type, width, height):
    """pipelines.py:1842-1850, vectorised over points: True everywhere unless a foreground / background mask filters."""
    if mask_video is None or generate_type not in ("foreground_edit", "background_edit"):
        return np.ones(x.shape, dtype=bool)
    inside = (x >= 0) & (x < width) & (y >= 0) & (y < height)
    out = np.zeros(x.shape, dtype=bool)
    out[inside] = mask_video[frame_idx, y[inside], x[inside]] > 0.5
    return out


def draw_order(depths):
    """Indices in drawing order (far to near).  Reference: depths.argsort()[::-1] (pipelines.py:1227,1714,1795); ties: see the header."""
    return np.argsort(depths, kind="stable")[::-1]


def paint(pixels, colors, height, width, half):
    """Squares of half-width `half` around integer `pixels` [n, 2] (x, y), drawn in the given order with `colors` [n, 3] uint8 onto a
    black frame; returns [H, W, 3] uint8.  (PIL: ImageDraw.rectangle([x - h, y - h, x + h, y + h], fill, outline), pipelines.py:1243-1253.)"""
    rank = np.full((height, width), -1, dtype=np.int64)
    n = pixels.shape[0]
    order = np.arange(n, dtype=np.int64)
    for dy in range(-half, half + 1):
        for dx in range(-half, half + 1):
            x, y = pixels[:, 0] + dx, pixels[:, 1] + dy
            ok = (x >= 0) & (x < width) & (y >= 0) & (y < height)
            np.maximum.at(rank, (y[ok], x[ok]), order[ok])
    img = np.zeros((height, width, 3), dtype=np.uint8)
    hit = rank >= 0
    img[hit] = colors[rank[hit]]
    return img


def tracking_colors(first_frame_pts, height, width, rng=None):
    """pipelines.py:1523-1545: red <- u / W, green <- v / H, blue <- inverse depth between its 2nd and 98th percentile (first frame)."""
    n = first_frame_pts.shape[0]
    colors = np.zeros((n, 3), dtype=np.uint8)
    colors[:, 0] = (np.clip((first_frame_pts[:, 0] - 0) / (width - 0), 0, 1) * 255).astype(np.uint8)
    colors[:, 1] = (np.clip((first_frame_pts[:, 1] - 0) / (height - 0), 0, 1) * 255).astype(np.uint8)
    z_values = first_frame_pts[:, 2]
    if np.all(z_values == 0):
        colors[:, 2] = (rng or np.random).randint(0, 256, n, dtype=np.uint8)      # unpinnable by construction (the reference draws from the global RNG)
    else:
        inv_z = 1 / (z_values + 1e-10)
        p2, p98 = np.percentile(inv_z, 2), np.percentile(inv_z, 98)
        colors[:, 2] = (np.clip((inv_z - p2) / (p98 - p2 + 1e-10), 0, 1) * 255).astype(np.uint8)
    return colors


def _visible_pixels(pts_t, vis_t, width, height, y_min):
    """The selection every renderer starts with: visible, finite, truncated to int, inside the frame (y >= y_min).  Returns the
    ORIGINAL indices of the selected points, their pixels [m, 2] int64 and their depths."""
    idx = np.nonzero(np.asarray(vis_t).astype(bool))[0]
    pixels, depths = pts_t[idx, :2], pts_t[idx, 2]
    valid = np.isfinite(pixels).all(axis=1)
    idx, pixels, depths = idx[valid], pixels[valid].astype(int), depths[valid]
    inside = (pixels[:, 0] >= 0) & (pixels[:, 0] < width) & (pixels[:, 1] >= y_min) & (pixels[:, 1] < height)
    return idx[inside], pixels[inside], depths[inside]


def tracking_frames(points, vis_mask, height, width, point_wise=4, mask_video=None, generate_type="full_edit", rng=None):
    """fun_visualize_tracking_with_depth (pipelines.py:1501-1575): list of T frames [H, W, 3] uint8.  NB its frame test is
    `y > 0` (valid_mask, pipelines.py:1211): points in image row 0 are not drawn in THIS video."""
    points = np.asarray(points)
    t_n, n, _ = points.shape
    vis = np.ones((t_n, n), dtype=bool) if vis_mask is None else np.asarray(vis_mask).reshape(t_n, n)
    colors = tracking_colors(points[0], height, width, rng)
    frames = []
    for i in range(t_n):
        idx, pixels, depths = _visible_pixels(points[i], vis[i], width, height, 1)
        order = draw_order(depths.astype(np.float64))
        idx, pixels = idx[order], pixels[order]
        keep = should_draw(pixels[:, 0], pixels[:, 1], mask_video, i, generate_type, width, height)
        frames.append(paint(pixels[keep], colors[idx[keep]], height, width, point_wise // 2))
    return frames


def cosine_encodings(points, height, width, levels=4):
    """apply_cosine_positional_encoding (pipelines.py:1577-1641) with the reference's own torch expressions (float32, CPU):
    list of `levels` arrays [T, N, 3] = cos(2^i pi * normalised (x, y, inverse depth))."""
    import torch
    pts = torch.as_tensor(np.asarray(points))
    x_n = torch.clamp((pts[:, :, 0] - 0) / (width - 0), 0, 1)
    y_n = torch.clamp((pts[:, :, 1] - 0) / (height - 0), 0, 1)
    z = pts[:, :, 2]
    if torch.all(z == 0):
        z_n = torch.rand_like(z)                       # unpinnable by construction
    else:
        inv_z = 1 / (z + 1e-10)
        inv_np = inv_z.numpy()
        p2, p98 = np.percentile(inv_np, 2), np.percentile(inv_np, 98)
        z_n = torch.clamp((inv_z - torch.tensor(p2, dtype=inv_z.dtype)) / (torch.tensor(p98, dtype=inv_z.dtype) - torch.tensor(p2, dtype=inv_z.dtype) + 1e-10), 0, 1)
    norm = torch.zeros_like(pts)
    norm[:, :, 0], norm[:, :, 1], norm[:, :, 2] = x_n, y_n, z_n
    return [torch.cos(((2 ** i) * np.pi) * norm).numpy() for i in range(levels)]


def cosine_colors(encoded_first_frame):
    """_generate_colors_from_points (pipelines.py:1675-1692): every channel <- (cos + 1) / 2."""
    n = encoded_first_frame.shape[0]
    colors = np.zeros((n, 3), dtype=np.uint8)
    for c in range(3):
        colors[:, c] = (np.clip((encoded_first_frame[:, c] + 1) / 2, 0, 1) * 255).astype(np.uint8)
    return colors


def cosine_frames(encoded, original_points, vis_mask, height, width, mask_video=None, generate_type="full_edit"):
    """One level of _visualize_cosine_encoded_tracking (pipelines.py:1730-1761): positions are the ORIGINAL points, colours the cosine
    code of the first frame; squares are +-2 whatever point_wise says (pipelines.py:1724-1725)."""
    original_points = np.asarray(original_points)
    t_n, n, _ = original_points.shape
    vis = np.ones((t_n, n), dtype=bool) if vis_mask is None else np.asarray(vis_mask).reshape(t_n, n)
    colors = cosine_colors(encoded[0])
    frames = []
    for t in range(t_n):
        idx, pixels, depths = _visible_pixels(original_points[t], vis[t], width, height, 0)
        order = draw_order(depths)
        idx, pixels = idx[order], pixels[order]
        keep = should_draw(pixels[:, 0], pixels[:, 1], mask_video, t, generate_type, width, height)
        frames.append(paint(pixels[keep], colors[idx[keep]], height, width, 2))
    return frames


def depth_colors(points, vis_mask):
    """Per-frame colours of _visualize_depth_tracking (pipelines.py:1775-1792): depth of the VISIBLE points clipped to its 2nd ... 98th
    percentile, through the Spectral colormap.  [T, N, 3] uint8 (rows of invisible points stay 0: never drawn)."""
    points = np.asarray(points)
    t_n, n, _ = points.shape
    vis = np.ones((t_n, n), dtype=bool) if vis_mask is None else np.asarray(vis_mask).reshape(t_n, n)
    out = np.zeros((t_n, n, 3), dtype=np.uint8)
    for t in range(t_n):
        v = vis[t].astype(bool)
        d = points[t, v, 2]
        if d.size == 0:
            continue
        p2, p98 = np.percentile(d, [2, 98])
        norm = (np.clip(d, p2, p98) - p2) / (p98 - p2) if p98 > p2 else np.zeros_like(d)
        out[t, v] = spectral_bytes(norm)
    return out


def depth_frames(points, vis_mask, height, width, point_wise=4, mask_video=None, generate_type="full_edit"):
    """_visualize_depth_tracking (pipelines.py:1763-1820)."""
    points = np.asarray(points)
    t_n, n, _ = points.shape
    vis = np.ones((t_n, n), dtype=bool) if vis_mask is None else np.asarray(vis_mask).reshape(t_n, n)
    colors = depth_colors(points, vis)
    frames = []
    for t in range(t_n):
        idx = np.nonzero(vis[t].astype(bool))[0]
        order = draw_order(points[t, idx, 2])
        idx = idx[order]
        uv = points[t, idx, :2]
        fin = np.isfinite(uv[:, 0]) & np.isfinite(uv[:, 1])
        idx, pixels = idx[fin], uv[fin].astype(int)      # int(): truncation towards zero, like astype(int)
        inside = (pixels[:, 0] >= 0) & (pixels[:, 0] < width) & (pixels[:, 1] >= 0) & (pixels[:, 1] < height)
        idx, pixels = idx[inside], pixels[inside]
        keep = should_draw(pixels[:, 0], pixels[:, 1], mask_video, t, generate_type, width, height)
        frames.append(paint(pixels[keep], colors[t, idx[keep]], height, width, point_wise // 2))
    return frames


def frames_to_tensor(frames):
    """_convert_frames_to_tensor (pipelines.py:1658-1660) + the callers' unsqueeze(0): [1, 3, T, H, W] float32 in [0, 1]."""
    import torch
    return (torch.from_numpy(np.stack(frames)).permute(3, 0, 1, 2).float() / 255.0).unsqueeze(0)


def visualize_tracking(points, vis_mask=None, point_wise=4, height=480, width=720, cos_level=4, generate_type="full_edit", mask_video=None):
    """visualize_tracking_DELTA without the file output (pipelines.py:1852-1902): (tracking [1,3,T,H,W], {level: [1,3,T,H,W]}, depth)."""
    points = np.asarray(points)
    t_n, n, _ = points.shape
    vis = None if vis_mask is None else np.asarray(vis_mask).reshape(t_n, n)
    tracking = frames_to_tensor(tracking_frames(points, vis, height, width, point_wise, mask_video, generate_type))
    enc = cosine_encodings(points, height, width, cos_level)
    cos = {i: frames_to_tensor(cosine_frames(e, points, vis, height, width, mask_video, generate_type)) for i, e in enumerate(enc)}
    depth = frames_to_tensor(depth_frames(points, vis, height, width, point_wise, mask_video, generate_type))
    return tracking, cos, depth
