"""GPU: the HIP conditioning rasteriser (csrc/raster.hip behind flexam_amd/conditioning_raster.py, through the C ABI) -- bit-exact
against the reference's fixtures (tests/golden/g13_raster_*) and the numpy oracle, and at the clip's full size (97 x 512 x 896, a
4-pixel grid of 28672 tracked points) through properties that do not need the oracle: every painted pixel shows a point whose square
covers it and no nearer drawn point does, point order does not matter, a second run is identical."""
import numpy as np
import pytest
import torch

from oracle import make_golden_raster as G
from oracle import raster as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bytes(video):
    """[1, 3, T, H, W] float in [0, 1] -> [T, H, W, 3] uint8; the float must be EXACTLY byte / 255 in float32."""
    v = video[0].permute(1, 2, 3, 0).cpu()
    b = (v * 255).round().to(torch.uint8)
    assert torch.equal(b.float() / 255.0, v)
    return b.numpy()


@pytest.mark.parametrize("name", ("plain", "edges", "foreground", "wide"))
def test_hip_rasteriser_matches_reference_fixture(golden, name):
    from flexam_amd import conditioning_raster as P
    g = golden(f"g13_raster_{name}")
    pts, vis, mask, gen, point_wise = G.case(name)
    tracking, cos, depth = P.visualize_tracking_DELTA(torch.from_numpy(pts), torch.from_numpy(vis), False, point_wise, G.H, G.W, 4, gen,
                                                      mask_video=mask, device=DEV)
    assert tracking.shape == (1, 3, G.T, G.H, G.W) and tracking.dtype == torch.float32 and tracking.is_cuda
    assert np.array_equal(_bytes(tracking), g["tracking"].numpy())
    for i in range(4):
        assert np.array_equal(_bytes(cos[i]), g[f"cos{i}"].numpy())
    assert np.array_equal(_bytes(depth), g["depth"].numpy())
    fr = P.fun_visualize_tracking_with_depth(torch.from_numpy(pts), torch.from_numpy(vis), G.H, G.W, point_wise, mask, gen, device=DEV)
    assert fr.dtype == torch.uint8 and np.array_equal(fr.cpu().numpy(), g["tracking"].numpy())


def test_hip_rasteriser_vs_oracle_ties_nan_depths_and_background_mask():
    """What the fixtures cannot hold (the reference's order among equal depths is not reproducible): many equal depths, NaN and
    negative depths, -0.0, a [T, N, 1] visibility tensor, background_edit with a mask."""
    from flexam_amd import conditioning_raster as P
    rng = np.random.default_rng(5)
    t_n, n, h, w = 5, 3000, 72, 104
    pts = np.stack([rng.uniform(-8, w + 8, (t_n, n)), rng.uniform(-8, h + 8, (t_n, n)), rng.integers(1, 6, (t_n, n)).astype(np.float64)], -1).astype(np.float32)
    pts[1, :50, 2] = np.nan
    pts[2, 50:90, 2] = -pts[2, 50:90, 2]
    pts[3, 90:120, 2] = -0.0
    pts[3, 120:150, 2] = 0.0
    vis = rng.random((t_n, n, 1)) > 0.15
    mask = (rng.random((t_n, h, w)) > 0.5).astype(np.float32)
    for gen, m, pw in (("full_edit", None, 4), ("background_edit", mask, 4), ("full_edit", mask, 8)):
        tr, cos, dep = P.visualize_tracking_DELTA(pts, torch.from_numpy(vis), False, pw, h, w, 2, gen, mask_video=m, device=DEV)
        o_t, o_c, o_d = O.visualize_tracking(pts, vis, pw, h, w, 2, gen, m)
        assert torch.equal(tr.cpu(), o_t) and torch.equal(dep.cpu(), o_d)
        for i in range(2):
            assert torch.equal(cos[i].cpu(), o_c[i])


def test_float64_tracks_and_seeded_zero_depth_clip():
    """float64 tracks: frames equal the float64 oracle's (truncation, frame test and depth order in the input's precision, where a
    float32 cast moves points across pixel / frame borders and merges depths); all-zero depths: the clip is reproducible from the two
    explicit generators (numpy for the tracking colours, torch for the cosine z code)."""
    from flexam_amd import conditioning_raster as P
    rng = np.random.default_rng(11)
    t_n, n, h, w = 3, 2000, 64, 64
    pts = np.stack([rng.uniform(-4, w + 4, (t_n, n)), rng.uniform(-4, h + 4, (t_n, n)), rng.uniform(1, 2, (t_n, n))], -1)      # float64
    pts[:, :200, 0] = np.floor(pts[:, :200, 0]) + 1 - 1e-9          # just under a pixel border: float32 rounds these UP a pixel
    pts[:, 200:300, 1] = h - 1e-10                                  # just inside the last row: float32 puts them outside the frame
    pts[:, 300:600, 2] = 1.5 + rng.integers(0, 300, (t_n, 300)) * 1e-13      # distinct in float64, one value in float32
    assert len(np.unique(pts[0, 300:600, 2].astype(np.float32))) == 1
    vis = rng.random((t_n, n)) > 0.1
    tr, cos, dep = P.visualize_tracking_DELTA(pts, vis, False, 4, h, w, 2, device=DEV)
    o_t, o_c, o_d = O.visualize_tracking(pts, vis, 4, h, w, 2)
    assert torch.equal(tr.cpu(), o_t) and torch.equal(dep.cpu(), o_d) and all(torch.equal(cos[i].cpu(), o_c[i]) for i in range(2))
    tr32, _, _ = P.visualize_tracking_DELTA(pts.astype(np.float32), vis, False, 4, h, w, 2, device=DEV)
    assert not torch.equal(tr32, tr)                                # the case does separate the two precisions
    z0 = pts.astype(np.float32)
    z0[..., 2] = 0
    runs = [P.visualize_tracking_DELTA(z0, vis, False, 4, h, w, 2, device=DEV, generator=np.random.default_rng(4),
                                       torch_generator=torch.Generator().manual_seed(9)) for _ in range(2)]
    assert torch.equal(runs[0][0], runs[1][0]) and all(torch.equal(runs[0][1][i], runs[1][1][i]) for i in range(2))
    other = P.visualize_tracking_DELTA(z0, vis, False, 4, h, w, 2, device=DEV, generator=np.random.default_rng(5), torch_generator=torch.Generator().manual_seed(10))
    assert not torch.equal(other[0], runs[0][0]) and not torch.equal(other[1][0], runs[0][1][0])


def test_raster_resolve_float_is_the_correctly_rounded_quotient():
    """out_f32 = byte / 255 for every byte value, as torch's `.float() / 255.0` gives it (pipelines.py:1660)."""
    from flexam_amd import hip as H
    pts = torch.zeros(1, 256, 3, device=DEV)
    pts[0, :, 0] = torch.arange(256, device=DEV) + 0.5
    pts[0, :, 1] = 0.5
    pts[0, :, 2] = 1.0
    colors = torch.arange(256, dtype=torch.uint8, device=DEV)[:, None].repeat(1, 3).contiguous()
    keys = H.raster_keys(pts.contiguous(), None, 1, 256, 0)
    u8, f32 = H.raster_resolve(keys, colors, want_u8=True, want_f32=True)
    assert torch.equal(u8[0, 0, :, 0].cpu(), torch.arange(256, dtype=torch.uint8))
    assert torch.equal(f32[1, 0, 0].cpu(), torch.arange(256, dtype=torch.uint8).float() / 255.0)


def _grid_clip(t_n, h, w, step, seed):
    rng = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(step // 2, h, step), np.arange(step // 2, w, step), indexing="ij")
    base = np.stack([xs.ravel(), ys.ravel()], -1).astype(np.float32)
    n = base.shape[0]
    pts = np.zeros((t_n, n, 3), np.float32)
    drift = rng.normal(0, 0.6, (n, 2)).astype(np.float32)
    for t in range(t_n):
        pts[t, :, :2] = base + drift * t + rng.normal(0, 0.3, (n, 2))
    pts[:, :, 2] = rng.permutation(t_n * n).reshape(t_n, n).astype(np.float32) / (t_n * n) * 9 + 0.5       # all depths distinct
    return pts, rng.random((t_n, n)) > 0.05


def test_full_size_clip_properties_and_two_frames_vs_oracle():
    """BASELINE size: 97 x 512 x 896, 28672 points per frame (a 4-pixel grid that drifts apart)."""
    from flexam_amd import conditioning_raster as P
    from flexam_amd import hip as H
    t_n, h, w = 97, 512, 896
    pts, vis = _grid_clip(t_n, h, w, 4, 11)
    n = pts.shape[1]
    tr, cos, dep = P.visualize_tracking_DELTA(pts, vis, False, 4, h, w, 4, device=DEV)
    assert tr.shape == (1, 3, t_n, h, w) and len(cos) == 4 and dep.shape == tr.shape
    # (1) two frames against the oracle
    enc3 = O.cosine_encodings(pts, h, w, 4)[3]                                                   # its depth code is normalised over the WHOLE clip
    u8 = lambda v, t: (v[0, :, t].permute(1, 2, 0).cpu() * 255).round().to(torch.uint8).numpy()
    for t in (0, 61):
        sub_p, sub_v = pts[[0, t]], vis[[0, t]]                                                  # frame 0 defines the colours
        assert np.array_equal(u8(tr, t), O.tracking_frames(sub_p, sub_v, h, w, 4)[1])
        assert np.array_equal(u8(cos[3], t), O.cosine_frames(enc3[[0, t]], sub_p, sub_v, h, w)[1])
        assert np.array_equal(u8(dep, t), O.depth_frames(sub_p, sub_v, h, w, 4)[1])
    # (2) the key image: the winner of every painted pixel covers it, is drawn, and no drawn point covering it is nearer
    d_pts, d_vis = torch.from_numpy(pts).to(DEV), torch.from_numpy(vis).to(DEV)
    keys = H.raster_keys(d_pts, d_vis, h, w, 2, 0)
    t_sel = 40
    k = keys[t_sel].cpu().numpy().view(np.uint64)
    win = (k & np.uint64(0xFFFFFFFF)).astype(np.int64)
    painted = k != np.uint64(0xFFFFFFFFFFFFFFFF)
    yy, xx = np.nonzero(painted)
    wi = win[painted]
    px = pts[t_sel, wi, :2].astype(int)
    assert vis[t_sel, wi].all() and (np.abs(px[:, 0] - xx) <= 2).all() and (np.abs(px[:, 1] - yy) <= 2).all()
    drawn = np.nonzero(vis[t_sel])[0]
    dp = pts[t_sel, drawn, :2].astype(int)
    inside = (dp[:, 0] >= 0) & (dp[:, 0] < w) & (dp[:, 1] >= 0) & (dp[:, 1] < h)
    best = np.full((h, w), np.inf, np.float32)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            x, y = dp[inside, 0] + dx, dp[inside, 1] + dy
            ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
            np.minimum.at(best, (y[ok], x[ok]), pts[t_sel, drawn[inside][ok], 2])
    assert np.array_equal(np.isfinite(best), painted) and np.array_equal(best[painted], pts[t_sel, wi, 2])
    # (3) the order of the points does not matter (distinct depths), and a second run is identical
    perm = np.random.default_rng(3).permutation(n)
    tr2, cos2, dep2 = P.visualize_tracking_DELTA(pts[:, perm], vis[:, perm], False, 4, h, w, 4, device=DEV)
    assert torch.equal(tr, tr2) and torch.equal(dep, dep2) and all(torch.equal(cos[i], cos2[i]) for i in range(4))
    tr3, _, _ = P.visualize_tracking_DELTA(pts, vis, False, 4, h, w, 4, device=DEV)
    assert torch.equal(tr, tr3)


def test_raster_argument_errors():
    from flexam_amd import conditioning_raster as P
    from flexam_amd import hip as H
    pts = torch.zeros(2, 5, 3, device=DEV)
    with pytest.raises(RuntimeError):
        H.raster_keys(pts[:, :, :2], None, 8, 8, 2)
    with pytest.raises(RuntimeError):
        H.raster_keys(pts, torch.ones(2, 4, dtype=torch.uint8, device=DEV), 8, 8, 2)
    with pytest.raises(RuntimeError):
        H.raster_resolve(H.raster_keys(pts, None, 8, 8, 2), torch.zeros(5, 4, dtype=torch.uint8, device=DEV))
    spread = pts.clone()
    spread[:, :, 0] = torch.arange(5, device=DEV, dtype=torch.float32)           # x = 0 .. 4, equal depths: pixel column 6 shows point 4 only
    keys5 = H.raster_keys(spread, None, 8, 8, 2)
    with pytest.raises(RuntimeError, match="index 5 points"):                     # a colour table of another point count (round-5 advice)
        H.raster_resolve(keys5, torch.zeros(4, 3, dtype=torch.uint8, device=DEV))
    with pytest.raises(RuntimeError, match="index 5 points"):
        H.raster_resolve(keys5, torch.zeros(2, 7, 3, dtype=torch.uint8, device=DEV))
    # keys from elsewhere (no point count attached): the C side clamps -- an index past the table resolves to black, not to a wild read
    anon = keys5.clone()
    u8, _ = H.raster_resolve(anon, torch.full((3, 3), 200, dtype=torch.uint8, device=DEV), want_u8=True, want_f32=False)
    drawn = (keys5 != -1)
    idx = (keys5 & 0xFFFFFFFF)
    past = drawn & (idx >= 3)
    assert bool(past.any()) and bool((drawn & (idx < 3)).any())
    assert torch.equal(u8[..., 0] == 200, drawn & (idx < 3)) and int(u8[past].max()) == 0
    with pytest.raises(RuntimeError, match="y_min"):
        H.raster_keys(pts, None, 8, 8, 2, y_min=2)                    # the C ABI's own check (FLEXAM_E_SHAPE + message)
    with pytest.raises(RuntimeError, match="half"):
        H.raster_keys(pts, None, 8, 8, -1)
    with pytest.raises(NotImplementedError):
        P.visualize_tracking_DELTA(pts.cpu(), None, True, 4, 8, 8)
    with pytest.raises(NotImplementedError):
        P.visualize_tracking_DELTA(pts.cpu(), None, False, 4, 8, 8, mask_path="mask.mp4")
    with pytest.raises(ValueError):
        P.visualize_tracking_DELTA(pts.cpu(), torch.ones(2, 4), False, 4, 8, 8)


def test_tracks_to_conditioning_latents_through_the_pipeline():
    """The whole hand-over the reference does on the host (pipelines.py:1874-1902 -> :1167-1185 -> PIPE.py:655-822): tracks -> six videos
    (left on the GPU) -> the sampler's VAE encode.  The GPU-resident videos must encode to exactly what the same videos handed over as
    host tensors encode to, and to what the oracle's videos give through the oracle's encoder (>= 40 dB)."""
    from flexam_amd import conditioning_raster as P
    from test_pipeline_pixels_gpu import FRAMES, H, W, make_pipe
    from oracle import cases as C
    from oracle import vae as OV
    pipe, cfg, dsd, vsd = make_pipe()
    rng = np.random.default_rng(9)
    n = 500
    pts = np.stack([rng.uniform(-4, W + 4, (FRAMES, n)), rng.uniform(-4, H + 4, (FRAMES, n)), rng.uniform(0.5, 5, (FRAMES, n))], -1).astype(np.float32)
    vis = rng.random((FRAMES, n)) > 0.1
    tr, cos, dep = P.visualize_tracking_DELTA(pts, vis, False, 4, H, W, 4, device=DEV)
    shape = (1, 48, 3, H // 16, W // 16)
    mask = torch.full((1, 1, FRAMES, H, W), 255.0)
    on_gpu = pipe.encode_conditioning(tr, mask, tr, dep, cos, None, H, W, shape)          # all-255 mask: `video` is not encoded
    on_host = pipe.encode_conditioning(tr.cpu(), mask, tr.cpu(), dep.cpu(), {k: v.cpu() for k, v in cos.items()}, None, H, W, shape)
    assert torch.equal(on_gpu.control_latents, on_host.control_latents) and torch.equal(on_gpu.additional_control, on_host.additional_control)
    o_tr, o_cos, o_dep = O.visualize_tracking(pts, vis, 4, H, W, 4)
    enc = lambda v: OV.vae_encode(vsd, v * 2 - 1, C.VAE_ENC_SMALL["temporal_down"], OV.LATENT_MEAN, OV.LATENT_STD)
    want_add = torch.cat([enc(o_dep)] + [enc(o_cos[k]) for k in sorted(o_cos)], dim=1)
    p_c, p_a = C.psnr(on_gpu.control_latents.float().cpu(), enc(o_tr)), C.psnr(on_gpu.additional_control.float().cpu(), want_add)
    print(f"tracks -> latents: control {p_c:.1f} dB, depth + cosine levels {p_a:.1f} dB")
    assert p_c >= 40.0 and p_a >= 40.0
