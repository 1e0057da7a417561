"""CPU: the conditioning rasteriser's oracle (oracle/raster.py) against the fixtures the reference's own methods produced
(tests/golden/g13_raster_*: pipelines.py:1501-1641, 1658-1850 run through PIL, oracle/make_golden_raster.py), the product's host-side
colour tables (flexam_amd/conditioning_raster.py: numpy, no GPU needed) against the oracle, the restated Spectral table against
matplotlib where it is installed, and -- where /root/reference is mounted -- the oracle against the live reference on another seed."""
import numpy as np
import pytest
import torch

from oracle import make_golden_raster as G
from oracle import raster as O

CASES = ("plain", "edges", "foreground", "wide")


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(golden, name):
    g = golden(f"g13_raster_{name}")
    pts, vis, mask, gen, point_wise = G.case(name)
    assert np.array_equal(g["points"].numpy(), pts, equal_nan=True) and np.array_equal(g["visible"].numpy().astype(bool), vis)   # the seeded inputs regenerate
    assert np.array_equal(np.stack(O.tracking_frames(pts, vis, G.H, G.W, point_wise, mask, gen)), g["tracking"].numpy())
    enc = O.cosine_encodings(pts, G.H, G.W, 4)
    for i in range(4):
        assert np.array_equal(enc[i], g[f"encoding{i}"].numpy(), equal_nan=True)
        assert np.array_equal(np.stack(O.cosine_frames(enc[i], pts, vis, G.H, G.W, mask, gen)), g[f"cos{i}"].numpy())
    assert np.array_equal(np.stack(O.depth_frames(pts, vis, G.H, G.W, point_wise, mask, gen)), g["depth"].numpy())


def test_fixtures_exercise_the_corner_cases(golden):
    """The `edges` fixture must actually contain what it claims: a point in image row 0 (index 9, frame 0) that the cosine / depth
    videos draw and the tracking video does not (its frame test is y > 0, pipelines.py:1211), and occlusion."""
    g = golden("g13_raster_edges")
    pts, vis, mask, gen, point_wise = G.case("edges")
    assert int(pts[0, 9, 1]) == 0 and vis[0, 9]
    without = vis.copy()
    without[0, 9] = False
    assert np.array_equal(np.stack(O.tracking_frames(pts, without, G.H, G.W, point_wise, mask, gen)), g["tracking"].numpy())
    enc = O.cosine_encodings(pts, G.H, G.W, 1)
    assert not np.array_equal(np.stack(O.cosine_frames(enc[0], pts, without, G.H, G.W, mask, gen)), g["cos0"].numpy())
    assert 0 < int((g["tracking"].numpy() > 0).any(-1).sum()) < 25 * G.T * G.N


def test_product_colour_tables_match_the_oracle():
    from flexam_amd import conditioning_raster as P
    for name in CASES:
        pts, vis, _, _, _ = G.case(name)
        assert np.array_equal(P._tracking_colors(pts[0], G.H, G.W), O.tracking_colors(pts[0], G.H, G.W))
        assert np.array_equal(P._depth_colors(pts, vis), O.depth_colors(pts, vis))
        enc_p, enc_o = P.apply_cosine_positional_encoding(torch.from_numpy(pts), G.H, G.W, 4), O.cosine_encodings(pts, G.H, G.W, 4)
        for a, b in zip(enc_p, enc_o):
            assert np.array_equal(a.numpy(), b, equal_nan=True)
            assert np.array_equal(P._generate_colors_from_points(a[0].numpy(), G.N), O.cosine_colors(b[0]))
    x = np.concatenate([np.linspace(0, 1, 20001).astype(np.float32), np.array([0, 1, np.nan, 0.5], np.float32)])
    assert np.array_equal(P._spectral_bytes(x), O.spectral_bytes(x))


def test_all_zero_depths_take_the_random_branches():
    """pipelines.py:1536-1537, 1616-1618: unpinnable against the reference (it draws from the global RNGs); here the branch is seeded."""
    from flexam_amd import conditioning_raster as P
    pts, _, _, _, _ = G.case("plain")
    pts = pts.copy()
    pts[..., 2] = 0
    a = P._tracking_colors(pts[0], G.H, G.W, np.random.RandomState(3))
    b = P._tracking_colors(pts[0], G.H, G.W, np.random.RandomState(3))
    assert np.array_equal(a, b) and len(np.unique(a[:, 2])) > 50
    # both numpy generator kinds (round-5 advice: default_rng() has no randint), each reproducible
    c = P._tracking_colors(pts[0], G.H, G.W, np.random.default_rng(3))
    d = P._tracking_colors(pts[0], G.H, G.W, np.random.default_rng(3))
    assert np.array_equal(c, d) and len(np.unique(c[:, 2])) > 50 and c.dtype == np.uint8
    assert np.array_equal(c[:, :2], a[:, :2])                                     # only the blue channel is random
    e = P.apply_cosine_positional_encoding(torch.from_numpy(pts), G.H, G.W, 2, generator=torch.Generator().manual_seed(5))
    f = P.apply_cosine_positional_encoding(torch.from_numpy(pts), G.H, G.W, 2, generator=torch.Generator().manual_seed(5))
    assert torch.equal(e[1], f[1]) and e[0][..., 2].std() > 0.1


def test_float64_tracks_keep_the_decisions_of_their_own_precision():
    """The reference truncates pixel coordinates and orders depths in the INPUT's dtype (pipelines.py:1556-1560, 1227).  The kernel
    works in float32, so for float64 tracks the host hands it the truncated coordinates and per-frame depth ranks
    (_Frames._kernel_points): same pixels, same frame test, same order as the float64 oracle -- where a plain float32 cast moves a
    point out of the frame (63.99999999 -> 64.0f) or merges distinct depths into a tie."""
    from flexam_amd.conditioning_raster import _Frames
    w, h = 64, 48
    pts = np.array([[[63.99999999, 10.2, 2.0], [w - 1e-9, h - 1e-9, 1.0 + 1e-12], [5.5, 0.99999999999, 1.0], [-0.5, -0.9999999, 3.0],
                     [7.0, 7.0, np.nan], [np.inf, 3.0, 1.0], [-1.0, 5.0, 1.0], [1e300, 1.0, -0.0], [2.0, 2.0, 0.0]]], dtype=np.float64)
    assert np.float32(pts[0, 0, 0]) == 64.0 and np.float32(pts[0, 1, 2]) == np.float32(pts[0, 2, 2])      # what the plain cast would do
    kp = _Frames._kernel_points(pts)
    assert kp.dtype == np.float32
    vis = np.ones(pts.shape[1], bool)
    idx, pixels, depths = O._visible_pixels(pts[0], vis, w, h, 0)                  # the oracle, in float64
    # the kernel's decisions on kp: finite, -1 < u < W, -1 < v < H, then truncation
    u, v = kp[0, :, 0], kp[0, :, 1]
    with np.errstate(invalid="ignore"):
        inside = (u > -1) & (u < w) & (v > -1) & (v < h)
    assert np.array_equal(np.nonzero(inside)[0], np.sort(idx))
    for i, (x, y) in zip(idx, pixels):
        assert (int(kp[0, i, 0]), int(kp[0, i, 1])) == (int(x), int(y))
    # depth order: ranks are exact in float32 and order like the float64 depths (ties stay ties, NaN stays NaN)
    z64, z32 = pts[0, :, 2], kp[0, :, 2]
    assert np.isnan(z32[4]) and z32[7] == z32[8]                                   # -0.0 == 0.0
    fin = ~np.isnan(z64)
    for a in np.nonzero(fin)[0]:
        for b in np.nonzero(fin)[0]:
            assert (z64[a] < z64[b]) == (z32[a] < z32[b]) and (z64[a] == z64[b]) == (z32[a] == z32[b])
    # float32 and narrower inputs pass through untouched
    p32 = pts.astype(np.float32)
    assert np.array_equal(_Frames._kernel_points(p32), p32, equal_nan=True)


def test_spectral_table_is_matplotlibs():
    matplotlib = pytest.importorskip("matplotlib")
    cm = matplotlib.colormaps["Spectral"]
    x = np.concatenate([np.linspace(0, 1, 100001).astype(np.float32), np.array([0, 1, 0.5, np.nan, 0.999999, 1e-9], np.float32)])
    assert np.array_equal((cm(x, bytes=False)[:, :3] * 255).astype(np.uint8), O.spectral_bytes(x))
    x64 = np.linspace(0, 1, 4099)
    assert np.array_equal((cm(x64, bytes=False)[:, :3] * 255).astype(np.uint8), O.spectral_bytes(x64))


def test_equal_depths_lower_index_on_top():
    """The one rule this build adds (the reference's order among equal depths is numpy's unstable argsort)."""
    pts = np.zeros((1, 3, 3), np.float32)
    pts[0, :, :2] = ((10, 10), (11, 10), (12, 10))
    pts[0, :, 2] = (2.0, 2.0, 1.0)                                              # point 2 is nearest; 0 and 1 tie
    fr = O.cosine_frames(np.array([[[1, 1, 1], [0, 0, 0], [-1, -1, -1]]], np.float32), pts, None, 24, 24)[0]
    assert fr[10, 8, 0] == 255 and fr[10, 9, 0] == 255                          # columns 8, 9: only point 0 / points 0 and 1 -> 0 on top
    assert fr[10, 10, 0] == 0 and fr[10, 14, 0] == 0 and fr[10, 13, 0] == 0     # from column 10 on point 2 covers (12 - 2 = 10)
    assert fr[10, 12, 0] == 0


@pytest.mark.needs_reference
def test_oracle_matches_live_reference_other_seed():
    import contextlib
    import io
    from oracle import ref_raster
    ref = ref_raster.load()
    rng = np.random.default_rng(77)
    t_n, n, h, w = 4, 900, 40, 72
    pts = np.stack([rng.uniform(-8, w + 8, (t_n, n)), rng.uniform(-8, h + 8, (t_n, n)), rng.uniform(0.2, 9, (t_n, n))], -1).astype(np.float32)
    vis = rng.random((t_n, n)) > 0.2
    mask = (rng.random((t_n, h, w)) > 0.4).astype(np.float32)
    u8 = lambda v: (v[0].permute(1, 2, 3, 0).numpy() * 255).round().astype(np.uint8)
    for gen, m in (("full_edit", None), ("background_edit", mask)):
        with contextlib.redirect_stdout(io.StringIO()):
            tr = np.stack(ref.fun_visualize_tracking_with_depth(torch.from_numpy(pts), torch.from_numpy(vis), h, w, point_wise=4, mask_video=m, generate_type=gen))
            enc = ref.apply_cosine_positional_encoding(torch.from_numpy(pts), h, w, 4)
            cos = ref._visualize_cosine_encoded_tracking(enc, pts, vis, h, w, False, mask_video=m, generate_type=gen)
            dep = u8(ref._visualize_depth_tracking(torch.from_numpy(pts), vis, h, w, 4, False, mask_video=m, generate_type=gen))
        got_t, got_c, got_d = O.visualize_tracking(pts, vis, 4, h, w, 4, gen, m)
        assert np.array_equal(u8(got_t), tr) and np.array_equal(u8(got_d), dep)
        for i in range(4):
            assert np.array_equal(u8(got_c[i]), u8(cos[i]))
