"""GPU: direct parity tests of the channels-last conv / VAE helper kernels against torch fp32
formulas (the end-to-end VAE and cnn-block tests cover them too; these localise a failure)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def bf16_close(got, want, atol=1e-3):
    got, want = got.float().cpu(), want.float()
    tol = 2.0 ** -8 * want.abs() + atol
    assert bool(((got - want).abs() <= tol).all()), f"max err {(got - want).abs().max():.4g}"


def padded_rows(x):
    """[C,F,H,W] fp32 -> rows [(f,hp,wp), C] with zero borders."""
    c, f, h, w = x.shape
    p = torch.zeros(f, h + 2, w + 2, c)
    p[:, 1:-1, 1:-1] = x.permute(1, 2, 3, 0)
    return p.view(-1, c)


def test_pack_conv3x3_unpack_equals_conv2d():
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(1)
    c, f, h, w, co = 40, 2, 6, 10, 24
    x = torch.randn(c, f, h, w, generator=g).to(BF).float()
    wt = (torch.randn(co, c, 3, 3, generator=g) / math.sqrt(c * 9)).to(BF).float()
    b = torch.randn(co, generator=g)
    cp = 64
    guard = (w + 2 + 1) * cp
    buf = torch.zeros(guard * 2 + f * (h + 2) * (w + 2) * cp, dtype=BF, device=dev())
    img = buf[guard:guard + f * (h + 2) * (w + 2) * cp].view(f, h + 2, w + 2, cp)
    H.pack_cl(x.to(dev()), img, 0)
    wp = torch.zeros(co, 3, 3, cp)
    wp[..., :c] = wt.permute(0, 2, 3, 1)
    koff = torch.tensor([((dh - 1) * (w + 2) + (dw - 1)) * cp for dh in range(3) for dw in range(3)], dtype=torch.int64)
    rows = f * (h + 2) * (w + 2)
    y = H.gemm(img.view(rows, cp), wp.view(co, 9 * cp).to(BF).to(dev()), b.to(dev()), a_koff=koff.to(dev()), m=rows, k=9 * cp,
               out_dtype=torch.float32)
    got = H.unpack_cl(y, co, f, h, w)
    want = torch.stack([F.conv2d(x[:, i][None], wt, b, padding=1)[0] for i in range(f)], dim=1)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-4)


def test_groupnorm_silu_with_residual():
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(2)
    c, f, h, w, groups = 48, 2, 5, 7, 6
    x = torch.randn(c, f, h, w, generator=g) * 2 + 0.3
    res = torch.randn(c, f, h, w, generator=g).to(BF)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    rows = padded_rows(x) + 7.0 * (padded_rows(torch.ones_like(x)) == 0)        # garbage in the border rows must be ignored
    rimg = torch.zeros(f, h + 2, w + 2, 64, dtype=BF)
    rimg[:, 1:-1, 1:-1, :c] = res.permute(1, 2, 3, 0)
    dst = torch.zeros(f, h + 2, w + 2, 64, dtype=BF, device=dev())
    H.groupnorm_silu_cl(rows.to(dev()), c, f, h, w, groups, gamma.to(dev()), beta.to(dev()), dst, residual=rimg.to(dev()))
    want = F.silu(F.group_norm(x[None], groups, gamma, beta, 1e-5))[0] + res.float()
    got = dst[:, 1:-1, 1:-1, :c].permute(3, 0, 1, 2)
    bf16_close(got, want, atol=2e-3)
    assert float(dst[:, 0].abs().max()) == 0 and float(dst[..., c:].abs().max()) == 0     # borders / pad channels untouched


def test_vae_prep_modes_upsample_dupup_scatter_softmax_unpatchify():
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(3)
    c, t, h, w = 32, 2, 3, 5
    x = torch.randn(c, t, h, w, generator=g)
    rows = padded_rows(x).to(dev())
    gamma = 1 + 0.1 * torch.randn(c, generator=g)
    # prep mode 2 (RMS_norm + SiLU) into an image with 2 history frames; mode 1 compact
    img = torch.zeros(t + 2, h + 2, w + 2, 64, dtype=BF, device=dev())
    H.vae_prep_cl(rows, c, t, h, w, img, mode=2, gamma=gamma.to(dev()), t0=2)
    want = F.silu(F.normalize(x, dim=0) * math.sqrt(c) * gamma.view(c, 1, 1, 1))
    bf16_close(img[2:, 1:-1, 1:-1, :c].permute(3, 0, 1, 2), want)
    assert float(img[:2].abs().max()) == 0
    comp = torch.zeros(t * h * w, c, dtype=BF, device=dev())
    H.vae_prep_cl(rows, c, t, h, w, comp, mode=1, gamma=gamma.to(dev()), compact=True)
    bf16_close(comp.view(t, h, w, c).permute(3, 0, 1, 2), F.normalize(x, dim=0) * math.sqrt(c) * gamma.view(c, 1, 1, 1))
    # nearest 2x upsample with frame de-interleave of a [rows, 2c'] matrix
    cc = c // 2
    up = torch.zeros(2 * t, 2 * h + 2, 2 * w + 2, 64, dtype=BF, device=dev())
    H.upsample2x_cl(rows, cc, t, h, w, up, interleave=True)
    y = x.view(2, cc, t, h, w)
    inter = torch.stack((y[0], y[1]), dim=2).reshape(cc, 2 * t, h, w)
    want_up = F.interpolate(inter.permute(1, 0, 2, 3), scale_factor=2.0, mode="nearest-exact").permute(1, 0, 2, 3)
    bf16_close(up[:, 1:-1, 1:-1, :cc].permute(3, 0, 1, 2), want_up)
    # DupUp3D add (temporal factor 2, first-chunk crop and regular)
    co = 16
    for first, t_in in ((True, 1), (False, 2)):
        xin = torch.randn(c, t_in, h, w, generator=g)
        to = 1 if first else 2 * t_in
        main = torch.randn(co, to, 2 * h, 2 * w, generator=g)
        mrows = padded_rows(main).to(dev())
        H.dupup_add_cl(mrows, co, to, 2 * h, 2 * w, padded_rows(xin).to(dev()), c, 2, 1 if first else 0)
        rep = co * 8 // c
        d = xin[None].repeat_interleave(rep, dim=1).view(1, co, 2, 2, 2, t_in, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(1, co, 2 * t_in, 2 * h, 2 * w)
        d = d[:, :, 1:] if first else d
        got = mrows.view(to, 2 * h + 2, 2 * w + 2, co)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).cpu()
        torch.testing.assert_close(got, main + d[0], rtol=1e-6, atol=1e-6)
    # scatter-add of compact rows, row softmax, final unpatchify + clamp
    yb = torch.randn(t * h * w, c, generator=g).to(BF)
    xr = padded_rows(x).to(dev())
    H.scatter_add_cl(xr, yb.to(dev()), c, t, h, w)
    got = xr.view(t, h + 2, w + 2, c)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).cpu()
    torch.testing.assert_close(got, x + yb.float().view(t, h, w, c).permute(3, 0, 1, 2), rtol=1e-6, atol=1e-6)
    s = torch.randn(9, 24, generator=g) * 3
    p = torch.empty(9, 64, dtype=BF, device=dev())
    H.softmax_rows(s.to(dev()), 0.25, p, 24)
    bf16_close(p[:, :24], torch.softmax(s * 0.25, dim=1), atol=1e-4)
    assert float(p[:, 24:].abs().max()) == 0
    v = torch.randn(12, t, h, w, generator=g) * 1.5
    video = torch.zeros(3, 5, 2 * h, 2 * w, device=dev())
    H.vae_unpatchify_clamp(padded_rows(v).to(dev()), t, h, w, video, 3)
    want_v = v.view(3, 2, 2, t, h, w).permute(0, 3, 4, 2, 5, 1).reshape(3, t, 2 * h, 2 * w).clamp(-1, 1)      # (c r q) -> (h q) (w r)
    torch.testing.assert_close(video[:, 3:5].cpu(), want_v, rtol=0, atol=0)
    assert float(video[:, :3].abs().max()) == 0


def test_patchify_s2d_avgdown_match_torch():
    """Encoder helpers: patchify -> image, space-to-depth -> image, AvgDown3D add (incl. the front zero pad)."""
    import sys
    from flexam_amd import hip as H
    from oracle import vae as OV
    g = torch.Generator().manual_seed(5)
    # patchify: video [3, 5, 8, 12] frames 1..4 -> image frames 2..4 of [5, 6, 8, 64]
    vid = torch.randn(3, 5, 8, 12, generator=g)
    img = torch.zeros(5, 6, 8, 64, dtype=BF, device=dev())
    H.vae_patchify_cl(vid.to(dev()), 1, 3, img, t0=2)
    want = OV.patchify2(vid[None])[0][:, 1:4]                                   # [12, 3, 4, 6]
    bf16_close(img[2:, 1:-1, 1:-1, :12].permute(3, 0, 1, 2), want)
    assert float(img[:2].abs().max()) == 0 and float(img[..., 12:].abs().max()) == 0
    # space-to-depth
    c, t, h, w, cs = 24, 2, 4, 6, 64
    x = torch.randn(c, t, h, w, generator=g)
    rows = padded_rows(x).to(dev())
    s2d = torch.zeros(t, h // 2 + 2, w // 2 + 2, 4 * cs, dtype=BF, device=dev())
    H.space_to_depth_cl(rows, c, t, h, w, s2d, cs)
    for a in range(2):
        for b in range(2):
            grp = s2d[:, 1:-1, 1:-1, (a * 2 + b) * cs:(a * 2 + b) * cs + c].permute(3, 0, 1, 2)
            bf16_close(grp, x[:, :, a::2, b::2])
    assert float(s2d[:, -1].abs().max()) == 0 and float(s2d[:, :, -1].abs().max()) == 0
    # AvgDown3D add: (ft, fs, Ti) incl. odd frame count (front pad) and the identity case
    for ci, co, ti, ft, fs in ((8, 16, 4, 2, 2), (8, 16, 1, 2, 2), (8, 8, 3, 1, 2), (8, 8, 2, 1, 1)):
        hi, wi = 4, 8
        xin = torch.randn(ci, ti, hi, wi, generator=g)
        want = OV.avg_down3d(xin[None], co, ft, fs)[0]                          # [co, to, ho, wo]
        to, ho, wo = want.shape[1:]
        base = torch.randn(co, to, ho, wo, generator=g)
        xm = padded_rows(base).to(dev())
        H.avgdown_add_cl(xm, co, to, ho, wo, padded_rows(xin).to(dev()), ci, ti, ft, fs)
        got = xm.view(to, ho + 2, wo + 2, co)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).cpu()
        torch.testing.assert_close(got, base + want, rtol=1e-5, atol=1e-5)


def test_deinterleave_phase_dupup_and_tapsum_match_torch():
    """r5 helpers of the VAE decoder: temporal de-interleave at the input resolution, interleave of the four phase outputs + DupUp3D
    shortcut, and the 27-tap gather of the folded head convolution."""
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(9)
    # deinterleave: rows [T, 2C] -> image of 2T frames, frame 2t + s = channels [sC, (s+1)C) of frame t
    c, t, h, w = 8, 3, 4, 6
    x = torch.randn(2 * c, t, h, w, generator=g)
    img = torch.zeros(2 * t, h + 2, w + 2, 64, dtype=BF, device=dev())
    H.deinterleave_cl(padded_rows(x).to(dev()), c, t, h, w, img)
    y = x.view(2, c, t, h, w)
    want = torch.stack((y[0], y[1]), dim=2).reshape(c, 2 * t, h, w)
    bf16_close(img[:, 1:-1, 1:-1, :c].permute(3, 0, 1, 2), want)
    assert float(img[..., c:].abs().max()) == 0 and float(img[:, 0].abs().max()) == 0
    # phase_dupup: x_main[(t', 2y + a, 2x + b)] = phases[a * 2 + b][(t', y, x)] + DupUp3D(x_in); first-chunk crop and regular
    ci, co = 32, 16
    for first, t_in in ((True, 1), (False, 2)):
        xin = torch.randn(ci, t_in, h, w, generator=g)
        to = 1 if first else 2 * t_in
        ph = torch.randn(4, co, to, h, w, generator=g)
        ph_rows = torch.stack([padded_rows(ph[i]) for i in range(4)]).to(dev())
        out = torch.full((to * (2 * h + 2) * (2 * w + 2), co), float("nan"), device=dev())
        H.phase_dupup_cl(ph_rows, out, co, to, 2 * h, 2 * w, padded_rows(xin).to(dev()), ci, 2, 1 if first else 0)
        inter = torch.zeros(co, to, 2 * h, 2 * w)
        for a in range(2):
            for b in range(2):
                inter[:, :, a::2, b::2] = ph[a * 2 + b]
        rep = co * 8 // ci
        d = xin[None].repeat_interleave(rep, dim=1).view(1, co, 2, 2, 2, t_in, h, w).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(1, co, 2 * t_in, 2 * h, 2 * w)
        d = d[:, :, 1:] if first else d
        got = out.view(to, 2 * h + 2, 2 * w + 2, co)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).cpu()
        torch.testing.assert_close(got, inter + d[0], rtol=1e-6, atol=1e-6)
    # tapsum: out = bias + sum over 3x3x3 taps of Y[(t + dt, h + dh - 1, w + dw - 1), tap * Co + o] == causal conv3d of the image
    cin, co, kt, t = 6, 12, 3, 2
    xs = torch.randn(cin, kt - 1 + t, h, w, generator=g)                       # history frames + current frames
    wt = torch.randn(co, cin, kt, 3, 3, generator=g) / math.sqrt(cin * 27)
    b = torch.randn(co, generator=g)
    wf = wt.permute(2, 3, 4, 0, 1).reshape(kt * 9 * co, cin)
    yrows = (padded_rows(xs) @ wf.t()).to(dev()).contiguous()                   # per-tap products, fp32
    out = torch.zeros(t * (h + 2) * (w + 2), co, device=dev())
    H.tapsum_cl(yrows, t, h, w, kt, co, b.to(dev()), out)
    want = F.conv3d(F.pad(xs[None], (1, 1, 1, 1, 0, 0)), wt, b)[0]              # [co, t, h, w]
    got = out.view(t, h + 2, w + 2, co)[:, 1:-1, 1:-1].permute(3, 0, 1, 2).cpu()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    assert float(out.view(t, h + 2, w + 2, co)[:, 0].abs().max()) == 0          # border rows are not written


def test_row_walk_kernels_beyond_65535_image_rows():
    """T x H above the grid's y extent (round-5 advice: long encoder chunks at full height, tall inputs): the rows move to
    (h on y, t on z).  vae_prep (span and wave-per-position forms, fp32 and bf16 rows) and space-to-depth on 200 x 352 = 70400 rows
    against the same kernels run as four 50-frame calls (row counts the y extent holds)."""
    from flexam_amd import hip as H
    g = torch.Generator().manual_seed(9)
    c, t, h, w, cp = 64, 200, 352, 4, 64
    assert t * h > 65535
    rows = torch.randn(t * (h + 2) * (w + 2), c, generator=g).to(dev())
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(dev())
    per = (h + 2) * (w + 2)
    for src in (rows, rows.to(BF)):
        img = torch.zeros(t, h + 2, w + 2, cp, dtype=BF, device=dev())
        H.vae_prep_cl(src, c, t, h, w, img, mode=2, gamma=gamma)
        ref = torch.zeros_like(img)
        for t0 in range(0, t, 50):
            H.vae_prep_cl(src[t0 * per:(t0 + 50) * per], c, 50, h, w, ref[t0:t0 + 50], mode=2, gamma=gamma)
        assert torch.equal(img, ref) and float(img[-1, 1:-1, 1:-1].abs().max()) > 0
    s2d = torch.zeros(t, h // 2 + 2, w // 2 + 2, 4 * cp, dtype=BF, device=dev())
    H.space_to_depth_cl(rows, c, t, h, w, s2d, cp)
    ref = torch.zeros_like(s2d)
    for t0 in range(0, t, 50):
        H.space_to_depth_cl(rows[t0 * per:(t0 + 50) * per], c, 50, h, w, ref[t0:t0 + 50], cp)
    assert torch.equal(s2d, ref) and float(s2d[-1, 1:-1, 1:-1].abs().max()) > 0
