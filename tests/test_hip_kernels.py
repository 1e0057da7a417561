"""GPU parity tests of each HIP kernel, called through the C ABI (flexam_amd.hip -> libflexam_hip.so),
against the fp32 oracle (oracle/dit.py) on the same seeded inputs.

Tolerances (stated per op, "HIP bf16 path vs fp32 restatement"):
  * GEMM / attention: inputs are bf16-representable, products accumulate in fp32 on MFMA, the
    output is rounded once to bf16  ->  |err| <= 2^-8 |out| + small absolute slack;
  * elementwise / norm kernels: fp32 math, one bf16 rounding of the output -> 1 bf16 ulp;
  * fp32-output kernels (residual, sampler step, small linear): 1e-5 relative.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def bf(x):
    return x.to(BF)


def assert_bf16_close(got, want, ulps=1.0, atol=0.0, msg=""):
    got, want = got.float().cpu(), want.float().cpu()
    tol = ulps * (2.0 ** -8) * want.abs() + atol
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{msg} {int(bad.sum())}/{bad.numel()} off; max err {(got - want).abs().max():.4g}"


@pytest.fixture(scope="module")
def H():
    from flexam_amd import hip
    hip.device_check()
    return hip


# ----------------------------------------------------------------------------- GEMM
def test_gemm_identity_asymmetric(H):
    """A = I with an asymmetric W catches transposed / permuted fragment maps exactly."""
    n, k = 512, 256
    a = torch.zeros(256, k, dtype=BF, device=dev())
    a[torch.arange(256), torch.arange(256)] = 1
    w = bf(torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 251 - 125).to(dev())
    out = H.gemm(a, w, out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), w.float().cpu().t()[:256].contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (300, 200, 128), (1000, 772, 192), (2048, 1536, 3072), (77, 3072, 640)])
def test_gemm_exact_integers(H, m, n, k):
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    w = torch.randint(-3, 4, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    out = H.gemm(bf(a).to(dev()), bf(w).to(dev()), b.to(dev()), out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), a @ w.t() + b, rtol=0, atol=0)


def test_gemm_full_size_one_hot_rows_select_weight_columns(H):
    """Size-independent property at the BASELINE shape (M = 23296 tokens, N = K = 3072): with one-hot rows of A the
    product is a gather of W's columns -- exact in bf16, every tile of the 91 x 12 grid and every K block is exercised."""
    g = torch.Generator().manual_seed(17)
    m, n, k = 23296, 3072, 3072
    idx = torch.randint(0, k, (m,), generator=g)
    a = torch.zeros(m, k, dtype=BF)
    a[torch.arange(m), idx] = 1.0
    w = bf(torch.randn(n, k, generator=g))
    out = H.gemm(a.to(dev()), w.to(dev()))
    assert torch.equal(out.cpu(), w[:, idx].t().contiguous())
    x = torch.zeros(m, n, device=dev())
    H.gemm_gate_residual(a.to(dev()), w.to(dev()), None, x)
    assert torch.equal(x.cpu(), w[:, idx].t().float())


@pytest.mark.parametrize("m,n,k", [(5000, 3072, 3072), (2912, 3072, 14336), (1500, 1024, 1024), (70 * 256, 256, 2048),
                                   (160, 1024, 27648), (640, 1024, 27648), (1792, 512, 13824)])
def test_gemm_tail_split_k_is_exact_and_repeatable(H, m, n, k, monkeypatch):
    """Shapes whose tile count leaves the last round of the 256 CUs mostly empty: those tiles are cut along K (partial sums
    through the per-call scratch, summed in slice order by the finish launch).  Integer data -> exact; ten launches -> identical
    bits; all three epilogues.  The K = 27648 / 13824 rows are the VAE decoder's 3x3x3 convolutions at 1024 / 512 channels, where
    r1's in-launch hand-off returned wrong sums (profiles/r2_splitk_handoff_bug.txt)."""
    g = torch.Generator().manual_seed(m + k)
    lim = 2 if k < 20000 else 1
    a = torch.randint(-lim, lim + 1, (m, k), generator=g).float()
    w = torch.randint(-lim, lim + 1, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    ad, wd, bd = bf(a).to(dev()), bf(w).to(dev()), b.to(dev())
    outs = [H.gemm(ad, wd, bd, out_dtype=torch.float32) for _ in range(10)]
    torch.testing.assert_close(outs[0].cpu(), want, rtol=0, atol=0)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    x = x0.clone().to(dev())
    H.gemm_gate_residual(ad, wd, bd, x)
    torch.testing.assert_close(x.cpu(), x0 + want.to(BF).float(), rtol=0, atol=0)
    out16 = H.gemm(ad, wd, bd, epilogue=H.EPI_GELU_TANH)
    monkeypatch.setenv("FLEXAM_GEMM_SPLITK", "0")          # read once per process: this only documents the switch
    ref16 = torch.nn.functional.gelu(want, approximate="tanh").to(BF)
    assert_bf16_close(out16, ref16.float(), ulps=2.0, atol=1e-2, msg="gelu epilogue after split-K")


@pytest.mark.parametrize("m,n,k", [(16384, 2048, 256), (12000, 3072, 448), (23296, 3072, 192)])
def test_gemm_many_units_per_workgroup_exact(H, m, n, k):
    """More tiles than CUs: every persistent workgroup runs several units back to back, i.e. the path where the next unit's
    first two K blocks are prefetched under the epilogue and waited for with counted vmcnt while the epilogue stores are still
    in flight.  Integer data -> exact, for the bf16, fp32, GELU and gated-residual (per-row table) epilogues, three launches each."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    w = torch.randint(-2, 3, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    ad, wd, bd = bf(a).to(dev()), bf(w).to(dev()), b.to(dev())
    gate = torch.randint(-2, 3, (4, n), generator=g).float()
    rows = torch.randint(0, 4, (m,), generator=g, dtype=torch.int32)
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    ref16 = torch.nn.functional.gelu(want, approximate="tanh").to(BF)
    for _ in range(3):
        torch.testing.assert_close(H.gemm(ad, wd, bd).float().cpu(), want.to(BF).float(), rtol=0, atol=0)
        torch.testing.assert_close(H.gemm(ad, wd, bd, out_dtype=torch.float32).cpu(), want, rtol=0, atol=0)
        assert_bf16_close(H.gemm(ad, wd, bd, epilogue=H.EPI_GELU_TANH), ref16.float(), ulps=2.0, atol=1e-2, msg="gelu, many units")
        x = x0.clone().to(dev())
        H.gemm_gate_residual(ad, wd, bd, x, gate.to(dev()), rows.to(dev()))
        torch.testing.assert_close(x.cpu(), x0 + want.to(BF).float() * gate[rows.long()], rtol=0, atol=0)


@pytest.mark.parametrize("m,n,k", [(1000, 160, 192), (4097, 320, 5184), (777, 480, 256), (2048, 640, 1728), (300, 160, 27648)])
def test_gemm_160_wide_tile_shape_exact(H, m, n, k, monkeypatch):
    """Output widths that are multiples of 160 but not of 256 (the VAE encoder's 160 / 320 / 640 channels) run on a 256 x 160 tile
    (4 x 2 waves, 5 n-tiles per wave) instead of 62.5 %-filled 256-wide ones: integer data -> exact for the bf16, fp32 and
    gated-residual outputs, with and without the tail split-K (K = 27648), identical to the 256-wide shape's results."""
    monkeypatch.setenv("FLEXAM_GEMM_N160", "2")                 # also N = 640 (default: N <= 480 only)
    g = torch.Generator().manual_seed(m + n + k)
    lim = 2 if k < 20000 else 1
    a = torch.randint(-lim, lim + 1, (m, k), generator=g).float()
    w = torch.randint(-lim, lim + 1, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    ad, wd, bd = bf(a).to(dev()), bf(w).to(dev()), b.to(dev())
    torch.testing.assert_close(H.gemm(ad, wd, bd, out_dtype=torch.float32).cpu(), want, rtol=0, atol=0)
    torch.testing.assert_close(H.gemm(ad, wd, bd).float().cpu(), want.to(BF).float(), rtol=0, atol=0)
    gate = torch.randint(-2, 3, (3, n), generator=g).float()
    rows = torch.randint(0, 3, (m,), generator=g, dtype=torch.int32)
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    x = x0.clone().to(dev())
    H.gemm_gate_residual(ad, wd, bd, x, gate.to(dev()), rows.to(dev()))
    torch.testing.assert_close(x.cpu(), x0 + want.to(BF).float() * gate[rows.long()], rtol=0, atol=0)
    ref16 = torch.nn.functional.gelu(want, approximate="tanh")
    assert_bf16_close(H.gemm(ad, wd, bd, epilogue=H.EPI_GELU_TANH), ref16, ulps=2.0, atol=1e-2, msg="gelu, 160-wide shape")
    monkeypatch.setenv("FLEXAM_GEMM_N160", "0")
    torch.testing.assert_close(H.gemm(ad, wd, bd, out_dtype=torch.float32).cpu(), want, rtol=0, atol=0)


@pytest.mark.parametrize("m,n,k", [(2912, 3072, 3072), (2912, 3072, 14336), (1000, 768, 192), (4097, 384, 5184), (200, 192, 27648), (5824, 3072, 3072)])
def test_gemm_192_wide_tile_shape_exact(H, m, n, k, monkeypatch):
    """The 192 x 192 tile (r6: 2 x 4 waves, 6 m-tiles x 3 n-tiles each; LDS-staged epilogues with 96-byte wave rows), forced wherever
    N % 192 == 0 (FLEXAM_GEMM_N192=2; by itself the launcher takes it where it wins, e.g. a rank-of-eight's 2912 x 3072 launches = 16 x
    16 tiles = one full round of 256 CUs): ragged row counts, interior and edge tiles, several units per workgroup, the tail split-K
    (K = 27648 / 14336), all four epilogues incl. the per-row gate table -- integer data, exact, and identical to the 256-wide plans."""
    monkeypatch.setenv("FLEXAM_GEMM_N192", "2")
    g = torch.Generator().manual_seed(m + n + k)
    lim = 2 if k < 8000 else 1
    a = torch.randint(-lim, lim + 1, (m, k), generator=g).float()
    if k >= 8000:                                                # sparse rows keep |y| < 256, so the bf16 rounding of y changes nothing
        a = a * (torch.randint(0, 8, (m, k), generator=g) == 0).float()
    w = torch.randint(-lim, lim + 1, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    ad, wd, bd = bf(a).to(dev()), bf(w).to(dev()), b.to(dev())
    gate = torch.randint(-2, 3, (3, n), generator=g).float()
    rows = torch.randint(0, 3, (m,), generator=g, dtype=torch.int32)
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    ref16 = torch.nn.functional.gelu(want, approximate="tanh")
    res = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("FLEXAM_GEMM_N192", mode)
        o32 = H.gemm(ad, wd, bd, out_dtype=torch.float32)
        torch.testing.assert_close(o32.cpu(), want, rtol=0, atol=0)
        o16 = H.gemm(ad, wd, bd)
        torch.testing.assert_close(o16.float().cpu(), want.to(BF).float(), rtol=0, atol=0)
        og = H.gemm(ad, wd, bd, epilogue=H.EPI_GELU_TANH)
        assert_bf16_close(og, ref16, ulps=2.0, atol=1e-2, msg="gelu, 192-wide shape")
        xs = []
        for _ in range(2):
            x = x0.clone().to(dev())
            H.gemm_gate_residual(ad, wd, bd, x, gate.to(dev()), rows.to(dev()))
            xs.append(x)
        assert torch.equal(xs[0], xs[1])
        torch.testing.assert_close(xs[0].cpu(), x0 + want.to(BF).float() * gate[rows.long()], rtol=0, atol=0)
        xn = x0.clone().to(dev())
        H.gemm_gate_residual(ad, wd, None, xn)                 # no bias, no gate
        torch.testing.assert_close(xn.cpu(), x0 + (a @ w.t()).to(BF).float(), rtol=0, atol=0)
        res[mode] = (o16, og, xs[0])
    for u, v in zip(res["2"], res["0"]):
        assert torch.equal(u, v)                                # the same bits as the 256-wide plans (GELU included: same fp32 sums)


def test_cu_budget_plans_the_persistent_grids_for_fewer_cus(H):
    """flexam_set_cu_budget (r6): the library plans its one-workgroup-per-CU grids for a multiple of 8 CUs below the device's count (room
    for a collective's kernels beside compute); flexam_device_cus answers with the budget, results do not change, 0 restores all CUs."""
    full = H.num_cus()
    g = torch.Generator().manual_seed(8)
    m, n, k = 5000, 3072, 512
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    w = torch.randint(-2, 3, (n, k), generator=g).float()
    want = a @ w.t()
    q, kk, v = (torch.randn(1, 3000, 4, 128, generator=g).to(BF).to(dev()) for _ in range(3))
    try:
        base = H.attn_fwd(q, kk, v)
        H.set_cu_budget(full - 8)
        assert H.num_cus() == full - 8
        torch.testing.assert_close(H.gemm(bf(a).to(dev()), bf(w).to(dev()), out_dtype=torch.float32).cpu(), want, rtol=0, atol=0)
        assert_bf16_close(H.attn_fwd(q, kk, v), base.float().cpu(), ulps=2.0, atol=1e-3, msg="attention under a CU budget")      # (another tail split plan)
        for bad in (4, full + 8, full - 3):
            with pytest.raises(RuntimeError):
                H.set_cu_budget(bad)
    finally:
        H.set_cu_budget(0)
    assert H.num_cus() == full


@pytest.mark.parametrize("mt", [4, 5, 6, 7, 8])
def test_gemm_every_tile_height_exact(H, mt, monkeypatch):
    """The launch heuristic picks a tile height (32*MT rows) per shape; force each one and check exact integer results on
    ragged sizes, for the plain, fp32-out and gated-residual epilogues."""
    monkeypatch.setenv("FLEXAM_GEMM_MT", str(mt))
    g = torch.Generator().manual_seed(100 + mt)
    m, n, k = 1000, 772, 192
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    w = torch.randint(-3, 4, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    out = H.gemm(bf(a).to(dev()), bf(w).to(dev()), b.to(dev()), out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), want, rtol=0, atol=0)
    out16 = H.gemm(bf(a).to(dev()), bf(w).to(dev()), b.to(dev()))
    torch.testing.assert_close(out16.float().cpu(), want.to(BF).float(), rtol=0, atol=0)
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    gate = torch.randint(-2, 3, (3, n), generator=g).float()
    rows = torch.randint(0, 3, (m,), generator=g, dtype=torch.int32)
    x = x0.clone().to(dev())
    H.gemm_gate_residual(bf(a).to(dev()), bf(w).to(dev()), b.to(dev()), x, gate.to(dev()), rows.to(dev()))
    torch.testing.assert_close(x.cpu(), x0 + want.to(BF).float() * gate[rows.long()], rtol=0, atol=0)


@pytest.mark.parametrize("epi", ["none", "gelu"])
def test_gemm_random_bf16_out_strided(H, epi):
    g = torch.Generator().manual_seed(5)
    m, n, k = 700, 1024, 512
    big = bf(torch.randn(m, 3 * k, generator=g))
    a = big[:, k:2 * k]                                  # row stride 3k: a slice of a fused buffer
    w = bf(torch.randn(n, k, generator=g) / math.sqrt(k))
    b = torch.randn(n, generator=g) * 0.1
    outbuf = torch.zeros(m, 2 * n, dtype=BF, device=dev())
    H.gemm(big.to(dev())[:, k:2 * k], w.to(dev()), b.to(dev()), out=outbuf[:, n:],
           epilogue=H.EPI_GELU_TANH if epi == "gelu" else H.EPI_NONE)
    want = a.float() @ w.float().t() + b
    if epi == "gelu":
        want = F.gelu(want, approximate="tanh")
    assert_bf16_close(outbuf[:, n:], want, ulps=1.0, atol=2e-3, msg="gemm")
    assert float(outbuf[:, :n].abs().max()) == 0.0     # nothing written outside the view


def test_gemm_koff_implicit_conv(H):
    """Per-K-block A offsets: a 3-tap 1-D 'convolution' over rows of a [R, 64] buffer."""
    g = torch.Generator().manual_seed(9)
    rows, cin, cout, m = 400, 64, 128, 300
    x = bf(torch.randn(rows, cin, generator=g))
    w = bf(torch.randn(cout, 3 * cin, generator=g) / math.sqrt(3 * cin))
    koff = torch.tensor([0 * cin, 1 * cin, 2 * cin], dtype=torch.int64)        # tap t reads row m + t
    out = H.gemm(x.to(dev()), w.to(dev()), None, a_koff=koff.to(dev()), m=m, k=3 * cin, out_dtype=torch.float32)
    xa = torch.cat([x[0:m], x[1:m + 1], x[2:m + 2]], dim=1).float()
    torch.testing.assert_close(out.cpu(), xa @ w.float().t(), rtol=1e-4, atol=1e-4)


def test_gemm_gate_residual(H):
    g = torch.Generator().manual_seed(10)
    m, n, k = 520, 512, 256
    a = bf(torch.randn(m, k, generator=g))
    w = bf(torch.randn(n, k, generator=g) / math.sqrt(k))
    b = torch.randn(n, generator=g) * 0.1
    x = torch.randn(m, n, generator=g)
    gate = torch.randn(4, n, generator=g)
    rows = torch.randint(0, 4, (m,), generator=g, dtype=torch.int32)
    xd = x.clone().to(dev())
    H.gemm_gate_residual(a.to(dev()), w.to(dev()), b.to(dev()), xd, gate.to(dev()), rows.to(dev()))
    y = bf(a.float() @ w.float().t() + b).float()
    want = x + y * gate[rows.long()]
    # y is rounded to bf16 inside the kernel exactly as here, up to fp32 accumulation order
    torch.testing.assert_close(xd.cpu(), want, rtol=0, atol=2.0 ** -7 * 4)
    xd2 = x.clone().to(dev())
    H.gemm_gate_residual(a.to(dev()), w.to(dev()), b.to(dev()), xd2)           # gate = 1 (cross-attention form)
    torch.testing.assert_close(xd2.cpu(), x + y, rtol=0, atol=2.0 ** -7 * 4)


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v):
    from oracle import dit as O
    return O.attention(q.float(), k.float(), v.float())


@pytest.mark.parametrize("b,h,lq,lk", [(1, 2, 256, 256), (2, 3, 300, 77), (1, 1, 64, 1000), (1, 2, 1111, 1111)])
def test_attention_matches_oracle(H, b, h, lq, lk):
    g = torch.Generator().manual_seed(lq + lk)
    q = bf(torch.randn(b, lq, h, 128, generator=g))
    k = bf(torch.randn(b, lk, h, 128, generator=g))
    v = bf(torch.randn(b, lk, h, 128, generator=g))
    out = H.attn_fwd(q.to(dev()), k.to(dev()), v.to(dev()))
    assert_bf16_close(out, _attn_ref(q, k, v), ulps=2.0, atol=6e-3, msg="attention")


@pytest.mark.parametrize("splits,b,h,lq,lk", [(2, 1, 2, 300, 1111), (3, 2, 1, 64, 1600), (4, 1, 2, 513, 2055), (8, 1, 1, 256, 4100), (5, 1, 1, 100, 640)])
def test_attention_split_kv_matches_oracle(H, splits, b, h, lq, lk):
    """Keys cut into ranges (sequence-parallel ranks hold few query rows): partial softmaxes + merge must equal the one-pass
    result; ragged Lq / Lk, a last split shorter than the others, a partial last key tile."""
    g = torch.Generator().manual_seed(splits * 1000 + lk)
    q = bf(torch.randn(b, lq, h, 128, generator=g))
    k = bf(torch.randn(b, lk, h, 128, generator=g))
    v = bf(torch.randn(b, lk, h, 128, generator=g))
    out = H.attn_fwd(q.to(dev()), k.to(dev()), v.to(dev()), kv_splits=splits)
    assert_bf16_close(out, _attn_ref(q, k, v), ulps=2.0, atol=6e-3, msg=f"split-KV attention S={splits}")
    one = H.attn_fwd(q.to(dev()), k.to(dev()), v.to(dev()), kv_splits=1)
    assert_bf16_close(out, one.float().cpu(), ulps=2.0, atol=4e-3, msg="split vs single pass")


@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_partial_calls_over_disjoint_key_sets_merge_to_the_full_softmax(H, prescaled):
    """flexam_attn_fwd_partial + flexam_attn_merge: the keys are visited as (local chunk), (chunks before), (chunks after) in
    separate calls with their own range counts -- the sequence-parallel local-chunk-first pattern -- and merged once."""
    g = torch.Generator().manual_seed(91)
    b, h, lq, lk = 2, 3, 300, 1456
    c = 128 ** -0.5 * math.log2(math.e)
    q0 = torch.randn(b, lq, h, 128, generator=g)
    q = bf(q0 * c) if prescaled else bf(q0)
    k = bf(torch.randn(b, lk, h, 128, generator=g))
    v = bf(torch.randn(b, lk, h, 128, generator=g))
    qd, kd, vd = q.to(dev()), k.to(dev()), v.to(dev())
    lo, hi = 364, 728                                          # the "local" chunk (not tile aligned: 364 = 5.69 tiles)
    ws = H.attn_partial_workspace(b, h, lq, 8, dev())
    n = H.attn_fwd_partial(qd, kd[:, lo:hi], vd[:, lo:hi], ws, 0, 1, prescaled=prescaled)
    n += H.attn_fwd_partial(qd, kd[:, :lo], vd[:, :lo], ws, n, 1, prescaled=prescaled)
    n += H.attn_fwd_partial(qd, kd[:, hi:], vd[:, hi:], ws, n, 3, prescaled=prescaled)
    assert n == 1 + 1 + H.attn_effective_splits(lk - hi, 3)
    out = torch.empty(b, lq, h, 128, device=dev(), dtype=BF)
    H.attn_merge(out, ws, n, prescaled=prescaled)
    ref = _attn_ref(q / c if prescaled else q, k, v) if not prescaled else None
    one = H.attn_fwd(qd, kd, vd, prescaled=prescaled)
    assert_bf16_close(out, one.float().cpu(), ulps=2.0, atol=4e-3, msg="partial calls + merge vs single pass")
    if ref is not None:
        assert_bf16_close(out, ref, ulps=2.0, atol=6e-3, msg="partial calls + merge vs oracle")
    with pytest.raises(RuntimeError):                          # more slots than the workspace holds
        H.attn_fwd_partial(qd, kd, vd, ws, 7, 4)


def test_rmsnorm_rope_scatter_writes_the_exchange_layouts(H):
    """flexam_rmsnorm_rope_scatter: q, k normed + rotated and v copied straight into the sequence-parallel send layouts --
    [B, sp, lc, 3G] of the all-to-all mode and [B, lc, 2C] (k | v only) of the all-gather mode -- bit-identical to the in-place
    kernel followed by the pack copies it replaces."""
    from flexam_amd.rope import rope_tables
    g = torch.Generator().manual_seed(23)
    c, hd, sp, B, lc, tok0 = 512, 128, 2, 2, 37, 74
    G, L = c // sp, 148
    qkv = bf(torch.randn(B * lc, 3 * c, generator=g)).to(dev())
    wq, wk = (1 + 0.1 * torch.randn(c, generator=g)).to(dev()), (1 + 0.1 * torch.randn(c, generator=g)).to(dev())
    cos, sin = (t.to(dev()) for t in rope_tables((1, 1, L), L, hd))
    ref = qkv.clone()
    H.rmsnorm_rope(ref[:, 0:c], wq, ref[:, c:2 * c], wk, rope_cos=cos, rope_sin=sin, tokens_per_batch=lc, token_offset=tok0, head_dim=hd)
    W = 3 * G
    send = torch.zeros(B, sp, lc, W, device=dev(), dtype=BF)
    flat = send.view(-1)
    H.rmsnorm_rope_scatter(qkv[:, 0:c], wq, qkv[:, c:2 * c], wk, qkv[:, 2 * c:], flat, flat[G:], flat[2 * G:], ld_out=W, out_bs=sp * lc * W,
                           col_block=G, block_stride=lc * W, rope_cos=cos, rope_sin=sin, tokens_per_batch=lc, token_offset=tok0, head_dim=hd)
    want = ref.view(B, lc, 3, sp, G).permute(0, 3, 1, 2, 4).reshape(B, sp, lc, W)
    assert torch.equal(send, want)
    kv = torch.zeros(B, lc, 2 * c, device=dev(), dtype=BF)
    fk = kv.view(-1)
    H.rmsnorm_rope_scatter(None, None, qkv[:, c:2 * c], wk, qkv[:, 2 * c:], None, fk, fk[c:], ld_out=2 * c, out_bs=lc * 2 * c, col_block=c,
                           block_stride=0, rope_cos=cos, rope_sin=sin, tokens_per_batch=lc, token_offset=tok0, head_dim=hd)
    assert torch.equal(kv.view(B * lc, 2 * c), ref[:, c:])


def test_attention_tail_split_matches_single_pass(H):
    """Only the work units of the last, partial round are split: units before `split_from_unit` take the one-pass kernel."""
    g = torch.Generator().manual_seed(77)
    b, h, lq, lk = 2, 3, 700, 1300                       # 2 * 3 * 3 = 18 units
    q = bf(torch.randn(b, lq, h, 128, generator=g))
    k = bf(torch.randn(b, lk, h, 128, generator=g))
    v = bf(torch.randn(b, lk, h, 128, generator=g))
    ref = _attn_ref(q, k, v)
    for from_unit in (0, 5, 17):
        out = H.attn_fwd(q.to(dev()), k.to(dev()), v.to(dev()), kv_splits=2, split_from_unit=from_unit)
        assert_bf16_close(out, ref, ulps=2.0, atol=6e-3, msg=f"tail split from unit {from_unit}")


@pytest.mark.parametrize("b,h,lq,lk,splits,from_unit", [(2, 3, 700, 1300, 2, 5), (2, 3, 700, 1300, 3, 17), (1, 5, 2600, 900, 4, 9),
                                                          (1, 24, 1100, 1100, 3, 64), (2, 2, 257, 4100, 5, 3),
                                                          (1, 3, 700, 1300, 2, 8), (1, 3, 700, 1300, 2, 1), (1, 1, 256, 640, 2, 0)])
def test_attention_one_launch_tail_equals_two_launches(H, monkeypatch, b, h, lq, lk, splits, from_unit):
    """The default form of a split-KV call -- whole units and the split tail in ONE launch, an eighth of both kinds per XCD
    (AttnParams::whole_units) -- against the two-launch form (FLEXAM_ATTN_FUSED_TAIL=0): same workgroup programs on the same
    data, so bit-identical; unit counts that are not multiples of 8 leave idle workgroups in the padded grid."""
    g = torch.Generator().manual_seed(b * 1000 + lq)
    q = bf(torch.randn(b, lq, h, 128, generator=g)).to(dev())
    k = bf(torch.randn(b, lk, h, 128, generator=g)).to(dev())
    v = bf(torch.randn(b, lk, h, 128, generator=g)).to(dev())
    monkeypatch.setenv("FLEXAM_ATTN_FUSED_TAIL", "0")
    two = H.attn_fwd(q, k, v, kv_splits=splits, split_from_unit=from_unit).clone()
    monkeypatch.setenv("FLEXAM_ATTN_FUSED_TAIL", "1")
    one = H.attn_fwd(q, k, v, kv_splits=splits, split_from_unit=from_unit)
    assert torch.equal(one, two)
    assert_bf16_close(one.cpu(), _attn_ref(q.cpu(), k.cpu(), v.cpu()), ulps=2.0, atol=6e-3, msg="one-launch tail vs oracle")


def test_attention_split_plan():
    f = __import__("flexam_amd.hip", fromlist=["attn_split_plan"]).attn_split_plan
    assert f(48, 11648, 11648) == (3, 2048)    # one GPU: 2208 units = 8 full rounds + 160 units cut in 3
    assert f(24, 11648, 11648) == (3, 1024)    # CFG-parallel pair: 1104 units
    s8, from8 = f(24, 2912, 11648)
    assert s8 >= 4 and from8 == 256            # 8 GPUs: 288 units for 256 CUs
    assert f(48, 11648, 512) == (1, 2208)      # text cross-attention: 8 key tiles, never split
    assert f(16, 4096, 4096) == (1, 256)       # exactly one full round: nothing to split


def test_attention_strided_qkv_and_scale(H):
    """q, k, v as column slices of one fused [B, L, 3*H*128] projection buffer (the DiT layout)."""
    g = torch.Generator().manual_seed(3)
    b, l, h = 2, 320, 2
    qkv = bf(torch.randn(b, l, 3 * h * 128, generator=g) * 2.0)
    d = qkv.to(dev())
    views = [d[:, :, i * h * 128:(i + 1) * h * 128].unflatten(2, (h, 128)) for i in range(3)]
    out = H.attn_fwd(*views)
    q, k, v = (qkv[:, :, i * h * 128:(i + 1) * h * 128].unflatten(2, (h, 128)) for i in range(3))
    assert_bf16_close(out, _attn_ref(q, k, v), ulps=2.0, atol=6e-3, msg="attention strided")


@pytest.mark.parametrize("b,h,lq,lk,splits", [(1, 2, 300, 1111, None), (2, 1, 256, 640, 3), (1, 1, 77, 2055, 4)])
def test_attention_prescaled_q_matches_oracle(H, b, h, lq, lk, splits):
    """FLEXAM_ATTN_PRESCALED: q carries softmax_scale * log2(e) from its producer (one rounding to bf16 AFTER the multiply, as
    when the factor sits in the RMSNorm weight); the kernel then works on exp2(q.k) with the row reference inside the MFMA
    chain.  Oracle: softmax over the same rounded q, i.e. attention(q' / (scale log2 e), k, v).  Includes a late spike that
    forces the reference to move, and the split-KV merge."""
    g = torch.Generator().manual_seed(lq * 3 + lk)
    qf = torch.randn(b, lq, h, 128, generator=g)
    k = torch.randn(b, lk, h, 128, generator=g)
    v = torch.randn(b, lk, h, 128, generator=g)
    k[0, lk - 40, 0] = qf[0, 5, 0] * 4.0
    c = 128 ** -0.5 * 1.4426950408889634
    qs, k, v = bf(qf * c), bf(k), bf(v)
    kw = {} if splits is None else dict(kv_splits=splits)
    out = H.attn_fwd(qs.to(dev()), k.to(dev()), v.to(dev()), prescaled=True, **kw)
    assert_bf16_close(out, _attn_ref(qs.float() / c, k, v), ulps=2.0, atol=6e-3, msg="attention, pre-scaled q")


def test_attention_online_softmax_rescale_spike(H):
    """Force the running max to jump at a late key tile (cdna guide rule 26): one key aligned with
    one query and scaled up, placed in the 5th tile; rows that see it must still be exact."""
    g = torch.Generator().manual_seed(4)
    l = 512
    q = torch.randn(1, l, 1, 128, generator=g)
    k = torch.randn(1, l, 1, 128, generator=g)
    v = torch.randn(1, l, 1, 128, generator=g)
    k[0, 300, 0] = q[0, 17, 0] * 3.0
    k[0, 450, 0] = q[0, 200, 0] * 5.0
    q, k, v = bf(q), bf(k), bf(v)
    out = H.attn_fwd(q.to(dev()), k.to(dev()), v.to(dev()))
    assert_bf16_close(out, _attn_ref(q, k, v), ulps=2.0, atol=6e-3, msg="attention spike")


def test_attention_linearity_in_v_full_size(H):
    """Size-independent property at the BASELINE shape (L = 11648, one head pair): attention is
    linear in V, and a constant V gives that constant back."""
    g = torch.Generator().manual_seed(6)
    l, h = 11648, 2
    q = bf(torch.randn(1, l, h, 128, generator=g)).to(dev())
    k = bf(torch.randn(1, l, h, 128, generator=g)).to(dev())
    ones = torch.ones(1, l, h, 128, dtype=BF, device=dev())
    out = H.attn_fwd(q, k, ones)
    torch.testing.assert_close(out.float(), torch.ones_like(out).float(), rtol=0, atol=2.0 ** -7)
    v1 = bf(torch.randn(1, l, h, 128, generator=g)).to(dev())
    o1 = H.attn_fwd(q, k, v1).float()
    o2 = H.attn_fwd(q, k, (v1.float() * 2).to(BF)).float()
    torch.testing.assert_close(o2, 2 * o1, rtol=2.0 ** -6, atol=1e-3)
    # spot-check 64 query rows against the oracle
    rows = torch.arange(0, l, l // 64)[:64]
    ref = _attn_ref(q[:, rows].cpu(), k.cpu(), v1.cpu())
    assert_bf16_close(o1[:, rows], ref, ulps=2.0, atol=4e-3, msg="attention full size rows")


def test_attention_prescaled_full_size_properties_and_rows(H):
    """The DiT's form of the kernel at the BASELINE shape (L = 11648, B = 2 x 24 heads as in one self-attention call, so that the
    tail split and its merge run): constant V comes back, and 64 spot-checked query rows per batch row match the oracle evaluated on
    the same pre-scaled, once-rounded q."""
    g = torch.Generator().manual_seed(8)
    l, h = 11648, 24
    c = 128 ** -0.5 * 1.4426950408889634
    qf = torch.randn(2, l, h, 128, generator=g)
    qs = bf(qf * c).to(dev())
    k = bf(torch.randn(2, l, h, 128, generator=g)).to(dev())
    ones = torch.ones(2, l, h, 128, dtype=BF, device=dev())
    out = H.attn_fwd(qs, k, ones, prescaled=True)
    torch.testing.assert_close(out.float(), torch.ones_like(out).float(), rtol=0, atol=2.0 ** -7)
    v = bf(torch.randn(2, l, h, 128, generator=g)).to(dev())
    o = H.attn_fwd(qs, k, v, prescaled=True).float()
    rows = torch.arange(0, l, l // 64)[:64]
    heads = [0, 11, 23]
    ref = _attn_ref(qs[:, rows][:, :, heads].float().cpu() / c, k[:, :, heads].cpu(), v[:, :, heads].cpu())
    assert_bf16_close(o[:, rows][:, :, heads], ref, ulps=2.0, atol=4e-3, msg="pre-scaled attention, full size rows")


def test_attention_prescaled_rows_at_the_704x1280_shape(H):
    """BASELINE configs[4]'s shape (97x704x1280 -> L = 22880 tokens, 358 key tiles, 90 q blocks per head: 4320 work units = 16 full rounds
    + 224 units whose keys are cut): constant V comes back and spot-checked rows of three heads match the oracle; the (batch, head) K/V
    panel spans 22880 x 3072 x 2 B x ... < 2 GiB, inside the kernel's 32-bit tile offsets."""
    g = torch.Generator().manual_seed(9)
    l, h = 22880, 24
    c = 128 ** -0.5 * 1.4426950408889634
    qs = bf(torch.randn(2, l, h, 128, generator=g) * c).to(dev())
    k = bf(torch.randn(2, l, h, 128, generator=g)).to(dev())
    ones = torch.ones(2, l, h, 128, dtype=BF, device=dev())
    out = H.attn_fwd(qs, k, ones, prescaled=True)
    torch.testing.assert_close(out.float(), torch.ones_like(out).float(), rtol=0, atol=2.0 ** -7)
    del ones, out
    v = bf(torch.randn(2, l, h, 128, generator=g)).to(dev())
    o = H.attn_fwd(qs, k, v, prescaled=True).float()
    rows = torch.cat([torch.arange(0, l, l // 40)[:40], torch.tensor([l - 1, l - 33, l - 97])])     # incl. the last, partial q block
    heads = [0, 13, 23]
    ref = _attn_ref(qs[:, rows][:, :, heads].float().cpu() / c, k[:, :, heads].cpu(), v[:, :, heads].cpu())
    assert_bf16_close(o[:, rows][:, :, heads], ref, ulps=2.0, atol=4e-3, msg="pre-scaled attention, L = 22880 rows")


# ----------------------------------------------------------------------------- row kernels
def test_ln_modulate_two_row_table(H):
    from oracle import dit as O
    g = torch.Generator().manual_seed(20)
    m, c = 333, 3072
    x = torch.randn(m, c, generator=g) * 2 + 0.5
    table = torch.randn(4, 2, c, generator=g) * 0.5           # [rows][shift|scale]
    rows = torch.randint(0, 4, (m,), generator=g, dtype=torch.int32)
    out = H.ln_modulate(x.to(dev()), shift=table.to(dev())[:, 0], scale=table.to(dev())[:, 1], row_index=rows.to(dev()))
    want = O.layer_norm(x, 1e-6) * table[rows.long(), 1] + table[rows.long(), 0]
    assert_bf16_close(out, want, ulps=1.0, atol=1e-6, msg="ln_modulate")
    # affine form (norm3) on a narrower row, batch-indexed table form
    x2 = torch.randn(64, 256, generator=g)
    w, b = torch.randn(256, generator=g), torch.randn(256, generator=g)
    out2 = H.ln_modulate(x2.to(dev()), ln_w=w.to(dev()), ln_b=b.to(dev()))
    assert_bf16_close(out2, O.layer_norm(x2, 1e-6, w, b), ulps=1.0, atol=1e-6, msg="ln affine")
    t2 = torch.randn(2, 2, 256, generator=g)
    out3 = H.ln_modulate(x2.to(dev()), shift=t2.to(dev())[:, 0], scale=t2.to(dev())[:, 1], rows_per_batch=32)
    idx = torch.arange(64) // 32
    assert_bf16_close(out3, O.layer_norm(x2, 1e-6) * t2[idx, 1] + t2[idx, 0], ulps=1.0, atol=1e-6, msg="ln batch rows")


def test_gate_residual(H):
    g = torch.Generator().manual_seed(21)
    m, c = 257, 3072
    x = torch.randn(m, c, generator=g)
    y = bf(torch.randn(m, c, generator=g))
    gate = torch.randn(3, c, generator=g)
    rows = torch.randint(0, 3, (m,), generator=g, dtype=torch.int32)
    xd = x.clone().to(dev())
    H.gate_residual(xd, y.to(dev()), gate.to(dev()), rows.to(dev()))
    torch.testing.assert_close(xd.cpu(), x + y.float() * gate[rows.long()], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c,heads", [(256, 2), (3072, 24)])
def test_rmsnorm_rope_matches_oracle(H, c, heads):
    from oracle import dit as O
    from flexam_amd.rope import rope_tables
    g = torch.Generator().manual_seed(22)
    grid = (3, 4, 6)
    l = grid[0] * grid[1] * grid[2] + 5                      # 5 pass-through tokens
    b = 2
    q = bf(torch.randn(b, l, c, generator=g) * 1.5)
    k = bf(torch.randn(b, l, c, generator=g))
    wq, wk = 1 + 0.1 * torch.randn(c, generator=g), 1 + 0.1 * torch.randn(c, generator=g)
    cos, sin = rope_tables(grid, l, 128)
    qd, kd = q.clone().to(dev()).view(b * l, c), k.clone().to(dev()).view(b * l, c)
    H.rmsnorm_rope(qd, wq.to(dev()), kd, wk.to(dev()), rope_cos=cos.to(dev()), rope_sin=sin.to(dev()), tokens_per_batch=l)
    ang = O.rope_angles(1024, 128)
    for got, x, w in ((qd, q, wq), (kd, k, wk)):
        want = O.rope_apply(O.rms_norm(x.float(), w, 1e-6).view(b, l, heads, 128), grid, ang).reshape(b * l, c)
        assert_bf16_close(got, want, ulps=1.0, atol=1e-6, msg="rmsnorm_rope")
    # no-rope, q only (cross-attention form)
    q2 = q.clone().to(dev()).view(b * l, c)
    H.rmsnorm_rope(q2, wq.to(dev()))
    assert_bf16_close(q2, O.rms_norm(q.float(), wq, 1e-6).view(b * l, c), ulps=1.0, atol=1e-6, msg="rmsnorm")


def test_mod_table_small_linear_sinusoid(H):
    from oracle import dit as O
    g = torch.Generator().manual_seed(23)
    nblk, R, C, B = 3, 4, 256, 2
    mod, e = torch.randn(nblk, 6, C, generator=g), torch.randn(R, 6, C, generator=g)
    mdens, dens = torch.randn(nblk, 2, C, generator=g), torch.randn(B, 2, C, generator=g)
    out = torch.empty(nblk, R, 6, C, device=dev())
    H.mod_table(mod.to(dev()), e.to(dev()), out, rows_per_batch=R // B, scale_mask=0b010010, mdens=mdens.to(dev()),
                dens=dens.to(dev()), dens_slots=0xFF1FF0)
    want = mod[:, None] + e[None]
    want[:, :, 1] += 1
    want[:, :, 4] += 1
    bidx = torch.arange(R) // (R // B)
    want[:, :, 0] += mdens[:, None, 0] + dens[bidx, 0][None]
    want[:, :, 3] += mdens[:, None, 1] + dens[bidx, 1][None]
    torch.testing.assert_close(out.cpu(), want, rtol=1e-6, atol=1e-6)

    t = torch.tensor([0.0, 24.4, 500.0, 1000.0])
    torch.testing.assert_close(H.sinusoid_embed(t.to(dev()), 256).cpu(), O.sinusoidal_embedding_1d(256, t).float(), rtol=0, atol=1e-6)

    x = torch.randn(4, 256, generator=g)
    for wdt in (torch.float32, BF):
        w = (torch.randn(768, 256, generator=g) / 16).to(wdt)
        b = torch.randn(768, generator=g)
        y = H.small_linear(x.to(dev()), w.to(dev()), b.to(dev()), silu_in=True)
        torch.testing.assert_close(y.cpu(), F.linear(F.silu(x), w.float(), b), rtol=1e-5, atol=1e-5)


def test_patchify_gemm_equals_conv3d_and_unpatchify(H):
    from oracle import dit as O
    g = torch.Generator().manual_seed(24)
    c, f, h, w, d = 20, 3, 8, 12, 256
    x = bf(torch.randn(c, f, h, w, generator=g))
    wt = bf(torch.randn(d, c, 1, 2, 2, generator=g) / math.sqrt(c * 4))
    bias = torch.randn(d, generator=g)
    kpad = 128                                                  # c*4 = 80 -> padded K
    a = torch.zeros(f * (h // 2) * (w // 2), kpad, dtype=BF, device=dev())
    H.patchify(x.float().to(dev()), a)
    wpad = torch.zeros(d, kpad, dtype=BF)
    wpad[:, : c * 4] = wt.flatten(1)
    out = H.gemm(a, wpad.to(dev()), bias.to(dev()), out_dtype=torch.float32)
    want = F.conv3d(x.float()[None], wt.float(), bias, stride=(1, 2, 2))[0].flatten(1).t()
    torch.testing.assert_close(out.cpu(), want, rtol=1e-4, atol=1e-4)

    cc = 48
    tok = torch.randn(5 + f * (h // 2) * (w // 2), 4 * cc, generator=g)
    got = H.unpatchify(tok.to(dev()), 5, cc, f, h, w)
    torch.testing.assert_close(got.cpu(), O.unpatchify(tok[5:], (f, h // 2, w // 2), (1, 2, 2), cc), rtol=0, atol=0)


def test_cfg_euler_blend(H):
    from oracle import dit as O
    g = torch.Generator().manual_seed(25)
    c, f, h, w = 48, 3, 8, 12
    n = f * (h // 2) * (w // 2)
    tu, tc = torch.randn(7 + n, 4 * c, generator=g), torch.randn(7 + n, 4 * c, generator=g)
    lat, known = torch.randn(c, f, h, w, generator=g), torch.randn(c, f, h, w, generator=g)
    mask = torch.ones(f, h, w)
    mask[0] = 0
    mask[1, :2] = 0.25
    ld = lat.clone().to(dev())
    H.cfg_euler_blend(tu.to(dev()), tc.to(dev()), 7, 6.0, -0.0371, ld, known.to(dev()), mask.to(dev()))
    vu = O.unpatchify(tu[7:], (f, h // 2, w // 2), (1, 2, 2), c)
    vc = O.unpatchify(tc[7:], (f, h // 2, w // 2), (1, 2, 2), c)
    x = lat + (-0.0371) * (vu + 6.0 * (vc - vu))
    want = (1 - mask) * known + mask * x
    torch.testing.assert_close(ld.cpu(), want, rtol=1e-5, atol=1e-5)


def test_small_linear_many_rows_is_row_batch_independent(H):
    """Foreground masks with soft edges give hundreds of distinct per-token timesteps: rows are embedded 32 per pass over the weight
    (4 outputs per wave).  Every output must be BIT-identical to the 8-row instance (same per-lane k order, same wave reduction),
    so a timestep's embedding does not depend on how many other timesteps the step has."""
    g = torch.Generator().manual_seed(231)
    x = torch.randn(29, 256, generator=g)
    for wdt, n in ((torch.float32, 771), (BF, 768)):                     # 771: the last wave's 4 outputs run past N
        w = (torch.randn(n, 256, generator=g) / 16).to(wdt)
        b = torch.randn(n, generator=g)
        wide = H.small_linear(x.to(dev()), w.to(dev()), b.to(dev()), silu_in=True)
        torch.testing.assert_close(wide.cpu(), F.linear(F.silu(x), w.float(), b), rtol=1e-5, atol=1e-5)
        narrow = torch.cat([H.small_linear(x[i:i + 8].to(dev()), w.to(dev()), b.to(dev()), silu_in=True) for i in range(0, 29, 8)])
        assert torch.equal(wide, narrow)


@pytest.mark.parametrize("c,f,h,w,tok0,cfg", [(48, 25, 32, 56, 448, True), (48, 3, 8, 12, 7, False), (16, 2, 4, 6, 0, True)])
def test_cfg_euler_blend_tiled_and_gather_forms_agree(H, c, f, h, w, tok0, cfg):
    """The LDS-tiled form (token rows read once, 16 B per lane) against the oracle at the config-2 latent shape, and against the
    gather form the library falls back to when the token rows are not 16-byte aligned (bit-identical: same arithmetic per element)."""
    from oracle import dit as O
    g = torch.Generator().manual_seed(26)
    n = f * (h // 2) * (w // 2)
    tu, tc = torch.randn(tok0 + n, 4 * c, generator=g), torch.randn(tok0 + n, 4 * c, generator=g)
    lat, known = torch.randn(c, f, h, w, generator=g), torch.randn(c, f, h, w, generator=g)
    mask = torch.rand(f, h, w, generator=g)
    mask[0] = 0
    ld = lat.clone().to(dev())
    H.cfg_euler_blend(tu.to(dev()), tc.to(dev()) if cfg else None, tok0, 6.0, -0.0371, ld, known.to(dev()), mask.to(dev()))
    vu = O.unpatchify(tu[tok0:], (f, h // 2, w // 2), (1, 2, 2), c)
    v = vu + 6.0 * (O.unpatchify(tc[tok0:], (f, h // 2, w // 2), (1, 2, 2), c) - vu) if cfg else vu
    want = (1 - mask) * known + mask * (lat + (-0.0371) * v)
    torch.testing.assert_close(ld.cpu(), want, rtol=1e-5, atol=1e-5)
    # misaligned token rows (row pitch 4C + 1 floats) take the gather kernel
    pad = lambda t: torch.cat([t, torch.zeros(t.shape[0], 1)], dim=1).to(dev())[:, :4 * c]
    ld2 = lat.clone().to(dev())
    H.cfg_euler_blend(pad(tu), pad(tc) if cfg else None, tok0, 6.0, -0.0371, ld2, known.to(dev()), mask.to(dev()))
    assert torch.equal(ld, ld2)
    vel = torch.empty(c, f, h, w, device=dev())
    H.cfg_velocity(tu.to(dev()), tc.to(dev()) if cfg else None, tok0, 6.0, vel)
    torch.testing.assert_close(vel.cpu(), v, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("lk,mult,prescaled", [(100, 29, True), (96, 417, True), (127, 386, False), (1, 512, True), (65, 2, False)])
def test_attention_weighted_last_key_equals_explicit_copies(H, lk, mult, prescaled):
    """flexam_attn_fwd_lastkey: softmax over {k_0 .. k_{Lk-2}, N copies of k_{Lk-1}} must equal plain attention over a context that
    really holds the N copies (what the reference's zero-padded text context is behind the text MLP); Lk a multiple of 32 takes
    the branch of the half tile that ends exactly at the last key; both score forms (q pre-scaled / scale in the kernel)."""
    g = torch.Generator().manual_seed(77 + lk)
    B, Lq, Hh, D = 2, 300, 2, 128
    scale = D ** -0.5
    q = bf(torch.randn(B, Lq, Hh, D, generator=g))
    k = bf(torch.randn(B, lk, Hh, D, generator=g))
    v = bf(torch.randn(B, lk, Hh, D, generator=g))
    k_full = torch.cat([k, k[:, -1:].expand(B, mult - 1, Hh, D)], dim=1).contiguous()
    v_full = torch.cat([v, v[:, -1:].expand(B, mult - 1, Hh, D)], dim=1).contiguous()
    qd = bf(q.float() * scale * math.log2(math.e)).to(dev()) if prescaled else q.to(dev())
    got = H.attn_fwd_lastkey(qd, k.to(dev()), v.to(dev()), float(mult), prescaled=prescaled)
    ref = H.attn_fwd(qd, k_full.to(dev()), v_full.to(dev()), prescaled=prescaled)
    # two bf16 outputs of two summation orders: a value next to a rounding boundary may land one bf16 step (2^-7 relative) apart
    assert_bf16_close(got, ref, ulps=4.0, atol=2e-3, msg=f"lastkey lk={lk} x{mult}")
    rel = ((got.float() - ref.float()).pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt()).item()
    assert rel < 5e-3, rel          # two independent bf16 roundings of the output: ~2e-3 by themselves
    from oracle import dit as O
    if not prescaled:
        want = O.attention(q.float(), k_full.float(), v_full.float())
        assert_bf16_close(got, want, ulps=4.0, atol=4e-3, msg="lastkey vs oracle")


@pytest.mark.parametrize("b,h,l,splits", [(1, 2, 1088, None), (1, 2, 1111, None), (2, 1, 1025, None), (1, 1, 2055, (4, 0)), (1, 2, 1500, (3, 7)),
                                          (1, 1, 1344, (21, 0))])
def test_attention_one_block_step_instance_equals_the_general_instance(H, b, h, l, splits):
    """The instance the DiT's self-attention runs (pre-scaled q, Lk > 1024, no weighted key): no key mask in the main loop, a key
    count that is not a multiple of 64 handled by SHIFTING the last tile to the window [Lk - 64, Lk) and masking the keys the tile
    before already held.  Compared with the oracle and, at bf16 precision, with the general instance (FLEXAM_ATTN_FULL=0).  Split key ranges (the shifted tile in the last range only; one tile per range at 21 ranges) included."""
    import os
    g = torch.Generator().manual_seed(l + b)
    c = 128 ** -0.5 * 1.4426950408889634
    qf = torch.randn(b, l, h, 128, generator=g)
    k = torch.randn(b, l, h, 128, generator=g)
    v = torch.randn(b, l, h, 128, generator=g)
    k[0, l - 3, 0] = qf[0, 5, 0] * 4.0                  # a spike inside the shifted tile: the reference moves there
    qs, k, v = bf(qf * c), bf(k), bf(v)
    kw = {} if splits is None else dict(kv_splits=splits[0], split_from_unit=splits[1])
    out = H.attn_fwd(qs.to(dev()), k.to(dev()), v.to(dev()), prescaled=True, **kw)
    os.environ["FLEXAM_ATTN_FULL"] = "0"
    try:
        gen = H.attn_fwd(qs.to(dev()), k.to(dev()), v.to(dev()), prescaled=True, **kw)
    finally:
        os.environ.pop("FLEXAM_ATTN_FULL")
    assert_bf16_close(out, _attn_ref(qs.float() / c, k, v), ulps=2.0, atol=6e-3, msg="attention, one-block-step instance")
    # not bit for bit: the instance sets its first reference in the prologue (the second half tile's chain then starts from -ref
    # instead of being shifted afterwards), and a ragged count changes the tile a key sits in; bf16-level agreement
    torch.testing.assert_close(out.float(), gen.float(), rtol=2.0 ** -6, atol=6e-3)


@pytest.mark.parametrize("h,lq,lk,mult", [(3, 25600, 127, 386.0), (5, 20000, 200, 1.0), (2, 40000, 64, 1.0)])
def test_short_context_instance_walks_units_and_heads(H, h, lq, lk, mult):
    """The text cross-attention's instance (at most 4 key tiles, pre-scaled q): one persistent workgroup per CU walks a contiguous
    range of (q block, head) units with the head's K/V tiles resident, reloading them where its range crosses into the next head.
    More units than CUs here, so workgroups walk several units and some cross a head boundary; ragged last q block; 2, 4 and 1 key
    tiles; a weighted last key.  Bit-equal to the one-workgroup-per-q-block path (FLEXAM_ATTN_SHORT=0), and sampled rows vs the oracle."""
    import os
    g = torch.Generator().manual_seed(lq + lk)
    c = 128 ** -0.5 * 1.4426950408889634
    q = bf(torch.randn(1, lq, h, 128, generator=g) * c).to(dev())
    k = bf(torch.randn(1, lk, h, 128, generator=g)).to(dev())
    v = bf(torch.randn(1, lk, h, 128, generator=g)).to(dev())
    run = (lambda: H.attn_fwd_lastkey(q, k, v, mult, prescaled=True)) if mult != 1.0 else (lambda: H.attn_fwd(q, k, v, prescaled=True))
    out = run()
    os.environ["FLEXAM_ATTN_SHORT"] = "0"
    try:
        gen = run()
    finally:
        os.environ.pop("FLEXAM_ATTN_SHORT")
    assert torch.equal(out, gen)
    rows = torch.arange(0, lq, lq // 50)[:50]
    kk, vv = k.cpu(), v.cpu()
    if mult != 1.0:                                  # the weighted last key as explicit copies
        n = int(mult) - 1
        kk = torch.cat([kk, kk[:, -1:].expand(1, n, h, 128)], dim=1)
        vv = torch.cat([vv, vv[:, -1:].expand(1, n, h, 128)], dim=1)
    ref = _attn_ref(q[:, rows].float().cpu() / c, kk, vv)
    assert_bf16_close(out[:, rows], ref, ulps=2.0, atol=4e-3, msg="short-context instance")
