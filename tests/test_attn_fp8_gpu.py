"""The quantised self-attention (MXFP8 operands, csrc/attn_fp8.inc) behind VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION -- the variant the
reference reaches through the third-party `sageattn` (FlexAM/models/attention_utils.py:195-203).  Its contract is softmax attention
within a tolerance; the tolerances below are the e4m3 format's: every operand element carries a relative rounding error of up to 2^-4
(3.6 % rms), so on unit-variance random logits -- where the output is an average of noise and nothing adds up coherently -- the
output differs from fp32 attention by 5 % rms; the bf16 kernel by 0.2 %."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634


@pytest.fixture(scope="module")
def H():
    from flexam_amd import hip
    hip.load_library()
    return hip


def _ref(q, k, v):                     # q in exp2 units
    s = torch.einsum("blhd,bmhd->bhlm", q.float(), k.float()) * math.log(2.0)
    return torch.einsum("bhlm,bmhd->blhd", torch.softmax(s, dim=-1), v.float())


def _inputs(B, Hh, L, seed, sharp=1.0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, L, Hh, 128, generator=g) * (128 ** -0.5 * LOG2E * sharp)
    k = torch.randn(B, L, Hh, 128, generator=g)
    v = torch.randn(B, L, Hh, 128, generator=g)
    return [t.to(torch.bfloat16).cuda() for t in (q, k, v)]


@pytest.mark.parametrize("B,Hh,L,splits", [(1, 1, 40, None), (1, 1, 64, None), (1, 2, 256, None), (1, 1, 300, None), (2, 3, 1111, None), (1, 2, 1024, (2, 0)),
                                           (1, 2, 1024, (4, 5)), (1, 1, 2912, None)])
def test_mxfp8_attention_against_fp32_attention(H, B, Hh, L, splits):
    """Whole and ragged tiles, a main-loop pass plus tail tiles, split key ranges with the merge launch."""
    q, k, v = _inputs(B, Hh, L, L)
    bufs = H.attn_fp8_pack(q, k, v)
    kw = {} if splits is None else dict(kv_splits=splits[0], split_from_unit=splits[1])
    o = H.attn_fwd_fp8(bufs, L, **kw).float()
    want = _ref(q, k, v)
    rel = float((o - want).norm() / want.norm())
    print(B, Hh, L, splits, "rel-RMS", rel)
    assert torch.isfinite(o).all() and rel <= 6.5e-2


def test_mxfp8_attention_follows_a_moving_row_maximum(H):
    """Keys sorted so that the row maxima keep growing along the key axis: the reference of the online softmax moves many times
    (every move rescales O, the row sum and the packed e4m3 P of the tile in flight); sharp logits, so a stale factor would show."""
    q, k, v = _inputs(1, 2, 1536, 7, sharp=3.0)
    order = torch.argsort((k.float() * q.float().mean(dim=1, keepdim=True)).sum(-1), dim=1)      # ascending mean score per head
    k = torch.gather(k, 1, order.unsqueeze(-1).expand_as(k)).contiguous()
    v = torch.gather(v, 1, order.unsqueeze(-1).expand_as(v)).contiguous()
    o = H.attn_fwd_fp8(H.attn_fp8_pack(q, k, v), 1536).float()
    want = _ref(q, k, v)
    rel = float((o - want).norm() / want.norm())
    print("rel-RMS", rel)
    assert torch.isfinite(o).all() and rel <= 1e-1


def test_mxfp8_attention_with_one_dominant_key_returns_its_value_row(H):
    """One key per query dominates (scores 40 exp2-units above the rest): the output must be that key's V row to e4m3 precision --
    checks the Q / K channel maps, the key order of the transposed V operand and both kinds of scale on exact structure."""
    B, Hh, L = 1, 2, 512
    g = torch.Generator().manual_seed(3)
    k = torch.randn(B, L, Hh, 128, generator=g)
    v = torch.randn(B, L, Hh, 128, generator=g)
    perm = torch.randperm(L, generator=g)
    q = k[:, perm] * (40.0 / 128.0)                 # q_i . k_perm(i) ~ 40, q_i . k_j ~ N(0, 3.5)
    q, k, v = (t.to(torch.bfloat16).cuda() for t in (q, k, v))
    o = H.attn_fwd_fp8(H.attn_fp8_pack(q, k, v), L).float()
    want = v.float()[:, perm.cuda()]
    rel = float((o - want).norm() / want.norm())
    print("rel-RMS vs the selected V rows", rel)
    assert rel <= 4.5e-2                            # e4m3 rounding of V (3.6 % rms)


def test_attention_seam_takes_the_reference_switch(H):
    """flexam_amd.attention(..., attention_type="SAGE_ATTENTION") and the environment form, as the reference's seam takes them."""
    from flexam_amd.attention_utils import attention
    q, k, v = _inputs(1, 2, 384, 11)
    want = attention(q, k, v)                                          # bf16 kernel
    with torch.no_grad():
        got = attention(q, k, v, attention_type="SAGE_ATTENTION")
        os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"
        try:
            got_env = attention(q, k, v)
        finally:
            os.environ.pop("VIDEOX_ATTENTION_TYPE")
    assert torch.equal(got, got_env) and not torch.equal(got, want)
    rel = float((got.float() - want.float()).norm() / want.float().norm())
    assert rel <= 6.5e-2
    with torch.enable_grad():                                           # the reference falls back to flash attention under grad
        assert torch.equal(attention(q, k, v, attention_type="SAGE_ATTENTION"), want)
    cross = attention(q, k[:, :77], v[:, :77], attention_type="SAGE_ATTENTION")      # Lq != Lk: the bf16 kernel
    assert torch.equal(cross, attention(q, k[:, :77], v[:, :77]))


def _mx_quant(x32):
    """OCP MX block quantisation as the pack kernel does it: x32 [..., 32] fp32 -> (e4m3 bytes [..., 32] uint8, E8M0 byte [...] uint8),
    scale = the smallest power of two with amax / scale <= 448."""
    amax = x32.abs().amax(dim=-1)
    t = (amax / 448.0).float()
    bits = t.view(torch.int32)
    e = ((bits >> 23) & 255) + ((bits & 0x7FFFFF) != 0).int()
    e = e.clamp(1, 253)
    inv = ((254 - e) << 23).view(torch.float32)
    q = (x32 * inv.unsqueeze(-1)).to(torch.float8_e4m3fn).view(torch.uint8)
    return q, e.to(torch.uint8)


def test_pack_kernel_bytes_scales_and_layouts_exactly(H):
    """flexam_attn_fp8_pack against a torch restatement, byte for byte: e4m3 rounding and E8M0 scales of Q and K per 32 channels, of V
    per channel and 32-key half tile; the Q rows, the swizzled K image, the transposed V image in the key order the P.V operand needs
    (byte j of lane half h = key 32 (j / 16) + 8 ((j % 16) / 4) + 4 h + j % 4), zero rows past L."""
    B, Hh, L = 1, 2, 150                       # 3 tiles, the last one ragged (22 keys)
    q, k, v = _inputs(B, Hh, L, 5)
    k[0, 7, 1, 40:48] *= 37.0                  # an outlier block
    q8, qs, kv8 = H.attn_fp8_pack(q, k, v)
    q8, qs, kv8 = q8.cpu(), qs.cpu(), kv8.cpu()
    Lp, T = 256, 3
    pad = lambda t: torch.cat([t.float().cpu(), torch.zeros(B, T * 64 - L, Hh, 128)], dim=1)      # [B, 192, H, 128]
    qf, kf, vf = pad(q), pad(k), pad(v)
    # Q: [B][H][Lp][128] bytes, scales as 4 bytes per row
    wq, sq = _mx_quant(qf.permute(0, 2, 1, 3).reshape(B, Hh, T * 64, 4, 32))
    assert torch.equal(q8[:, :, :T * 64].reshape(B, Hh, T * 64, 4, 32), wq)
    assert torch.equal(qs.view(torch.uint8).reshape(B, Hh, Lp, 4)[:, :, :T * 64], sq)
    assert int(q8[:, :, T * 64:].abs().max()) == 0
    # K image: row `r` of a tile, 16-byte chunk c at position c ^ ((r >> 1) & 7); scales at 16384 + 4 r + block
    wk, sk = _mx_quant(kf.permute(0, 2, 1, 3).reshape(B, Hh, T, 64, 4, 32))
    wk = wk.reshape(B, Hh, T, 64, 8, 16)
    rec = kv8.reshape(B, Hh, T, -1)
    for r in range(64):
        for c in range(8):
            pos = c ^ ((r >> 1) & 7)
            assert torch.equal(rec[..., r * 128 + 16 * pos: r * 128 + 16 * pos + 16], wk[:, :, :, r, c]), (r, c)
    assert torch.equal(rec[..., 16384:16384 + 256].reshape(B, Hh, T, 64, 4), sk)
    # V^T image: row d, chunk (2 b + h) at position (2 b + h) ^ ((d >> 2) & 3), byte e = key 32 b + (e & 3) + 8 (e >> 2) + 4 h
    vt = vf.permute(0, 2, 1, 3).reshape(B, Hh, T, 64, 128)                   # [.., key, d]
    keys = torch.tensor([[32 * b + (e & 3) + 8 * (e >> 2) + 4 * h for h in range(2) for e in range(16)] for b in range(2)])    # [b][16 h + e]
    blocks = vt[:, :, :, keys, :].permute(0, 1, 2, 5, 3, 4)                  # [B, H, T, d, b, 32]
    wv, sv = _mx_quant(blocks)
    for d in range(128):
        for b in range(2):
            for h in range(2):
                pos = (2 * b + h) ^ ((d >> 2) & 3)
                got = rec[..., 8192 + d * 64 + 16 * pos: 8192 + d * 64 + 16 * pos + 16]
                assert torch.equal(got, wv[:, :, :, d, b, 16 * h:16 * h + 16]), (d, b, h)
    assert torch.equal(rec[..., 16640:16640 + 256].reshape(B, Hh, T, 128, 2), sv)


def test_fused_rmsnorm_rope_writes_the_same_operands_as_the_two_step_path(H):
    """flexam_rmsnorm_rope_mx (RMSNorm + RoPE written as MXFP8 Q rows / K image directly, quantised from fp32) against
    flexam_rmsnorm_rope + flexam_attn_fp8_pack (quantised from the bf16 intermediate): same scale bytes except where bf16's rounding
    moves a block maximum across a power of two, values within one e4m3 step -- and the attention that reads them agrees with the
    two-step path at the format's noise level.  Ragged token count (padding rows stay zero), token offset, two samples."""
    B, L, nh = 2, 200, 24
    g = torch.Generator().manual_seed(21)
    qkv = (torch.randn(B * L, 3 * nh * 128, generator=g) * 1.5).to(torch.bfloat16).cuda()
    wq = (torch.rand(nh * 128, generator=g) * 0.2 + 0.05).cuda()
    wk = (torch.rand(nh * 128, generator=g) + 0.5).cuda()
    ang = torch.rand(L + 5, 64, generator=g) * 6.28
    cos, sin = ang.cos().cuda(), ang.sin().cuda()
    d = nh * 128
    q2, k2 = qkv[:, 0:d].clone(), qkv[:, d:2 * d].clone()
    H.rmsnorm_rope(q2, wq, k2, wk, rope_cos=cos, rope_sin=sin, tokens_per_batch=L, token_offset=5, head_dim=128)
    v4 = qkv.view(B, L, 3 * d)[:, :, 2 * d:].unflatten(2, (nh, 128))
    two = H.attn_fp8_pack(q2.view(B, L, nh, 128), k2.view(B, L, nh, 128), v4)
    one = H.attn_fp8_buffers(B, nh, L, qkv.device)
    H.rmsnorm_rope_mx(qkv[:, 0:d], wq, qkv[:, d:2 * d], wk, one, cos, sin, L, 5)
    H.attn_fp8_pack(None, None, v4, one)
    # scale bytes of the valid rows: equal or one apart (padding rows: the pack kernel writes scale byte 1 for its zero rows, the
    # fused kernel leaves the zero the buffers are created with); the V half identical
    qa, qb = one[1].view(torch.uint8).reshape(B, nh, 256, 4)[:, :, :L], two[1].view(torch.uint8).reshape(B, nh, 256, 4)[:, :, :L]
    diff = (qa.int() - qb.int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 0.02
    ra, rb = one[2].reshape(B, nh, 4, -1), two[2].reshape(B, nh, 4, -1)
    assert torch.equal(ra[..., 8192:16384], rb[..., 8192:16384]) and torch.equal(ra[..., 16640:16896], rb[..., 16640:16896])      # V image, V scales
    ka, kb = ra[..., 16384:16640].reshape(B, nh, 256, 4)[:, :, :L], rb[..., 16384:16640].reshape(B, nh, 256, 4)[:, :, :L]
    kd = (ka.int() - kb.int()).abs()
    assert int(kd.max()) <= 1 and float((kd != 0).float().mean()) < 0.02
    same = (ra[:, :, :3, :8192] == rb[:, :, :3, :8192]).float().mean()
    assert float(same) > 0.9                       # K bytes: identical except where the bf16 intermediate rounded differently
    assert int(one[0][:, :, L:].abs().max()) == 0 and int(ra[:, :, 3, 8 * 128:64 * 128].abs().max()) == 0      # padding rows untouched (zero)
    o1 = H.attn_fwd_fp8(one, L).float()
    o2 = H.attn_fwd_fp8(two, L).float()
    ref = H.attn_fwd(q2.view(B, L, nh, 128), k2.view(B, L, nh, 128), v4, prescaled=True).float()
    r1, r2 = float((o1 - ref).norm() / ref.norm()), float((o2 - ref).norm() / ref.norm())
    print("fused", r1, "two-step", r2)
    assert r1 <= 1.15 * r2 + 5e-3


@pytest.mark.parametrize("B,Hh,chunk,n_chunks,lk", [(1, 2, 256, 4, 1024), (2, 3, 128, 3, 300), (1, 1, 64, 5, 257), (1, 24, 2944, 4, 11648)])
def test_chunked_key_records_equal_the_contiguous_ones(H, B, Hh, chunk, n_chunks, lk):
    """flexam_attn_fwd_fp8_chunked (r6): the key / value records of a sequence cut into `n_chunks` chunks of `chunk` tokens (a multiple
    of 64), each packed on its own -- what every sequence-parallel rank does with its tokens -- and stacked rank-major, against ONE pack
    of the whole sequence: the records are the same bytes, so (1) all queries over the chunked records = the plain call's bits, and (2)
    the queries of ONE chunk (Lq != Lk, a rank's local rows) agree with the same rows of the plain call to the fp32 merge's rounding.
    `lk` real keys: the last chunks hold pads (rows of zeros here), masked by the key count; the last case is a rank of four at the
    bench's 11648 tokens (chunks of 46 tiles)."""
    lp = chunk * n_chunks
    q, k, v = _inputs(B, Hh, lp, 7 + lk)
    for t in (q, k, v):
        t[:, lk:] = 0                                          # pad tokens behind the real ones
    plain = H.attn_fwd_fp8(H.attn_fp8_pack(q[:, :lk].contiguous(), k[:, :lk].contiguous(), v[:, :lk].contiguous()), lk)
    per = [H.attn_fp8_pack(q[:, c * chunk:(c + 1) * chunk].contiguous(), k[:, c * chunk:(c + 1) * chunk].contiguous(),
                           v[:, c * chunk:(c + 1) * chunk].contiguous()) for c in range(n_chunks)]
    kv8 = torch.stack([p[2] for p in per]).contiguous()        # [chunks, B, H, chunk / 64, record]: the all-gather's rank-major result
    assert kv8.shape[3] == chunk // 64
    for c in range(n_chunks):
        if c * chunk >= lk:
            continue                                           # a rank that holds pads only computes rows nobody reads
        got = H.attn_fwd_fp8_chunked(per[c][0], per[c][1], kv8, chunk, lk)
        n = min(chunk, lk - c * chunk)
        want = plain[:, c * chunk:c * chunk + n]
        err = (got[:, :n].float() - want.float()).abs().max().item()
        assert err <= 2.0 ** -7 * want.float().abs().max().item() + 1e-6, (c, err)      # a bf16 ulp of the largest output: other split plans, same records
    if lk == lp and lk % 256 == 0:                             # no pads, whole q blocks: the full query set over chunked records, bit for bit
        qfull = H.attn_fp8_pack(q, k, v)
        assert torch.equal(H.attn_fwd_fp8_chunked(qfull[0], qfull[1], kv8, lk, lk), plain)
    with pytest.raises(RuntimeError):
        H.attn_fwd_fp8_chunked(per[0][0], per[0][1], kv8, chunk, lp + 1)             # more keys than the chunks hold
    with pytest.raises(RuntimeError):
        H.attn_fwd_fp8_chunked(per[0][0], per[0][1], kv8[:, :, :, :, :-1], chunk, lk)  # not record-sized
