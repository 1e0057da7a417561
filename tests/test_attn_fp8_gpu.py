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


@pytest.mark.parametrize("B,Hh,L,splits", [(1, 1, 64, None), (1, 2, 256, None), (1, 1, 300, None), (2, 3, 1111, None), (1, 2, 1024, (2, 0)),
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
