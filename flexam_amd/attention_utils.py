"""`attention()` operator seam of the reference (FlexAM/models/attention_utils.py:174-233), served by
the gfx950 flash-attention kernel.  Layout [B, L, N, D] like the reference; D must be 128.

The reference dispatches on VIDEOX_ATTENTION_TYPE to flash-attn / SageAttention / torch SDPA; here
every value of that switch maps to the one HIP kernel (there is nothing else to dispatch to), and
unsupported options raise instead of silently changing semantics."""
import warnings

import torch

from . import hip


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None, attention_type=None, attn_mask=None):
    if causal or attn_mask is not None or dropout_p != 0.0 or tuple(window_size) != (-1, -1):
        raise NotImplementedError("flexam_amd.attention: causal / mask / dropout / window are not used on the FlexAM path")
    if q_scale is not None:
        q = q * q_scale
    b, lq, n, d = q.shape
    lk = k.shape[1]
    out_dtype = q.dtype
    q, k, v = (u.to(torch.bfloat16).contiguous() for u in (q, k, v))
    uniform = True
    if k_lens is not None:
        kl = [int(x) for x in k_lens]
        if any(x != kl[0] for x in kl):
            uniform = False
        lk_eff = kl
    if q_lens is not None and any(int(x) != lq for x in q_lens):
        warnings.warn("flexam_amd.attention: q_lens shorter than Lq are computed and left in place (rows past q_lens are not zeroed)")
    if k_lens is None:
        return hip.attn_fwd(q, k, v, softmax_scale=softmax_scale).to(out_dtype)
    if uniform:
        return hip.attn_fwd(q, k[:, :lk_eff[0]], v[:, :lk_eff[0]], softmax_scale=softmax_scale).to(out_dtype)
    out = torch.empty_like(q)
    for i in range(b):                                  # ragged key lengths: one launch per sample
        hip.attn_fwd(q[i:i + 1], k[i:i + 1, :lk_eff[i]], v[i:i + 1, :lk_eff[i]], out=out[i:i + 1], softmax_scale=softmax_scale)
    return out.to(out_dtype)
