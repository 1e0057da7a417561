"""`attention()` operator seam of the reference (FlexAM/models/attention_utils.py:174-233), served by
the gfx950 flash-attention kernel.  Layout [B, L, N, D] like the reference; D must be 128.

The reference dispatches on VIDEOX_ATTENTION_TYPE to flash-attn / SageAttention / torch SDPA (attention_utils.py:195-233).
Here FLASH_ATTENTION and every other value map to the bf16 HIP kernel; SAGE_ATTENTION -- the reference's quantised attention --
maps to the MXFP8 kernel (csrc/attn_fp8.inc: e4m3 Q, K, P and V with E8M0 block scales, fp32 softmax) for self-attention shapes
(Lq == Lk, no key lengths) and to the bf16 kernel otherwise.  Unsupported options raise instead of silently changing semantics."""
import os
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
    if attention_type is None:
        attention_type = os.environ.get("VIDEOX_ATTENTION_TYPE", "FLASH_ATTENTION")
    if attention_type == "SAGE_ATTENTION" and torch.is_grad_enabled():
        attention_type = "FLASH_ATTENTION"              # as the reference does (attention_utils.py:196-197)
    if attention_type == "SAGE_ATTENTION" and k_lens is None and lq == lk:
        scale = (softmax_scale if softmax_scale is not None else d ** -0.5) * 1.4426950408889634
        bufs = hip.attn_fp8_pack((q.float() * scale).to(torch.bfloat16), k, v)
        return hip.attn_fwd_fp8(bufs, lq).to(out_dtype)
    if k_lens is None:
        return hip.attn_fwd(q, k, v, softmax_scale=softmax_scale).to(out_dtype)
    if uniform:
        return hip.attn_fwd(q, k[:, :lk_eff[0]], v[:, :lk_eff[0]], softmax_scale=softmax_scale).to(out_dtype)
    out = torch.empty_like(q)
    for i in range(b):                                  # ragged key lengths: one launch per sample
        hip.attn_fwd(q[i:i + 1], k[i:i + 1, :lk_eff[i]], v[i:i + 1, :lk_eff[i]], out=out[i:i + 1], softmax_scale=softmax_scale)
    return out.to(out_dtype)
