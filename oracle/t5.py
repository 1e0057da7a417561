"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the reference's umT5 text encoder:
`WanT5EncoderModel` = FlexAM/models/wan_text_encoder.py:256-305 built from T5LayerNorm :44-56, T5Attention :59-109
(no 1/sqrt(d) scaling, additive relative-position bias + key mask), T5FeedForward :112-130 (gated tanh-GELU),
T5SelfAttention :133-163, T5RelativeEmbedding :208-253.  Pinned by golden G12 (outputs of the reference module
on seeded weights, oracle/make_golden.py) and live in tests/test_oracle_vs_reference.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

T5_TINY = dict(vocab=100, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, shared_pos=False)
UMT5_XXL = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False)


def t5_param_shapes(cfg: dict) -> Dict[str, tuple]:
    d, da, df, n = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"]
    s = {"token_embedding.weight": (cfg["vocab"], d), "norm.weight": (d,)}
    if cfg["shared_pos"]:
        s["pos_embedding.embedding.weight"] = (cfg["num_buckets"], n)
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        s[p + "norm1.weight"], s[p + "norm2.weight"] = (d,), (d,)
        for w in "qkv":
            s[p + f"attn.{w}.weight"] = (da, d)
        s[p + "attn.o.weight"] = (d, da)
        s[p + "ffn.gate.0.weight"], s[p + "ffn.fc1.weight"], s[p + "ffn.fc2.weight"] = (df, d), (df, d), (d, df)
        if not cfg["shared_pos"]:
            s[p + "pos_embedding.embedding.weight"] = (cfg["num_buckets"], n)
    return s


def relative_buckets(lq: int, lk: int, num_buckets: int, max_dist: int = 128) -> Tensor:
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (:219-253) -> int64 [lq, lk]."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def t5_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    return w * (x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps))


def gelu_tanh(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encode(sd: Dict[str, Tensor], cfg: dict, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
    """WanT5EncoderModel.forward (eval: dropout off) -> [B, L, dim]."""
    n = cfg["num_heads"]
    c = cfg["dim_attn"] // n
    x = sd["token_embedding.weight"][input_ids]
    b, l, _ = x.shape
    buckets = relative_buckets(l, l, cfg["num_buckets"])
    shared = sd["pos_embedding.embedding.weight"][buckets].permute(2, 0, 1).unsqueeze(0) if cfg["shared_pos"] else None
    key_mask = attention_mask.view(b, 1, 1, l) == 0
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        e = shared if shared is not None else sd[p + "pos_embedding.embedding.weight"][buckets].permute(2, 0, 1).unsqueeze(0)
        h = t5_norm(x, sd[p + "norm1.weight"])
        q = F.linear(h, sd[p + "attn.q.weight"]).view(b, l, n, c)
        k = F.linear(h, sd[p + "attn.k.weight"]).view(b, l, n, c)
        v = F.linear(h, sd[p + "attn.v.weight"]).view(b, l, n, c)
        bias = (x.new_zeros(b, n, l, l) + e).masked_fill(key_mask, torch.finfo(x.dtype).min)
        attn = F.softmax((torch.einsum("binc,bjnc->bnij", q, k) + bias).float(), dim=-1)
        x = x + F.linear(torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, l, n * c), sd[p + "attn.o.weight"])
        h = t5_norm(x, sd[p + "norm2.weight"])
        x = x + F.linear(F.linear(h, sd[p + "ffn.fc1.weight"]) * gelu_tanh(F.linear(h, sd[p + "ffn.gate.0.weight"])),
                         sd[p + "ffn.fc2.weight"])
    return t5_norm(x, sd["norm.weight"])


def seeded_t5_weights(cfg: dict, seed: int) -> Dict[str, Tensor]:
    """Weights with O(1) activations through the stack (norm weights near 1, projections ~ 1/sqrt(fan_in)).  T5 attention
    has no 1/sqrt(head_dim) factor (the trained q projection carries it, init std (dim*dim_attn)^-0.5,
    wan_text_encoder.py:29), so q is drawn 1/sqrt(head_dim) smaller: logits of order one, not a saturated softmax."""
    g = torch.Generator().manual_seed(seed)
    hd = cfg["dim_attn"] // cfg["num_heads"]
    sd = {}
    for k, shp in t5_param_shapes(cfg).items():
        if k.endswith("norm.weight") or "norm1" in k or "norm2" in k:
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif "pos_embedding" in k:
            sd[k] = torch.randn(shp, generator=g)
        elif k == "token_embedding.weight":
            sd[k] = torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(shp[1])
            if k.endswith("attn.q.weight"):
                sd[k] = sd[k] / math.sqrt(hd)
    return sd


def t5_case(cfg: dict, seed: int = 77, batch: int = 2, length: int = 24, lens=(24, 9)):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg["vocab"], (batch, length), generator=g)
    mask = torch.zeros(batch, length, dtype=torch.long)
    for b in range(batch):
        mask[b, :lens[b % len(lens)]] = 1
        ids[b, lens[b % len(lens)]:] = 0
    return ids, mask
