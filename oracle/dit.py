"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the FlexAM DiT forward.

This is the parity oracle for the MI355X HIP path (never shipped, never on the
product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it).  It restates, op for op, what the reference computes in

    /root/reference/FlexAM/models/wan_transformer3d_FlexAM.py   ("FX.py")
    /root/reference/FlexAM/models/attention_utils.py            ("ATT.py")

as plain functions over a state dict that uses the reference's own parameter
names (FX.py:623-705, 381-420, 475-491), so a reference checkpoint or a seeded
random state dict drives the reference module, this oracle and the HIP model
identically.  Parity pinning: tests/test_oracle_golden.py checks every function
here against golden vectors produced by the reference modules themselves
(oracle/make_golden.py, run where /root/reference is mounted).

All arithmetic is fp32 (sinusoid / RoPE angles in fp64 like the reference); on
CPU the reference's CUDA autocast is a no-op, so reference-on-CPU == this file
up to fp32 summation order.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- embeddings
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    """FX.py:31-41 -- [cos | sin] halves, fp64."""
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half)
    ang = torch.outer(pos, inv)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_angles(max_len: int, head_dim: int, theta: float = 10000.0, riflex=None) -> Tensor:
    """Angle table [max_len, head_dim/2] fp64.  FX.py:45-52 builds exp(i*angle) per axis and
    FX.py:655-665 concatenates the three axes with widths d-4*(d//6), 2*(d//6), 2*(d//6)
    (in *real* dims; pairs = half of that: 22/21/21 for d=128).  riflex = (k, L_test, L_test_scale): the temporal
    axis' k-th frequency becomes 0.9 * 2 pi / L_test / L_test_scale (enable_riflex, FX.py:57-113, 774-788)."""
    d = head_dim
    parts = []
    for ax, axis_dim in enumerate((d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6))):
        inv = 1.0 / torch.pow(theta, torch.arange(0, axis_dim, 2, dtype=torch.float64) / axis_dim)
        if ax == 0 and riflex is not None:
            k, l_test, scale = riflex
            inv[k - 1] = 0.9 * 2 * torch.pi / l_test
            if scale is not None:
                inv[k - 1] = inv[k - 1] / scale
        parts.append(torch.outer(torch.arange(max_len, dtype=torch.float64), inv))
    return torch.cat(parts, dim=1)


def rope_apply(x: Tensor, grid: Sequence[int], angles: Tensor) -> Tensor:
    """FX.py:137-164.  x [B, L, N, D]; tokens are (f, h, w) row-major over `grid`; pairs are
    interleaved (x[2i], x[2i+1]); the first c-2*(c//3) pairs rotate by the frame index, the
    next c//3 by h, the last c//3 by w; tokens past f*h*w pass through."""
    b, l, n, d = x.shape
    c = d // 2
    f, h, w = (int(v) for v in grid)
    seq = f * h * w
    cf, ch, cw = c - 2 * (c // 3), c // 3, c // 3
    a_f, a_h, a_w = angles.split([cf, ch, cw], dim=1)
    ang = torch.cat([
        a_f[:f].view(f, 1, 1, cf).expand(f, h, w, cf),
        a_h[:h].view(1, h, 1, ch).expand(f, h, w, ch),
        a_w[:w].view(1, 1, w, cw).expand(f, h, w, cw),
    ], dim=-1).reshape(seq, 1, c)
    cos, sin = ang.cos(), ang.sin()
    xr = x[:, :seq].to(torch.float64).reshape(b, seq, n, c, 2)
    re, im = xr[..., 0], xr[..., 1]
    out = torch.stack([re * cos - im * sin, re * sin + im * cos], dim=-1).reshape(b, seq, n, d)
    out = torch.cat([out, x[:, seq:].to(torch.float64)], dim=1)
    return out.to(x.dtype)


# ----------------------------------------------------------------------------- norms / attention
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """WanRMSNorm, FX.py:173-189: over the whole last dim (3072), not per head."""
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def layer_norm(x: Tensor, eps: float, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None) -> Tensor:
    """WanLayerNorm, FX.py:192-202 (affine only for norm3)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


_BF16_EMULATION = [False]


class bf16_emulation:
    """`with bf16_emulation():` -- the oracle with the ROUNDINGS of the reference's own GPU path put in: under `torch.autocast(bfloat16)`
    every nn.Linear reads bf16 operands and writes a bf16 result (fp32 accumulation), and flash-attention takes bf16 q, k, v, multiplies
    bf16 probabilities with V and returns bf16 (FX.py:242-262, ATT.py:66-122).  Norms, modulation, the residual stream and the softmax
    stay fp32 as in the reference (FX.py:173-202,444-468).  Not a bit-level model of any kernel: it says how far a CORRECT bf16
    implementation lands from the fp32 oracle on a given problem, which is the yardstick for problems where that distance is large
    (peaked softmax rows: a bf16 rounding of q or k moves a score by 2^-9 of its size, and rows with 2-3 effective keys flip)."""

    def __enter__(self):
        self._prev = _BF16_EMULATION[0]
        _BF16_EMULATION[0] = True

    def __exit__(self, *exc):
        _BF16_EMULATION[0] = self._prev
        return False


def _r(t: Optional[Tensor]) -> Optional[Tensor]:
    return t.to(torch.bfloat16).to(torch.float32) if (_BF16_EMULATION[0] and t is not None) else t


def attention(q: Tensor, k: Tensor, v: Tensor, k_lens: Optional[Sequence[int]] = None) -> Tensor:
    """ATT.py:174-233 semantics on the path: plain softmax(q k^T / sqrt(D)) v, non-causal; k_lens (self-attention of a PADDED
    sequence only, FX.py:918-925 -> ATT.py:87-95,115-122): keys past k_lens[b] are masked, as flash-attention's varlen call does.
    (The reference's CPU fallback -- torch SDPA, ATT.py:221-233 -- ignores k_lens with a warning; nothing on the FlexAM path pads, so the
    goldens do not see the difference: that corner is parity-unpinned.)  Layout [B, L, N, D]."""
    if q.shape[2] > 1 and q.shape[0] * q.shape[1] * k.shape[1] * q.shape[2] > (1 << 28):
        # same arithmetic one head at a time: bounds the score matrix (13 GB at L = 11648, 24 heads)
        return torch.cat([attention(q[:, :, i:i + 1], k[:, :, i:i + 1], v[:, :, i:i + 1], k_lens) for i in range(q.shape[2])], dim=2)
    q, k, v = (_r(u).transpose(1, 2) for u in (q, k, v))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if k_lens is not None:
        for b, n in enumerate(k_lens):
            s[b, :, :, int(n):] = float("-inf")
    return _r(torch.matmul(_r(s.softmax(dim=-1)), v)).transpose(1, 2)


def linear(sd: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return _r(F.linear(_r(x), _r(sd[name + ".weight"]), _r(sd.get(name + ".bias"))))


# ----------------------------------------------------------------------------- block
def self_attention(sd, p, x, grid, angles, num_heads, eps, k_lens=None):
    """WanSelfAttention.forward, FX.py:230-262 (k_lens = seq_lens, :251-256)."""
    b, l, d = x.shape
    hd = d // num_heads
    q = rms_norm(linear(sd, p + ".q", x), sd[p + ".norm_q.weight"], eps).view(b, l, num_heads, hd)
    k = rms_norm(linear(sd, p + ".k", x), sd[p + ".norm_k.weight"], eps).view(b, l, num_heads, hd)
    v = linear(sd, p + ".v", x).view(b, l, num_heads, hd)
    q, k = rope_apply(q, grid, angles), rope_apply(k, grid, angles)
    return linear(sd, p + ".o", attention(q, k, v, k_lens).flatten(2))


def cross_attention(sd, p, x, context, num_heads, eps):
    """WanCrossAttention.forward, FX.py:353-371 (text only; padded rows take part)."""
    b, l, d = x.shape
    hd = d // num_heads
    q = rms_norm(linear(sd, p + ".q", x), sd[p + ".norm_q.weight"], eps).view(b, l, num_heads, hd)
    k = rms_norm(linear(sd, p + ".k", context), sd[p + ".norm_k.weight"], eps).view(b, -1, num_heads, hd)
    v = linear(sd, p + ".v", context).view(b, -1, num_heads, hd)
    return linear(sd, p + ".o", attention(q, k, v).flatten(2))


def block_forward(sd, p, x, e0, dens0, grid, angles, context, num_heads, eps=1e-6, seq_lens=None):
    """WanAttentionBlock.forward, FX.py:422-472.
    e0: [B, L, 6, C] (per-token) or [B, 6, C]; dens0: [B, 2, C]."""
    if e0.dim() > 3:
        e = [u.squeeze(2) for u in (sd[p + ".modulation"].unsqueeze(0) + e0).chunk(6, dim=2)]
    else:
        e = (sd[p + ".modulation"] + e0).chunk(6, dim=1)
    dm = (sd[p + ".modulation_density"] + dens0).chunk(2, dim=1)
    h = layer_norm(x, eps) * (1 + e[1]) + e[0] + dm[0]
    x = x + self_attention(sd, p + ".self_attn", h, grid, angles, num_heads, eps, seq_lens) * e[2]
    n3 = layer_norm(x, eps, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    x = x + cross_attention(sd, p + ".cross_attn", n3, context, num_heads, eps)
    h = layer_norm(x, eps) * (1 + e[4]) + e[3] + dm[1]
    y = linear(sd, p + ".ffn.2", F.gelu(linear(sd, p + ".ffn.0", h), approximate="tanh"))
    return x + y * e[5]


def head_forward(sd, x, e, dens, eps=1e-6):
    """Head.forward, FX.py:493-507.  e: [B, L, C] or [B, C]; dens: [B, C]."""
    if e.dim() > 2:
        m = [u.squeeze(2) for u in (sd["head.modulation"].unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)]
    else:
        m = (sd["head.modulation"] + e.unsqueeze(1)).chunk(2, dim=1)
    dm = sd["head.modulation_density"] + dens.unsqueeze(1)
    return linear(sd, "head.head", layer_norm(x, eps) * (1 + m[1]) + m[0] + dm)


# ----------------------------------------------------------------------------- stem pieces
def cnn_block(sd, control_latents: Tensor, additional_control: Tensor) -> Tensor:
    """FX.py:680-705 + 869-880: 5 convs (1,3,3) with GroupNorm+SiLU and two residual adds."""
    def stage(i, x, groups):
        x = F.conv3d(x, sd[f"cnn_conv{i}.0.weight"], sd[f"cnn_conv{i}.0.bias"], padding=(0, 1, 1))
        x = F.group_norm(x, groups, sd[f"cnn_conv{i}.1.weight"], sd[f"cnn_conv{i}.1.bias"], 1e-5)
        return F.silu(x)
    x1 = stage(1, torch.cat([control_latents, additional_control], dim=1), 24)
    x2 = stage(2, x1, 24) + x1
    x3 = stage(3, x2, 12)
    x4 = stage(4, x3, 12) + x3
    return F.conv3d(x4, sd["cnn_conv5.weight"], sd["cnn_conv5.bias"])


def unpatchify(x: Tensor, grid: Sequence[int], patch: Sequence[int], out_dim: int) -> Tensor:
    """FX.py:1126-1149: [L, prod(patch)*c] -> [c, F*pt, H*ph, W*pw]."""
    f, h, w = grid
    u = x[: f * h * w].view(f, h, w, *patch, out_dim)
    u = torch.einsum("fhwpqrc->cfphqwr", u)
    return u.reshape(out_dim, f * patch[0], h * patch[1], w * patch[2])


def time_embed(sd, cfg, t: Tensor):
    """FX.py:928-944.  t [B] -> e [B,C], e0 [B,6,C];  t [B,L] -> e [B,L,C], e0 [B,L,6,C]."""
    dim, fd = cfg["dim"], cfg["freq_dim"]
    if t.dim() == 1:
        e = sinusoidal_embedding_1d(fd, t).float()
    else:
        e = sinusoidal_embedding_1d(fd, t.flatten()).unflatten(0, tuple(t.shape)).float()
    e = linear(sd, "time_embedding.2", F.silu(linear(sd, "time_embedding.0", e)))
    e0 = linear(sd, "time_projection.1", F.silu(e))
    return e, e0.unflatten(-1, (6, dim))


def density_embed(sd, cfg, density: Tensor):
    """FX.py:950-955."""
    d = sinusoidal_embedding_1d(cfg["freq_dim"], density).float()
    d = linear(sd, "density_embedding.2", F.silu(linear(sd, "density_embedding.0", d)))
    return d, linear(sd, "density_projection.1", F.silu(d)).unflatten(1, (2, cfg["dim"]))


def text_embed(sd, cfg, context: List[Tensor]) -> Tensor:
    """FX.py:958-964: zero-pad each prompt to text_len, Linear-GELU(tanh)-Linear."""
    ctx = torch.stack([torch.cat([u, u.new_zeros(cfg["text_len"] - u.size(0), u.size(1))]) for u in context])
    return linear(sd, "text_embedding.2", F.gelu(linear(sd, "text_embedding.0", ctx), approximate="tanh"))


# ----------------------------------------------------------------------------- TeaCache (FX.py:977-1051, cache_utils.py:21-76)
def teacache_state(coefficients, num_steps: int, rel_l1_thresh: float, num_skip_start_steps: int = 0) -> dict:
    import numpy as np
    return dict(rescale=np.poly1d(coefficients), num_steps=num_steps, thresh=rel_l1_thresh, skip_start=num_skip_start_steps,
                cnt=0, acc=0.0, prev_mod=None, residual=None, should_calc=True)


def _teacache_decide(tc: dict, e0: Tensor) -> bool:
    mod_inp = e0[:, -1, :] if e0.dim() > 3 else e0                              # FX.py:980-983
    if tc["cnt"] < tc["skip_start"]:
        calc, tc["acc"] = True, 0.0
    else:
        rel = ((mod_inp - tc["prev_mod"]).abs().mean() / tc["prev_mod"].abs().mean()).item()
        tc["acc"] += float(tc["rescale"](rel))
        calc = not (tc["acc"] < tc["thresh"])
        if calc:
            tc["acc"] = 0.0
    tc["prev_mod"], tc["should_calc"] = mod_inp, calc
    return calc


# ----------------------------------------------------------------------------- full forward
def dit_forward(sd: Dict[str, Tensor], cfg: dict, x: Tensor, t: Tensor, context: List[Tensor], seq_len: int,
                y: Optional[Tensor] = None, full_ref: Optional[Tensor] = None,
                additional_control: Optional[Tensor] = None, density: Optional[Tensor] = None,
                taps: Optional[dict] = None, teacache: Optional[dict] = None, riflex=None) -> Tensor:
    """WanTransformer3DModel_FlexAM.forward, FX.py:817-1123 (inference, sp=1, no TeaCache, no
    camera adapter, no subject_ref, clip_fea=None -- the FlexAM 5B call contract).

    x [B,C,F,H,W]; t [B] or [B,Lvid]; context list of [len_i, text_dim]; y [B,100,F,H,W];
    full_ref [B,C,H,W]; additional_control [B,240,F,H,W]; density [B] -> [B,out_dim,F,H,W].
    `taps` (optional dict) receives intermediate tensors for localising mismatches."""
    dim, nh, nl, patch = cfg["dim"], cfg["num_heads"], cfg["num_layers"], tuple(cfg["patch_size"])
    eps = cfg.get("eps", 1e-6)
    b = x.shape[0]
    if y is not None:
        if "cnn_conv1.0.weight" in sd and additional_control is not None:          # FX.py:869-881
            c = x.shape[1]
            y = torch.cat([cnn_block(sd, y[:, :c], additional_control), y[:, c:]], dim=1)
        x = torch.cat([x, y], dim=1)                                              # FX.py:883
    x = F.conv3d(x, sd["patch_embedding.weight"], sd["patch_embedding.bias"], stride=patch)   # FX.py:885
    grid = list(x.shape[2:])
    x = x.flatten(2).transpose(1, 2)                                              # [B, Lvid, C]
    if "ref_conv.weight" in sd and full_ref is not None:                          # FX.py:895-904
        r = F.conv2d(full_ref, sd["ref_conv.weight"], sd["ref_conv.bias"], stride=patch[1:]).flatten(2).transpose(1, 2)
        grid[0] += 1
        seq_len += r.size(1)
        x = torch.cat([r, x], dim=1)
        if t.dim() != 1 and t.size(1) < seq_len:
            t = torch.cat([t[:, -1:].repeat(1, seq_len - t.size(1)), t], dim=1)
    assert x.size(1) <= seq_len
    seq_lens = [x.size(1)] * b if x.size(1) < seq_len else None                   # FX.py:918: only a padded sequence needs the key mask
    x = torch.cat([x, x.new_zeros(b, seq_len - x.size(1), dim)], dim=1)           # FX.py:918-925
    if t.dim() != 1 and t.size(1) < seq_len:                                      # FX.py:930-934
        t = torch.cat([t, t[:, -1:].repeat(1, seq_len - t.size(1))], dim=1)
    e, e0 = time_embed(sd, cfg, t)
    dens, dens0 = density_embed(sd, cfg, density)
    ctx = text_embed(sd, cfg, context)
    angles = rope_angles(1024, dim // nh, riflex=riflex)
    if taps is not None:
        taps.update(x_embed=x.clone(), e=e, e0=e0, dens0=dens0, context=ctx)
    calc = True if teacache is None else _teacache_decide(teacache, e0)
    if not calc:
        x = x + teacache["residual"][-x.size(0):]                                 # FX.py:1003-1006 (a B = 1 cfg-skipped call takes the cond row)
    else:
        x_in = x
        for i in range(nl):
            x = block_forward(sd, f"blocks.{i}", x, e0, dens0, grid, angles, ctx, nh, eps, seq_lens)
            if taps is not None:
                taps[f"block{i}"] = x.clone()
        if teacache is not None:
            teacache["residual"] = x - x_in                                       # FX.py:1048-1051
    if teacache is not None:                                                      # FX.py:1119-1122
        teacache["cnt"] += 1
        if teacache["cnt"] == teacache["num_steps"]:
            teacache.update(cnt=0, acc=0.0, prev_mod=None, residual=None, should_calc=True)
    x = head_forward(sd, x, e, dens, eps)                                         # FX.py:1101
    if "ref_conv.weight" in sd and full_ref is not None:                          # FX.py:1106-1109
        x = x[:, r.size(1):]
        grid[0] -= 1
    return torch.stack([unpatchify(u, grid, patch, cfg["out_dim"]) for u in x])


# ----------------------------------------------------------------------------- parameter inventory
def dit_param_shapes(cfg: dict) -> Dict[str, tuple]:
    """State-dict inventory of Wan2_2Transformer3DModel_FlexAM for a config (names as in the
    reference: FX.py:623-705, 401-420, 484-491).  make_golden.py asserts this equals the
    reference module's own state_dict() keys and shapes."""
    d, f, nl = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    pt, ph, pw = cfg["patch_size"]
    s = {"patch_embedding.weight": (d, cfg["in_dim"], pt, ph, pw), "patch_embedding.bias": (d,)}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)
    lin("text_embedding.0", d, cfg["text_dim"]); lin("text_embedding.2", d, d)
    lin("time_embedding.0", d, cfg["freq_dim"]); lin("time_embedding.2", d, d)
    lin("time_projection.1", 6 * d, d)
    lin("density_embedding.0", d, cfg["freq_dim"]); lin("density_embedding.2", d, d)
    lin("density_projection.1", 2 * d, d)
    for i in range(nl):
        p = f"blocks.{i}"
        s[p + ".modulation"] = (1, 6, d)
        s[p + ".modulation_density"] = (1, 2, d)
        for a in ("self_attn", "cross_attn"):
            for w in "qkvo":
                lin(f"{p}.{a}.{w}", d, d)
            s[f"{p}.{a}.norm_q.weight"] = (d,)
            s[f"{p}.{a}.norm_k.weight"] = (d,)
        s[p + ".norm3.weight"] = (d,)
        s[p + ".norm3.bias"] = (d,)
        lin(p + ".ffn.0", f, d); lin(p + ".ffn.2", d, f)
    s["head.modulation"] = (1, 2, d)
    s["head.modulation_density"] = (1, 1, d)
    lin("head.head", pt * ph * pw * cfg["out_dim"], d)
    if cfg.get("add_ref_conv"):
        s["ref_conv.weight"] = (d, cfg["in_dim_ref_conv"], ph, pw)
        s["ref_conv.bias"] = (d,)
    if cfg.get("add_cnn_block"):
        chans = [(cfg["in_dim_cnn_block"], 192), (192, 192), (192, 96), (96, 96)]
        for i, (ci, co) in enumerate(chans, 1):
            s[f"cnn_conv{i}.0.weight"] = (co, ci, 1, 3, 3)
            s[f"cnn_conv{i}.0.bias"] = (co,)
            s[f"cnn_conv{i}.1.weight"] = (co,)
            s[f"cnn_conv{i}.1.bias"] = (co,)
        s["cnn_conv5.weight"] = (cfg["out_dim_cnn_block"], 96, 1, 1, 1)
        s["cnn_conv5.bias"] = (cfg["out_dim_cnn_block"],)
    return s


def _seeded_tensor(name: str, shape: tuple, r: Tensor) -> Tensor:
    """Scale rule of the seeded state dicts: activations stay O(1); nothing is left at the reference's zero-init
    (FX.py:1172-1188, VAE.py:258) so every path contributes to the output."""
    if name.endswith("gamma"):
        return 1.0 + 0.1 * r
    if name.endswith(".bias"):
        return 0.05 * r
    if "modulation" in name:
        return r / math.sqrt(shape[-1])
    if len(shape) == 1:                            # norm weights
        return 1.0 + 0.1 * r
    fan_in = 1                                     # linear / conv weights: ~1/sqrt(fan_in)
    for v in shape[1:]:
        fan_in *= v
    return r / math.sqrt(fan_in)


def seeded_state_dict(shapes: Dict[str, tuple], seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic random weights (CPU generator; same torch build here and on the GPU box): ONE generator walked over the
    sorted names (the goldens' checksums depend on exactly this order)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name in sorted(shapes):
        shape = shapes[name]
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        sd[name] = _seeded_tensor(name, shape, r).to(dtype)
    return sd


def seeded_state_dict_threaded(shapes: Dict[str, tuple], seed: int, threads: int = 16) -> Dict[str, Tensor]:
    """The same scale rule with ONE generator PER TENSOR (seeded by (seed, position in the sorted names)) so the tensors can be drawn
    on a thread pool: the 5 B parameters of the full-depth model in seconds instead of half a minute.  Not bit-equal to
    seeded_state_dict() -- a different, equally deterministic draw; no golden depends on it."""
    from concurrent.futures import ThreadPoolExecutor
    names = sorted(shapes)

    def draw(i):
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        r = torch.randn(shapes[names[i]], generator=g, dtype=torch.float32)
        return _seeded_tensor(names[i], shapes[names[i]], r)
    with ThreadPoolExecutor(threads) as ex:
        vals = list(ex.map(draw, range(len(names))))
    return dict(zip(names, vals))


DIT_5B = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=148, dim=3072, ffn_dim=14336,
              freq_dim=256, text_dim=4096, out_dim=48, num_heads=24, num_layers=30, eps=1e-6,
              add_ref_conv=True, in_dim_ref_conv=48, add_cnn_block=True, in_dim_cnn_block=288,
              out_dim_cnn_block=48)

DIT_TINY = dict(DIT_5B, dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=16)
