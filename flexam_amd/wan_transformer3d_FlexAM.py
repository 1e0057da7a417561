"""MI355X drop-in for FlexAM's DiT: `Wan2_2Transformer3DModel_FlexAM` / `WanTransformer3DModel_FlexAM`.

Same constructor, config attributes, state-dict key names, `forward` signature and feature
switches as the reference classes in FlexAM/models/wan_transformer3d_FlexAM.py (:526-1438), so
`pipelines.py:1120` / `nodes.py` can load a checkpoint into it and the sampler can call it
unchanged.  The modules below only *hold* parameters under the reference's names (nn.Linear /
nn.Conv3d are used as containers, never called); all arithmetic of `forward` runs in
libflexam_hip.so through `DiTEngine` (flexam_amd/dit_engine.py).  On a host without the HIP library
or without a GPU `forward` raises -- there is no eager fallback.
"""
import glob
import json
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .cache_utils import TeaCache
from .cfg_optimization import cfg_skip
from .dit_engine import DiTEngine
from .rope import rope_angle_table

F32 = torch.float32


class ModelConfig(dict):
    """`.config.patch_size` and `.config.get("add_ref_conv")` like a diffusers FrozenDict."""
    __getattr__ = dict.get


BF16 = torch.bfloat16
_QSCALE_LOG2E = 1.4426950408889634


def _on(t: torch.Tensor, dev, dtype) -> torch.Tensor:
    """`t` as a contiguous `dtype` tensor on `dev`; shares storage with `t` when it already is one (in-place edits of the
    parameter -- LoRA merges, `.data +=` -- are then seen by the kernels without a re-pack)."""
    d = t.detach()
    if d.device == dev and d.dtype == dtype and d.is_contiguous():
        return d
    return d.to(dev, dtype).contiguous()


def _fuse_rows(params, dev) -> torch.Tensor:
    """Row-wise concatenation of several weight matrices into ONE bf16 buffer (q|k|v, cross k|v) so that a single GEMM
    produces all of them.  When the parameters already are bf16 on `dev`, they are re-pointed at their row range of the
    fused buffer: state-dict keys, shapes and values are unchanged, in-place edits write through, and no second copy of
    the weights exists."""
    fused = torch.cat([p.detach().to(dev, BF16) for p in params])
    if all(p.dtype == BF16 and p.device == dev for p in params):
        off = 0
        for p in params:
            p.data = fused[off:off + p.shape[0]]
            off += p.shape[0]
    return fused


def _param_sig(module: nn.Module):
    """What a cached pack of `module`'s parameters depends on.  Storage-sharing packs follow in-place edits by themselves;
    copies (fp32 checkpoints, fp32 bias / norm rows) are caught through the version counter, which every in-place
    torch op on the parameter bumps (`p.data` edits do not: call `invalidate_engine()` after those)."""
    return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in module.parameters(recurse=False))


class HipLinear(nn.Linear):
    """nn.Linear as a parameter holder under the reference's key names whose `forward` is one `flexam_gemm_bf16` launch
    (bf16 operands, fp32 accumulate and bias, bf16 result: what autocast makes of nn.Linear in the reference)."""

    _pk = None

    def packed(self):
        dev = self.weight.device
        sig = _param_sig(self)
        if self._pk is None or self._pk[0] != sig:
            w = self.weight.detach()
            k = w.shape[1]
            if k % 64:                                        # the GEMM wants K in 64-element blocks: zero-pad once
                wp = torch.zeros(w.shape[0], (k + 63) // 64 * 64, device=dev, dtype=BF16)
                wp[:, :k] = w.to(BF16)
            else:
                wp = _on(w, dev, BF16)
            self._pk = (sig, wp, _on(self.bias, dev, F32) if self.bias is not None else None)
        return self._pk[1], self._pk[2]

    def forward(self, x, epilogue: int = 0):
        from . import hip
        w, b = self.packed()
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1]).to(BF16)
        if h.shape[1] != w.shape[1]:
            hp = torch.zeros(h.shape[0], w.shape[1], device=h.device, dtype=BF16)
            hp[:, :h.shape[1]] = h
            h = hp
        return hip.gemm(h.contiguous(), w, b, epilogue=epilogue).view(*lead, w.shape[0])


def _holder(n_out: int, n_in: int) -> nn.Linear:
    return HipLinear(n_in, n_out)


def _fp8_weight(pk: dict, name: str):
    """(e4m3 bytes, per-output-channel scales) of pack entry `name`, quantised once per pack (the pack is rebuilt when a parameter changes)."""
    from . import hip
    key = "_f8_" + name
    if key not in pk:
        pk[key] = hip.quantize_rows_fp8(pk[name])
    return pk[key]


def _proj_fp8(h: torch.Tensor, pk: dict, wname: str, bias, epilogue: int = 0) -> torch.Tensor:
    """h [M, K] bf16 -> bf16 [M, N] on the fp8 (OCP e4m3) MFMA path: rows of h quantised per call (absmax / 448), weights per output channel,
    fp32 accumulation, scales / bias / activation in the epilogue (csrc/gemm_fp8.hip) -- the module-seam form of DiTEngine.enable_fp8."""
    from . import hip
    a8, sa = hip.quantize_rows_fp8(h)
    w8, sw = _fp8_weight(pk, wname)
    return hip.gemm_fp8(a8, sa, w8, sw, bias, epilogue=epilogue)


class _Norm(nn.Module):
    """WanRMSNorm (wan_transformer3d_FlexAM.py:173-189; weight only) or the affine WanLayerNorm `norm3` (:192-202)."""

    def __init__(self, dim: int, bias: bool = False, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        from . import hip
        dev = x.device
        lead, c = x.shape[:-1], x.shape[-1]
        if hasattr(self, "bias"):                              # LayerNorm with affine, fp32 statistics, bf16 result
            return hip.ln_modulate(x.reshape(-1, c).to(F32).contiguous(), eps=self.eps, ln_w=_on(self.weight, dev, F32),
                                   ln_b=_on(self.bias, dev, F32)).view(*lead, c)
        h = x.reshape(-1, c).to(BF16).contiguous().clone()
        hip.rmsnorm_rope(h, _on(self.weight, dev, F32), eps=self.eps)
        return h.view(*lead, c)


class _Attn(nn.Module):
    """q/k/v/o projections + full-width RMSNorm weights (keys: q.weight ... norm_k.weight)."""

    def __init__(self, dim: int, num_heads: int = None, eps: float = 1e-6):
        super().__init__()
        self.dim, self.num_heads, self.eps = dim, num_heads, eps
        self.head_dim = dim // num_heads if num_heads else None
        self.q, self.k, self.v, self.o = (_holder(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = _Norm(dim, eps=eps), _Norm(dim, eps=eps)
        self._pk = None
        self._fp8 = False                                      # q|k|v (self-attention) / q (cross-attention) projection on the fp8 MFMA path (model.enable_fp8_gemm)

    def _sig(self):
        return tuple(_param_sig(m) for m in (self.q, self.k, self.v, self.o, self.norm_q, self.norm_k))

    def _q_scale(self) -> float:
        # softmax_scale * log2(e) rides on the RMSNorm weight of q: q leaves flexam_rmsnorm_rope in exp2 units with the one
        # rounding to bf16 it always had, and the attention kernel's FLEXAM_ATTN_PRESCALED form needs no multiply per score
        return (self.head_dim ** -0.5) * _QSCALE_LOG2E

    @staticmethod
    def _heads(t2d, b, l, nh, hd):
        return t2d.view(b, l, -1)[:, :, :nh * hd].unflatten(2, (nh, hd)) if t2d.shape[1] == nh * hd else None


class _SelfAttn(_Attn):
    """WanSelfAttention (wan_transformer3d_FlexAM.py:205-262) on the HIP kernels: fused q|k|v GEMM, full-width RMSNorm +
    3-axis RoPE in one pass, flash attention, output projection."""

    def packed(self):
        sig = self._sig()
        if self._pk is None or self._pk["sig"] != sig:
            dev = self.q.weight.device
            f32 = lambda t: _on(t, dev, F32)
            wqkv = _fuse_rows([self.q.weight, self.k.weight, self.v.weight], dev)
            self._pk = dict(wqkv=wqkv, bqkv=torch.cat([f32(self.q.bias), f32(self.k.bias), f32(self.v.bias)]),
                            wo=_on(self.o.weight, dev, BF16), bo=f32(self.o.bias),
                            nq=f32(self.norm_q.weight) * self._q_scale(), nk=f32(self.norm_k.weight))
            self._pk["sig"] = self._sig()                     # after the re-pointing of q/k/v
        return self._pk

    _rope_cache = None

    def _rope(self, grid_sizes, seq_len: int, freqs: torch.Tensor, dev):
        from .rope import rope_tables
        g = grid_sizes.tolist() if torch.is_tensor(grid_sizes) else list(grid_sizes)
        g = g if isinstance(g[0], (list, tuple)) else [g]
        if any(tuple(u) != tuple(g[0]) for u in g):
            raise NotImplementedError("flexam_amd: all samples of a batch share one token grid on the FlexAM path")
        key = (tuple(int(v) for v in g[0]), seq_len, freqs.data_ptr(), freqs._version, str(dev))
        if self._rope_cache is None or self._rope_cache[0] != key:
            ang = freqs.detach().cpu()
            ang = ang.angle().to(torch.float64) if ang.is_complex() else ang.to(torch.float64)   # the reference stores exp(i angle)
            cos, sin = rope_tables(key[0], seq_len, self.head_dim, ang)
            self._rope_cache = (key, cos.to(dev), sin.to(dev))
        return self._rope_cache[1], self._rope_cache[2]

    def forward(self, x, seq_lens, grid_sizes, freqs, dtype=torch.bfloat16, t=0):
        """x [B, L, C] -> [B, L, C] bf16; seq_lens [B] (all equal to L on this path), grid_sizes [B, 3], freqs [1024, C/heads/2]
        (angle table, or the reference's complex exp(i angle) table)."""
        from . import hip
        from .dist import current_sp_context
        b, l, c = x.shape
        nh, hd = self.num_heads, self.head_dim
        sp = current_sp_context()
        if sp is not None and sp["size"] > 1:
            return self._forward_chunk(x, seq_lens, sp)
        # A caller that hands over a PADDED sequence (FX.py:918-925: zero tokens behind seq_lens[b] real ones, the same count for every
        # sample) gets flash-attention's semantics: the pads are no keys (k_lens, FX.py:251-256 -> ATT.py:87-95); their own rows are
        # computed like any other and are the caller's to drop.  The model's own forward never pads (it does not create the tokens).
        n_keys = l
        if seq_lens is not None:
            lens = [int(v) for v in (seq_lens.tolist() if torch.is_tensor(seq_lens) else seq_lens)]
            if any(v != lens[0] for v in lens) or lens[0] > l or lens[0] < 1:
                raise NotImplementedError("flexam_amd: ragged seq_lens inside one batch do not occur on the FlexAM path (one clip per call)")
            n_keys = lens[0]
        pk = self.packed()
        h = x.reshape(b * l, c).to(BF16).contiguous()
        qkv = _proj_fp8(h, pk, "wqkv", pk["bqkv"]) if self._fp8 else hip.gemm(h, pk["wqkv"], pk["bqkv"])
        cos, sin = self._rope(grid_sizes, l, freqs, x.device)
        hip.rmsnorm_rope(qkv[:, 0:c], pk["nq"], qkv[:, c:2 * c], pk["nk"], eps=self.eps, rope_cos=cos, rope_sin=sin,
                         tokens_per_batch=l, token_offset=0, head_dim=hd)
        q3 = qkv.view(b, l, 3 * c)
        q4, k4, v4 = (q3[:, :, i * c:(i + 1) * c].unflatten(2, (nh, hd)) for i in range(3))
        if n_keys < l:
            ao = hip.attn_fwd(q4, k4[:, :n_keys], v4[:, :n_keys], prescaled=True)
        elif os.environ.get("VIDEOX_ATTENTION_TYPE", "FLASH_ATTENTION") == "SAGE_ATTENTION" and not torch.is_grad_enabled() and hd == 128:
            ao = hip.attn_fwd_fp8(hip.attn_fp8_pack(q4, k4, v4), l)      # the reference's attention() reads the switch per call (ATT.py:195-203)
        else:
            ao = hip.attn_fwd(q4, k4, v4, prescaled=True)
        return hip.gemm(ao.view(b * l, c), pk["wo"], pk["bo"]).view(b, l, c)


    def _forward_chunk(self, x, seq_lens, sp):
        """Sequence-parallel form of forward() for blocks that are called as modules (the reference's `usp_attn_forward`, re-bound at
        wan_transformer3d_FlexAM.py:807-815): x is this rank's token chunk [B, L/N, C]; q|k|v of the chunk, RMSNorm + RoPE at the
        chunk's GLOBAL token offset, ONE all-gather of the normed / rotated K|V over the sequence-parallel group (RCCL over xGMI; the
        rank-major concatenation is the token order), attention of the local queries to all keys, output projection.  The engine's
        fused path does the same exchange in head-group pieces overlapped with compute (DiTEngine._allgather_attention); this seam
        form favours being one plain forward a wrapper can trace."""
        from . import hip
        from .dist import all_gather_seq
        b, lc, c = x.shape
        nh, hd = self.num_heads, self.head_dim
        L = sp["seq_len"]                                                           # the REAL length; lc * size is the padded one
        Lp = lc * sp["size"]
        if not (L <= Lp < L + sp["size"]) or (seq_lens is not None and any(int(v) != L for v in (seq_lens.tolist() if torch.is_tensor(seq_lens) else seq_lens))):
            raise NotImplementedError(f"flexam_amd: a sequence-parallel chunk of {lc} tokens x {sp['size']} ranks is not the padded form of seq_lens / L = {L} "
                                      "(ceil(L / ranks) * ranks rows, FX.py:919-920)")
        pk = self.packed()
        h = x.reshape(b * lc, c).to(BF16).contiguous()
        qkv = _proj_fp8(h, pk, "wqkv", pk["bqkv"]) if self._fp8 else hip.gemm(h, pk["wqkv"], pk["bqkv"])
        hip.rmsnorm_rope(qkv[:, 0:c], pk["nq"], qkv[:, c:2 * c], pk["nk"], eps=self.eps, rope_cos=sp["rope_cos"], rope_sin=sp["rope_sin"],
                         tokens_per_batch=lc, token_offset=sp["token_offset"], head_dim=hd)
        q3 = qkv.view(b, lc, 3 * c)
        kv = all_gather_seq(q3[:, :, c:].contiguous(), sp["group"])                  # [B, Lp, 2C]; rows L .. Lp - 1 are the zero pad tokens:
        ao = hip.attn_fwd(q3[:, :, 0:c].unflatten(2, (nh, hd)), kv[:, :L, 0:c].unflatten(2, (nh, hd)), kv[:, :L, c:].unflatten(2, (nh, hd)),   # no keys (k_lens, FX.py:251-256)
                          prescaled=True)
        return hip.gemm(ao.view(b * lc, c), pk["wo"], pk["bo"]).view(b, lc, c)


class _CrossAttn(_Attn):
    """WanCrossAttention (wan_transformer3d_FlexAM.py:353-371): text-only K/V, no RoPE, padded text rows take part."""

    def packed(self):
        sig = self._sig()
        if self._pk is None or self._pk["sig"] != sig:
            dev = self.q.weight.device
            f32 = lambda t: _on(t, dev, F32)
            cwkv = _fuse_rows([self.k.weight, self.v.weight], dev)
            self._pk = dict(cwq=_on(self.q.weight, dev, BF16), cbq=f32(self.q.bias), cwkv=cwkv,
                            cbkv=torch.cat([f32(self.k.bias), f32(self.v.bias)]), cwo=_on(self.o.weight, dev, BF16),
                            cbo=f32(self.o.bias), cnq=f32(self.norm_q.weight) * self._q_scale(), cnk=f32(self.norm_k.weight))
            self._pk["sig"] = self._sig()
        return self._pk

    def context_kv(self, context2d: torch.Tensor) -> torch.Tensor:
        """context [B*T, C] bf16 -> [B*T, 2C]: normalised K | V (step-invariant: the engine calls this once per clip)."""
        from . import hip
        pk = self.packed()
        kv = hip.gemm(context2d, pk["cwkv"], pk["cbkv"])
        hip.rmsnorm_rope(kv[:, :self.dim], pk["cnk"], eps=self.eps)
        return kv

    def forward(self, x, context, context_lens, dtype=torch.bfloat16, t=0):
        """x [B, L, C], context [B, T, C] -> [B, L, C] bf16."""
        from . import hip
        b, l, c = x.shape
        nh, hd = self.num_heads, self.head_dim
        tl = context.shape[1]
        pk = self.packed()
        h = x.reshape(b * l, c).to(BF16).contiguous()
        q = _proj_fp8(h, pk, "cwq", pk["cbq"]) if self._fp8 else hip.gemm(h, pk["cwq"], pk["cbq"])
        hip.rmsnorm_rope(q, pk["cnq"], eps=self.eps)
        kv = self.context_kv(context.reshape(b * tl, c).to(BF16).contiguous()).view(b, tl, 2 * c)
        q4 = q.view(b, l, nh, hd)
        k4, v4 = kv[:, :, 0:c].unflatten(2, (nh, hd)), kv[:, :, c:].unflatten(2, (nh, hd))
        if context_lens is None:
            ao = hip.attn_fwd(q4, k4, v4, prescaled=True)
        else:                                                  # ragged text lengths: one launch per sample (ATT.py:87-95)
            ao = torch.empty(b, l, nh, hd, device=x.device, dtype=BF16)
            for i, n in enumerate(int(v) for v in context_lens):
                hip.attn_fwd(q4[i:i + 1], k4[i:i + 1, :n], v4[i:i + 1, :n], out=ao[i:i + 1], prescaled=True)
        return hip.gemm(ao.view(b * l, c), pk["cwo"], pk["cbo"]).view(b, l, c)


def adaln_rows(e: torch.Tensor, batch: int, seq_len: int):
    """AdaLN input `e` of a block / the head -> (rows [R, nj, C] fp32, int32 row index [B*L] or None, table rows per batch).
    The model's own forward hands the compact form over as an attribute of the materialised tensor (two distinct rows per
    sample in every demo mode); a foreign per-token tensor [B, L, nj, C] is used row by row."""
    meta = getattr(e, "_flexam_rows", None)
    if meta is not None:
        return meta
    if e.dim() == 3:                                           # [B, nj, C]: one row per sample
        return e.to(F32).contiguous(), None, 1
    b, l = e.shape[:2]
    if e.stride(1) == 0:                                       # an expanded per-sample row
        return e[:, 0].to(F32).contiguous(), None, 1
    rows = e.reshape(b * l, *e.shape[2:]).to(F32).contiguous()
    return rows, torch.arange(b * l, device=e.device, dtype=torch.int32), l


class _Block(nn.Module):
    """WanAttentionBlock (wan_transformer3d_FlexAM.py:381-472) on the HIP kernels.  Inside the model's own forward the engine
    runs the block with fused GEMM epilogues (flexam_amd/dit_engine.py); this `forward` is the reference's block-level seam
    (`transformer.blocks[i] = wrapper(block)`, comfyui/comfyui_nodes.py:67-71; `block.self_attn.forward = MethodType(...)`,
    wan_transformer3d_FlexAM.py:807-815): same signature, it calls `self.self_attn` / `self.cross_attn` as modules so a
    re-bound attention forward takes effect, and the engine routes a step through it whenever a block is not pristine."""

    def __init__(self, dim: int, ffn_dim: int, num_heads: int = None, eps: float = 1e-6):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.eps = dim, ffn_dim, num_heads, eps
        self.self_attn, self.cross_attn = _SelfAttn(dim, num_heads, eps), _CrossAttn(dim, num_heads, eps)
        self.norm3 = _Norm(dim, bias=True, eps=eps)
        self.ffn = nn.Sequential(_holder(ffn_dim, dim), nn.GELU(approximate="tanh"), _holder(dim, ffn_dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self.modulation_density = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)
        self._pk = None
        self._fp8 = False                                      # FFN on the fp8 MFMA path when the block is called as a module (set_fp8)

    def set_fp8(self, on: bool):
        """BASELINE configs[4] through the block-level seam: the same GEMMs DiTEngine.enable_fp8 moves to the fp8 pipe -- self-attention
        q|k|v, cross-attention q, FFN1 (+ GELU), FFN2 (+ gated residual) -- when this block runs as a module (replaced / wrapped /
        re-bound blocks: comfyui/comfyui_nodes.py:67-71, wan_transformer3d_FlexAM.py:807-815).  Activations: absmax rows per call."""
        self._fp8 = self.self_attn._fp8 = self.cross_attn._fp8 = bool(on)

    def pristine(self) -> bool:
        """No instance-level `forward` on the block or its attention modules (what types.MethodType re-binding creates)."""
        return not any("forward" in m.__dict__ for m in (self, self.self_attn, self.cross_attn))

    def packed(self):
        """Device-side parameter pack of this block (the engine's per-layer dict)."""
        sig = (_param_sig(self), _param_sig(self.norm3), _param_sig(self.ffn[0]), _param_sig(self.ffn[2]))
        sa, ca = self.self_attn.packed(), self.cross_attn.packed()
        if self._pk is None or self._pk["sig"] != sig or self._pk["sa"] is not sa or self._pk["ca"] is not ca:
            dev = self.modulation.device
            f32 = lambda t: _on(t, dev, F32)
            pk = dict(sa)
            pk.update(ca)
            pk.update(n3w=f32(self.norm3.weight), n3b=f32(self.norm3.bias), w1=_on(self.ffn[0].weight, dev, BF16),
                      b1=f32(self.ffn[0].bias), w2=_on(self.ffn[2].weight, dev, BF16), b2=f32(self.ffn[2].bias),
                      mod=f32(self.modulation)[0], mdens=f32(self.modulation_density)[0], sig=sig, sa=sa, ca=ca)
            self._pk = pk
        return self._pk

    @torch.no_grad()
    def forward(self, x, e, density_emb, seq_lens, grid_sizes, freqs, context, context_lens, dtype=torch.bfloat16, t=0):
        """x [B, L, C]; e [B, 6, C] or per-token [B, L, 6, C] (fp32); density_emb [B, 2, C]; context [B, T, C]
        -> [B, L, C] fp32 (the residual stream is fp32 from the first gated add on, wan_transformer3d_FlexAM.py:456)."""
        from . import hip
        b, l, c = x.shape
        pk = self.packed()
        xres = x.reshape(b * l, c).to(F32).clone()
        rows, idx, rpb = adaln_rows(e, b, l)
        tab = torch.empty(1, rows.shape[0], 6, c, device=x.device, dtype=F32)
        hip.mod_table(pk["mod"].unsqueeze(0), rows, tab, rpb, 0b010010, pk["mdens"].unsqueeze(0),
                      density_emb.to(F32).contiguous(), 0xFF1FF0)
        T = tab[0]
        hbuf = hip.ln_modulate(xres, eps=self.eps, shift=T[:, 0], scale=T[:, 1], row_index=idx, rows_per_batch=l)
        y = self.self_attn(hbuf.view(b, l, c), seq_lens, grid_sizes, freqs, dtype, t=t)
        hip.gate_residual(xres, y.reshape(b * l, c).to(BF16).contiguous(), gate=T[:, 2], row_index=idx, rows_per_batch=l)
        hip.ln_modulate(xres, out=hbuf, eps=self.eps, ln_w=pk["n3w"], ln_b=pk["n3b"])
        y = self.cross_attn(hbuf.view(b, l, c), context, context_lens, dtype, t=t)
        hip.gate_residual(xres, y.reshape(b * l, c).to(BF16).contiguous())
        hip.ln_modulate(xres, out=hbuf, eps=self.eps, shift=T[:, 3], scale=T[:, 4], row_index=idx, rows_per_batch=l)
        if self._fp8:
            mid = _proj_fp8(hbuf, pk, "w1", pk["b1"], epilogue=hip.EPI_GELU_TANH)
            a8, sa = hip.quantize_rows_fp8(mid)
            w8, sw = _fp8_weight(pk, "w2")
            hip.gemm_fp8_gate_residual(a8, sa, w8, sw, pk["b2"], xres, gate=T[:, 5], gate_row=idx, rows_per_batch=l)
        else:
            mid = hip.gemm(hbuf, pk["w1"], pk["b1"], epilogue=hip.EPI_GELU_TANH)
            hip.gemm_gate_residual(mid, pk["w2"], pk["b2"], xres, gate=T[:, 5], gate_row=idx, rows_per_batch=l)
        return xres.view(b, l, c)


class _Head(nn.Module):
    """Head (wan_transformer3d_FlexAM.py:475-507)."""

    def __init__(self, dim: int, out_features: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.head = _holder(out_features, dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)
        self.modulation_density = nn.Parameter(torch.randn(1, 1, dim) / dim ** 0.5)

    @torch.no_grad()
    def forward(self, x, e, density_emb):
        """x [B, L, C] fp32; e [B, C] or per-token [B, L, C]; density_emb [B, C] -> [B, L, out_features] fp32."""
        from . import hip
        b, l, c = x.shape
        dev = x.device
        meta = getattr(e, "_flexam_rows", None)
        if meta is not None:                                   # compact rows [R, C] + index from the model's own forward
            rows, idx, rpb = meta
            rows = rows.unsqueeze(1).expand(rows.shape[0], 2, c).contiguous()
        else:                                                  # e feeds both slots (shift, scale) of the head's modulation
            rows, idx, rpb = adaln_rows(e.unsqueeze(-2).expand(*e.shape[:-1], 2, c), b, l)
        tab = torch.empty(1, rows.shape[0], 2, c, device=dev, dtype=F32)
        hip.mod_table(_on(self.modulation, dev, F32), rows, tab, rpb, 0b10, _on(self.modulation_density, dev, F32),
                      density_emb.to(F32).reshape(b, 1, c).contiguous(), 0xF0)
        H = tab[0]
        hbuf = hip.ln_modulate(x.reshape(b * l, c).to(F32).contiguous(), eps=self.eps, shift=H[:, 0], scale=H[:, 1], row_index=idx,
                               rows_per_batch=l)
        w, bias = self.head.packed()
        return hip.gemm(hbuf, w, bias, out_dtype=F32).view(b, l, -1)


class WanTransformer3DModel_FlexAM(nn.Module):
    _supports_gradient_checkpointing = False

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192, freq_dim=256,
                 text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1), qk_norm=True, cross_attn_norm=True,
                 eps=1e-6, in_channels=16, hidden_size=2048, add_control_adapter=False, in_dim_control_adapter=24,
                 downscale_factor_control_adapter=8, add_ref_conv=False, in_dim_ref_conv=16, cross_attn_type=None,
                 add_cnn_block=False, in_dim_cnn_block=96, out_dim_cnn_block=16):
        super().__init__()
        cfg = dict(locals())
        cfg.pop("self")
        cfg.pop("__class__", None)
        cfg["patch_size"] = tuple(patch_size)
        self.config = ModelConfig(cfg)
        if add_control_adapter:
            raise NotImplementedError("control_adapter (camera branch) is not part of the FlexAM 5B path (SURVEY section 2)")
        if not (qk_norm and cross_attn_norm):
            raise NotImplementedError("FlexAM 5B uses qk_norm and cross_attn_norm; other variants are out of scope")
        if tuple(window_size) != (-1, -1):
            raise NotImplementedError("windowed attention is not used by FlexAM")
        if cross_attn_type not in (None, "cross_attn"):
            raise NotImplementedError(f"cross_attn_type {cross_attn_type}: only the text cross-attention of Wan2.2 is implemented")
        for k in ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim", "num_heads",
                  "num_layers", "eps"):
            setattr(self, k, self.config[k])
        pt, ph, pw = self.patch_size
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(_holder(dim, text_dim), nn.GELU(approximate="tanh"), _holder(dim, dim))
        self.time_embedding = nn.Sequential(_holder(dim, freq_dim), nn.SiLU(), _holder(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), _holder(dim * 6, dim))
        self.density_embedding = nn.Sequential(_holder(dim, freq_dim), nn.SiLU(), _holder(dim, dim))
        self.density_projection = nn.Sequential(nn.SiLU(), _holder(dim * 2, dim))
        self.blocks = nn.ModuleList([_Block(dim, ffn_dim, num_heads, eps) for _ in range(num_layers)])
        self.head = _Head(dim, pt * ph * pw * out_dim, eps)
        self.d = dim // num_heads
        self.ref_conv = nn.Conv2d(in_dim_ref_conv, dim, kernel_size=(ph, pw), stride=(ph, pw)) if add_ref_conv else None
        self.control_adapter = None
        if add_cnn_block:
            def stage(ci, co, groups):
                return nn.Sequential(nn.Conv3d(ci, co, kernel_size=(1, 3, 3), padding=(0, 1, 1)), nn.GroupNorm(groups, co), nn.SiLU())
            self.cnn_conv1, self.cnn_conv2 = stage(in_dim_cnn_block, 192, 24), stage(192, 192, 24)
            self.cnn_conv3, self.cnn_conv4 = stage(192, 96, 12), stage(96, 96, 12)
            self.cnn_conv5 = nn.Conv3d(96, out_dim_cnn_block, kernel_size=(1, 1, 1))
        else:
            self.cnn_conv1 = self.cnn_conv2 = self.cnn_conv3 = self.cnn_conv4 = self.cnn_conv5 = None
        self.freqs = rope_angle_table(1024, self.d)          # angle table (the reference stores exp(i*angle))
        self._riflex = None
        self.teacache = None
        self.cfg_skip_ratio = None
        self.current_steps = 0
        self.num_inference_steps = None
        self.gradient_checkpointing = False
        self.sp_world_size, self.sp_world_rank, self._sp_group = 1, 0, None
        self._parallel = None
        self._engine: Optional[DiTEngine] = None
        self._engine_sig = None
        self._cond_key = None
        self._cond_ident = None
        self.init_weights()

    # ------------------------------------------------------------------ parameters
    def init_weights(self):
        """Same scheme as the reference (wan_transformer3d_FlexAM.py:1151-1188)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for seq in (self.text_embedding, self.time_embedding):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, std=0.02)
        for seq in (self.density_embedding, self.density_projection):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.zeros_(m.weight)
                    nn.init.zeros_(m.bias)
        nn.init.zeros_(self.head.head.weight)

    def randomize_zero_init(self, std: float = 0.02, seed: int = 0):
        """Benchmarks / tests: redraw the tensors the reference zero-initialises (head, density MLPs)
        so random-init outputs are non-trivial."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in [self.head.head.weight] + [q for s in (self.density_embedding, self.density_projection) for q in s.parameters()]:
                p.copy_((torch.randn(p.shape, generator=g) * std).to(p.dtype))
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    @property
    def dtype(self):
        return self.patch_embedding.weight.dtype

    @property
    def device(self):
        return self.patch_embedding.weight.device

    def _rope_angles(self):
        if self._riflex is not None:
            k, l_test, scale = self._riflex
            return rope_angle_table(1024, self.d, riflex_k=k, riflex_l_test=l_test, riflex_scale=scale)
        return rope_angle_table(1024, self.d)

    def _signature(self):
        """Everything the engine's device-side packs depend on: identity of each block module (a replaced block), re-bound
        forwards, and (storage pointer, version, dtype, device) of every parameter.  ~1000 tuples: microseconds per call."""
        blocks = tuple((id(b), getattr(b, "pristine", lambda: False)()) for b in self.blocks)
        return blocks, tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in self.parameters())

    def invalidate_engine(self):
        """Drops the device-side parameter packs and per-clip caches.  Needed only after edits the version counters cannot see
        (`param.data` arithmetic on fp32 / bias / norm parameters; bf16 weight matrices are shared with the engine, not copied)."""
        self._engine = None
        self._cond_key = None
        self._cond_ident = None

    def engine(self) -> DiTEngine:
        if self._engine is not None and self._engine_sig != self._signature():
            self._engine = None                    # a block was replaced / re-bound, or a parameter changed in place
        if self._engine is None:
            self._engine = DiTEngine(self)
            self._engine_sig = self._signature()   # after packing: fused q|k|v buffers re-point the parameters they alias
            self._cond_key = None
            if self._parallel is not None:
                self._engine.set_parallel(**self._parallel)
            if getattr(self, "_fp8", False):
                self._engine.enable_fp8(True)
        return self._engine

    def trust_conditioning_identity(self, on: bool = True):
        """Opt-in shortcut for callers that pass the SAME conditioning tensor objects on every step and never write them behind
        PyTorch's back: when every tensor of a call is the very object of the last call (same storage, same version counter) the
        content key of that call is reused and forward() makes no readback.  Off by default: `_version` only counts PyTorch's own
        in-place operations -- a buffer refilled through `.data`, DLPack, a raw pointer (this library's own out= kernels) or another
        framework keeps its version, and the shortcut would then serve the previous clip's per-clip state.  `invalidate_conditioning()`
        drops the remembered identity by hand."""
        self._trust_ident = bool(on)
        self._cond_ident = None

    def invalidate_conditioning(self):
        """Forget the last call's conditioning (its content key and, with trust_conditioning_identity, its identity)."""
        self._cond_key = None
        self._cond_ident = None

    def _conditioning_key(self, context, y, full_ref, additional_control, density, latent_shape):
        """Content key of the step-invariant inputs.  The reference sampler rebuilds them with torch.cat on every step
        (PIPE.py:850-886), so identity says nothing about a change: one checksum launch per tensor into one buffer and ONE readback
        (~120 MB hashed in microseconds, against the cnn-block, the text MLP and 30 cross-K/V GEMMs a hit saves).  The RoPE variant
        is part of the key: the rotation tables live in the per-clip state.  With trust_conditioning_identity(True) a call whose
        tensors are the very objects of the last call (weak references still alive, same storage, same version counter) skips
        the readback; nothing is kept alive for it."""
        import weakref
        from . import hip
        tensors = [v for v in (y, full_ref, additional_control, density, *context) if v is not None]
        nones = tuple(v is None for v in (y, full_ref, additional_control, density))
        trust = getattr(self, "_trust_ident", False)
        if trust:
            ident = (tuple(latent_shape), self._riflex, nones, tuple((v.data_ptr(), v._version, tuple(v.shape), v.dtype) for v in tensors))
            last = self._cond_ident
            if (last is not None and last[0] == ident and len(last[2]) == len(tensors)
                    and all(r() is v for r, v in zip(last[2], tensors))):
                return last[1]
        sums = hip.checksums(tensors)
        key = (tuple(latent_shape), self._riflex, nones, tuple((tuple(v.shape), str(v.dtype), c) for v, c in zip(tensors, sums)))
        self._cond_ident = (ident, key, [weakref.ref(v) for v in tensors]) if trust else None
        return key

    # ------------------------------------------------------------------ feature switches (reference API)
    def enable_teacache(self, coefficients, num_steps: int, rel_l1_thresh: float, num_skip_start_steps: int = 0, offload: bool = True):
        self.teacache = TeaCache(coefficients, num_steps, rel_l1_thresh=rel_l1_thresh, num_skip_start_steps=num_skip_start_steps,
                                 offload=offload)

    def share_teacache(self, transformer=None):
        self.teacache = transformer.teacache

    def disable_teacache(self):
        self.teacache = None

    def enable_cfg_skip(self, cfg_skip_ratio, num_steps):
        if cfg_skip_ratio != 0:
            self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = cfg_skip_ratio, 0, num_steps
        else:
            self.disable_cfg_skip()

    def share_cfg_skip(self, transformer=None):
        self.cfg_skip_ratio = transformer.cfg_skip_ratio
        self.current_steps = transformer.current_steps
        self.num_inference_steps = transformer.num_inference_steps

    def disable_cfg_skip(self):
        self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = None, 0, None

    def enable_fp8_gemm(self, on: bool = True):
        """This build's extension (BASELINE.json configs[4]): QKV and FFN projections on fp8 (OCP e4m3) MFMA with per-row /
        per-channel scales; see DiTEngine.enable_fp8.  Off by default; parity tolerance is the fp8 one (tests/test_fp8_gpu.py)."""
        self._fp8 = bool(on)
        for mod in self.modules():                 # every native block, also those inside a caller's wrapper modules (blocks[i] = wrapper(block))
            if isinstance(mod, _Block):
                mod.set_fp8(self._fp8)
        if self._engine is not None:
            self._engine.enable_fp8(self._fp8)

    def enable_riflex(self, k=6, L_test=66, L_test_scale=4.886):
        self._riflex = (k, L_test, L_test_scale)
        self._rope_changed()

    def disable_riflex(self):
        self._riflex = None
        self._rope_changed()

    def _rope_changed(self):
        """The rotation tables are part of the engine's per-clip state (DiTEngine.set_conditioning): drop it with the angles,
        or a forward() with unchanged conditioning would keep rotating by the old table."""
        self.freqs = self._rope_angles()
        self._cond_key = None
        self._cond_ident = None
        if self._engine is not None:
            self._engine._angles = None
            self._engine.cond = None

    @staticmethod
    def default_cfg_parallel(world: int, num_heads: int, sp_mode: str = "allgather") -> bool:
        """Default layout of `world` ranks: split the CFG pair first (2 x world/2) unless the all-to-all exchange is selected and
        the heads divide over all ranks (then world >= 4 runs world-way token chunks with the pair batched).  `sp_mode` is
        FLEXAM_SP_MODE, whose default ("allgather") is DiTEngine's."""
        a2a = sp_mode == "ulysses" and num_heads % world == 0
        return world % 2 == 0 and (world == 2 or not a2a)

    def enable_multi_gpus_inference(self, group=None, cfg_parallel=None):
        """Multi-GPU inference over `group` (default: the world group).  Stands in for the reference's missing
        FlexAM/dist + xfuser USP (wan_transformer3d_FlexAM.py:801-815).  Layout (flexam_amd/dist.py):
          * cfg_parallel: the two classifier-free-guidance rows are independent until the guidance combine
            (PIPE.py:926-928), so a split by CFG row has no per-block traffic at all.  Default: on for TWO ranks (no exchange
            inside the blocks), and for larger even groups only when the all-to-all exchange is not available (heads not
            divisible by the ranks, or FLEXAM_SP_MODE=allgather).  With the all-to-all ("ulysses") exchange, N >= 4 ranks run
            pure sequence parallelism with the CFG pair batched on every rank: on the xGMI full mesh each of the N-1 peer links
            then carries 1/N of a rank's q|k|v, half of what a link carries when 2 x N/2 ranks only talk inside their half;
          * inside a sequence-parallel group, contiguous token chunks per rank and one exchange around self-attention per block;
          * one all-gather of the head output per step over the whole group."""
        import torch.distributed as dist
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        if cfg_parallel is None:
            cfg_parallel = self.default_cfg_parallel(world, self.num_heads, os.environ.get("FLEXAM_SP_MODE", "allgather"))
        if cfg_parallel and world % 2:
            raise ValueError("cfg_parallel needs an even number of ranks")
        if cfg_parallel:
            sp = world // 2
            members = dist.get_process_group_ranks(group) if group is not None else list(range(world))
            from .dist import subgroup
            halves = [subgroup(members[i * sp:(i + 1) * sp]) for i in range(2)]     # every rank creates both, once per process
            self._parallel = dict(sp_group=halves[rank // sp], sp_rank=rank % sp, sp_size=sp, world_group=group, world_size=world,
                                  cfg_size=2, cfg_row=rank // sp)
            self.sp_world_size, self.sp_world_rank = sp, rank % sp
        else:
            self._parallel = dict(sp_group=group, sp_rank=rank, sp_size=world, world_group=group, world_size=world)
            self.sp_world_size, self.sp_world_rank = world, rank
        self._sp_group = self._parallel["sp_group"]
        if self._engine is not None:
            self._engine.set_parallel(**self._parallel)

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _timestep_rows(t: torch.Tensor, batch: int, seq_len: int, ref_len: int, padded_len: int = None):
        """Per-token timesteps -> (distinct rows [B*U], int32 row index [B*L], U).  t [B, Lt]; seq_len = the L real tokens;
        padded_len = the caller's seq_len + ref_len when that is longer (a padded sequence, FX.py:918-925: the reference lines the
        timesteps up against the PADDED sequence and the real tokens are its first L)."""
        total = max(seq_len, padded_len or 0)
        if ref_len and t.size(1) < total:                         # FX.py:900-904: ref tokens take the last token's t
            t = torch.cat([t[:, -1:].repeat(1, total - t.size(1)), t], dim=1)
        if t.size(1) < total:                                     # FX.py:930-934
            t = torch.cat([t, t[:, -1:].repeat(1, total - t.size(1))], dim=1)
        t = t[:, :seq_len]
        uniq, inv = [], []
        for b in range(batch):
            u, i = torch.unique(t[b].float(), return_inverse=True)
            uniq.append(u)
            inv.append(i)
        U = max(u.numel() for u in uniq)
        rows = torch.stack([torch.cat([u, u[-1:].repeat(U - u.numel())]) for u in uniq]).reshape(-1)
        index = torch.stack([i + b * U for b, i in enumerate(inv)]).to(torch.int32).reshape(-1)
        return rows, index, U

    @cfg_skip()
    @torch.no_grad()
    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, y_camera=None, full_ref=None, subject_ref=None,
                cond_flag=True, additional_control=None, density=None):
        """x [B,C,F,H,W]; t [B] or [B, F*H*W/4]; context: list of B [len_i, text_dim]; y [B,100,F,H,W];
        full_ref [B,C,H,W]; additional_control [B,240,F,H,W]; density [B] -> [B,out_dim,F,H,W]."""
        if clip_fea is not None or y_camera is not None or subject_ref is not None:
            raise NotImplementedError("clip_fea / y_camera / subject_ref are not used by the FlexAM 5B path")
        if density is None:
            raise ValueError("density is mandatory for the FlexAM DiT (wan_transformer3d_FlexAM.py:951-955,1037)")
        eng = self.engine()
        if isinstance(x, (list, tuple)):
            x = torch.stack(list(x))
        B_full = x.shape[0]
        if eng.cfg_size == 2:                      # this rank computes one of the two CFG rows
            if B_full != 2:
                raise NotImplementedError("cfg-parallel inference expects the CFG pair (batch 2)")
            r = eng.cfg_row
            sl = lambda v: v[r:r + 1] if torch.is_tensor(v) else ([v[r]] if isinstance(v, (list, tuple)) else v)
            x, t, context, y, full_ref, additional_control, density = (sl(v) for v in (x, t, context, y, full_ref, additional_control, density))
        B = x.shape[0]
        dev = eng.device
        mv = lambda v: v.to(dev) if torch.is_tensor(v) else v
        y, full_ref, additional_control, density = mv(y), mv(full_ref), mv(additional_control), mv(density)
        context = [u.to(dev) for u in context]
        # step-invariant work (cnn-block, static patch columns, ref tokens, text MLP, cross K/V, density MLP): once per distinct
        # conditioning, not once per call -- the unmodified reference loop (PIPE.py:912-923) gets the hoisting too
        key = self._conditioning_key(context, y, full_ref, additional_control, density, tuple(x.shape[1:]))
        if key != self._cond_key or eng.cond is None:
            eng.set_conditioning(context, y, full_ref, additional_control, density, tuple(x.shape[1:]))
            self._cond_key = key
        cond = eng.cond
        L, ref_len = cond["L"], cond["ref_len"]
        if seq_len + ref_len < L:
            raise AssertionError(f"seq_len {seq_len} is shorter than the token sequence {L - ref_len}")
        t = t.to(dev)
        if t.dim() == 1:
            rows, index, U, shared = t.float(), None, 1, False
        else:
            rows, index, U = self._timestep_rows(t, B, L, ref_len, seq_len + ref_len)
            index = index.to(dev)
            shared = B > 1 and bool((rows.view(B, U) == rows.view(B, U)[0]).all())
        head_local = eng.run(x, rows, index, U, teacache=self.teacache, cond_flag=cond_flag, rows_shared=shared)
        if self.teacache is not None and cond_flag:          # FX.py:1119-1122
            self.teacache.cnt += 1
            if self.teacache.cnt == self.teacache.num_steps:
                self.teacache.reset()
        tokens = eng.gather_tokens(head_local)
        from . import hip
        c, f, h, w = x.shape[1:]
        # odd latent sizes: the stride-2 patch convolution drops the last row / column and unpatchify returns the cropped size
        # (FX.py:885,1126-1149).  `seq_len` longer than the token sequence (the pipeline's ceil(), PIPE.py:838-839) pads the reference's
        # sequence with zero tokens that flash-attention masks as keys (k_lens, FX.py:918-925,251-256) and unpatchify drops: the real
        # tokens see nothing of them, so the engine simply does not create them.
        he, we = h // 2 * 2, w // 2 * 2
        out = torch.empty(B_full, self.out_dim, f, he, we, device=eng.device, dtype=x.dtype if x.dtype in (F32, torch.bfloat16) else F32)
        for b in range(B_full):
            hip.unpatchify(tokens[b], ref_len, self.out_dim, f, he, we, out=out[b])
        return out

    # ------------------------------------------------------------------ checkpoints
    @classmethod
    def from_config(cls, config: dict, **extra):
        import inspect
        allowed = set(inspect.signature(WanTransformer3DModel_FlexAM.__init__).parameters) - {"self"}
        kw = {k: v for k, v in {**config, **extra}.items() if k in allowed}
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, transformer_additional_kwargs={}, low_cpu_mem_usage=False,
                        torch_dtype=torch.bfloat16):
        """Reads `config.json` + `diffusion_pytorch_model.{bin,safetensors}` or sharded *.safetensors
        with the reference's conventions (wan_transformer3d_FlexAM.py:1190-1332): yaml dict_mapping,
        zero-pad / crop of patch_embedding in-channels, size-mismatched keys skipped, strict=False."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file) as fh:
            config = json.load(fh)
        extra = dict(transformer_additional_kwargs)
        for src, dst in extra.pop("dict_mapping", {}).items():
            extra[dst] = config[src]
        model = cls.from_config(config, **extra)
        bin_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        st_file = bin_file.replace(".bin", ".safetensors")
        if os.path.exists(bin_file):
            state = torch.load(bin_file, map_location="cpu")
        else:
            from safetensors.torch import load_file
            files = [st_file] if os.path.exists(st_file) else sorted(glob.glob(os.path.join(pretrained_model_path, "*.safetensors")))
            state = {}
            for fpath in files:
                state.update(load_file(fpath))
        own = model.state_dict()
        pe = "patch_embedding.weight"
        if pe in state and own[pe].shape != state[pe].shape:
            merged = torch.zeros_like(own[pe])
            n = min(own[pe].shape[1], state[pe].shape[1])
            merged[:, :n] = state[pe][:, :n]
            state[pe] = merged
        state = {k: v for k, v in state.items() if k in own and own[k].shape == v.shape}
        missing, unexpected = model.load_state_dict(state, strict=False)
        print(f"### missing keys: {len(missing)}; \n### unexpected keys: {len(unexpected)};")
        return model.to(torch_dtype)


class Wan2_2Transformer3DModel_FlexAM(WanTransformer3DModel_FlexAM):
    """Wan2.2 variant: text-only cross-attention (wan_transformer3d_FlexAM.py:1335-1438)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("cross_attn_type", None)
        super().__init__(*args, cross_attn_type="cross_attn", **kwargs)


# The reference's module-level class names (wan_transformer3d_FlexAM.py:173,192,205,353,381,475): code that imports or
# isinstance-checks them finds the HIP modules.
WanRMSNorm = _Norm
WanLayerNorm = _Norm
WanSelfAttention = _SelfAttn
WanCrossAttention = _CrossAttn
WanAttentionBlock = _Block
Head = _Head
WAN_CROSSATTENTION_CLASSES = {"cross_attn": _CrossAttn}       # wan_transformer3d_FlexAM.py:374-378 (the Wan2.2 text-only entry)
