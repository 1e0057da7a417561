"""MI355X drop-in for `WanT5EncoderModel` (umT5-xxl text encoder of Wan2.2 / FlexAM).

Reference: FlexAM/models/wan_text_encoder.py:256-305 (forward :291-305, from_pretrained :306-394), layers :44-253.
Same constructor arguments, parameter names (`token_embedding.weight`, `blocks.i.{norm1,norm2}.weight`,
`blocks.i.attn.{q,k,v,o}.weight`, `blocks.i.ffn.{gate.0,fc1,fc2}.weight`, `blocks.i.pos_embedding.embedding.weight`
or `pos_embedding.embedding.weight`, `norm.weight`) and `forward(input_ids, attention_mask) -> (hidden,)`.

The encoder runs twice per clip on <= 512 tokens (PIPE.py:190-232), so it is built from the library's existing
pieces rather than a dedicated fused kernel: q|k|v, o, gate, fc1, fc2 are `flexam_gemm_bf16` launches (tanh-GELU and
the fp32 residual add in the epilogues); T5 attention has 64-wide heads, no 1/sqrt(d) scaling, an additive
relative-position bias and a padding mask, which the head_dim-128 flash kernel does not cover -- per head it is
S = Q K^T (GEMM, fp32), one `flexam_softmax_bias_rows` launch over all heads, O = P V (GEMM).  The bucket
lookup of the relative-position table is integer indexing done with torch (plumbing).  No CPU fallback.
"""
import inspect
import math
from typing import Optional

import torch
import torch.nn as nn

from . import hip
from .wan_transformer3d_FlexAM import ModelConfig

BF16, F32 = torch.bfloat16, torch.float32


def _relative_buckets(lq: int, lk: int, num_buckets: int, device, max_dist: int = 128) -> torch.Tensor:
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (wan_text_encoder.py:219-253)."""
    rel = torch.arange(lk, device=device).unsqueeze(0) - torch.arange(lq, device=device).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


class _Holder(nn.Module):
    """A module that only owns parameters under the reference's names."""


def _linear(out_f, in_f):
    m = _Holder()
    m.weight = nn.Parameter(torch.empty(out_f, in_f))
    return m


class _Engine:
    def __init__(self, model):
        self.device = dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("flexam_amd: the text encoder runs only on a GPU through libflexam_hip.so (no CPU fallback)")
        hip.device_check()
        sd = model.state_dict()
        bf = lambda k: sd[k].detach().to(dev, BF16).contiguous()
        f32 = lambda k: sd[k].detach().to(dev, F32).contiguous()
        self.emb = sd["token_embedding.weight"].detach().to(dev)
        self.norm = f32("norm.weight")
        self.shared = f32("pos_embedding.embedding.weight") if model.shared_pos else None
        self.layers = []
        for i in range(model.num_layers):
            p = f"blocks.{i}."
            self.layers.append(dict(
                n1=f32(p + "norm1.weight"), n2=f32(p + "norm2.weight"),
                wqkv=torch.cat([bf(p + "attn.q.weight"), bf(p + "attn.k.weight"), bf(p + "attn.v.weight")]).contiguous(),
                wo=bf(p + "attn.o.weight"), wg=bf(p + "ffn.gate.0.weight"), w1=bf(p + "ffn.fc1.weight"), w2=bf(p + "ffn.fc2.weight"),
                pos=None if model.shared_pos else f32(p + "pos_embedding.embedding.weight")))

    def bias(self, table, buckets):
        """[num_buckets, N] table -> fp32 [N * L, L] (row = head * L + query), the layout of the stacked score matrix."""
        n = table.shape[1]
        l = buckets.shape[0]
        return table[buckets].permute(2, 0, 1).reshape(n * l, l).contiguous()


class WanT5EncoderModel(nn.Module):
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        if dim_attn % num_heads or (dim_attn // num_heads) % 64 or dim % 64 or dim_ffn % 64:
            raise ValueError("WanT5EncoderModel (HIP): head_dim, dim and dim_ffn must be multiples of 64")
        self.config = ModelConfig(vocab=vocab, dim=dim, dim_attn=dim_attn, dim_ffn=dim_ffn, num_heads=num_heads, num_layers=num_layers,
                                  num_buckets=num_buckets, shared_pos=shared_pos, dropout=dropout)
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)

        def rel():
            m = _Holder()
            m.embedding = nn.Embedding(num_buckets, num_heads)
            return m
        self.pos_embedding = rel() if shared_pos else None
        self.blocks = nn.ModuleList()
        for _ in range(num_layers):
            b = _Holder()
            b.norm1, b.norm2 = _Holder(), _Holder()
            b.norm1.weight, b.norm2.weight = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.ones(dim))
            b.attn = _Holder()
            b.attn.q, b.attn.k, b.attn.v, b.attn.o = _linear(dim_attn, dim), _linear(dim_attn, dim), _linear(dim_attn, dim), _linear(dim, dim_attn)
            b.ffn = _Holder()
            b.ffn.gate = nn.ModuleList([_linear(dim_ffn, dim)])
            b.ffn.fc1, b.ffn.fc2 = _linear(dim_ffn, dim), _linear(dim, dim_ffn)
            b.pos_embedding = None if shared_pos else rel()
            self.blocks.append(b)
        self.norm = _Holder()
        self.norm.weight = nn.Parameter(torch.ones(dim))
        self._init_weights()
        self._engine: Optional[_Engine] = None

    def _init_weights(self):
        """init_weights of the reference (wan_text_encoder.py:21-35)."""
        d, da, df, n = self.dim, self.dim_attn, self.dim_ffn, self.num_heads
        with torch.no_grad():
            for b in self.blocks:
                nn.init.normal_(b.attn.q.weight, std=(d * da) ** -0.5)
                nn.init.normal_(b.attn.k.weight, std=d ** -0.5)
                nn.init.normal_(b.attn.v.weight, std=d ** -0.5)
                nn.init.normal_(b.attn.o.weight, std=(n * (da // n)) ** -0.5)
                nn.init.normal_(b.ffn.gate[0].weight, std=d ** -0.5)
                nn.init.normal_(b.ffn.fc1.weight, std=d ** -0.5)
                nn.init.normal_(b.ffn.fc2.weight, std=df ** -0.5)
                if b.pos_embedding is not None:
                    nn.init.normal_(b.pos_embedding.embedding.weight, std=(2 * self.num_buckets * n) ** -0.5)
            if self.pos_embedding is not None:
                nn.init.normal_(self.pos_embedding.embedding.weight, std=(2 * self.num_buckets * n) ** -0.5)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    @property
    def dtype(self):
        return self.norm.weight.dtype

    @property
    def device(self):
        return self.norm.weight.device

    def engine(self) -> _Engine:
        if self._engine is None:
            self._engine = _Engine(self)
        return self._engine

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.LongTensor] = None, attention_mask: Optional[torch.FloatTensor] = None):
        eng = self.engine()
        dev, d, da, df, n = eng.device, self.dim, self.dim_attn, self.dim_ffn, self.num_heads
        c = da // n
        b, l = input_ids.shape
        if l % 4:
            raise ValueError(f"WanT5EncoderModel (HIP): sequence length {l} must be a multiple of 4 (the pipeline pads to max_sequence_length)")
        lp = (l + 63) // 64 * 64                                                   # K granularity of the P.V GEMM
        x = eng.emb[input_ids.to(dev)].to(F32).reshape(b * l, d).contiguous()      # embedding lookup (dropout: eval)
        buckets = _relative_buckets(l, l, self.num_buckets, dev)
        shared_bias = eng.bias(eng.shared, buckets) if eng.shared is not None else None
        masks = None if attention_mask is None else attention_mask.to(dev, F32).contiguous()
        h = torch.empty(b * l, d, device=dev, dtype=BF16)
        ao = torch.empty(b * l, da, device=dev, dtype=BF16)
        s = torch.empty(n * l, l, device=dev, dtype=F32)
        p = torch.empty(n * l, lp, device=dev, dtype=BF16)
        for ly in eng.layers:
            bias = shared_bias if shared_bias is not None else eng.bias(ly["pos"], buckets)
            hip.t5_norm(x, ly["n1"], h)
            qkv = hip.gemm(h, ly["wqkv"])                                          # [b*l, 3*da] bf16
            for bi in range(b):
                rows = qkv[bi * l:(bi + 1) * l]
                vt = torch.zeros(n, c, lp, device=dev, dtype=BF16)
                vt[:, :, :l] = rows[:, 2 * da:].view(l, n, c).permute(1, 2, 0)      # V^T per head (re-layout, plumbing)
                for hd in range(n):
                    hip.gemm(rows[:, hd * c:(hd + 1) * c], rows[:, da + hd * c:da + (hd + 1) * c], out=s[hd * l:(hd + 1) * l])
                hip.softmax_bias_rows(s, p, l, 1.0, bias, masks[bi] if masks is not None else None)
                for hd in range(n):
                    hip.gemm(p[hd * l:(hd + 1) * l], vt[hd], out=ao[bi * l:(bi + 1) * l, hd * c:(hd + 1) * c])
            hip.gemm_gate_residual(ao, ly["wo"], None, x)                          # x += attn.o(...)
            hip.t5_norm(x, ly["n2"], h)
            gate = hip.gemm(h, ly["wg"], epilogue=hip.EPI_GELU_TANH)
            f1 = hip.gemm(h, ly["w1"])
            hip.gemm_gate_residual(hip.mul_bf16(f1, gate), ly["w2"], None, x)      # x += fc2(fc1(h) * gelu(gate(h)))
        out = torch.empty(b * l, d, device=dev, dtype=F32)
        hip.t5_norm(x, eng.norm, out)
        out = out.view(b, l, d)
        return (out.to(self.dtype) if self.dtype != F32 else out,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs={}, low_cpu_mem_usage=False, torch_dtype=torch.bfloat16):
        """wan_text_encoder.py:306-394: a flat state dict (.safetensors or torch pickle), strict=False, cast to torch_dtype."""
        allowed = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in additional_kwargs.items() if k in allowed})
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            state = load_file(pretrained_model_path)
        else:
            state = torch.load(pretrained_model_path, map_location="cpu")
        m, u = model.load_state_dict(state, strict=False)
        print(f"### missing keys: {len(m)}; \\n### unexpected keys: {len(u)};")
        return model.to(torch_dtype)
