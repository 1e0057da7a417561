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


def _holder(n_out: int, n_in: int) -> nn.Linear:
    return nn.Linear(n_in, n_out)


class _Norm(nn.Module):
    def __init__(self, dim: int, bias: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim))


class _Attn(nn.Module):
    """q/k/v/o projections + full-width RMSNorm weights (keys: q.weight ... norm_k.weight)."""

    def __init__(self, dim: int):
        super().__init__()
        self.q, self.k, self.v, self.o = (_holder(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = _Norm(dim), _Norm(dim)


class _Block(nn.Module):
    def __init__(self, dim: int, ffn_dim: int):
        super().__init__()
        self.self_attn, self.cross_attn = _Attn(dim), _Attn(dim)
        self.norm3 = _Norm(dim, bias=True)
        self.ffn = nn.Sequential(_holder(ffn_dim, dim), nn.GELU(approximate="tanh"), _holder(dim, ffn_dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self.modulation_density = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)


class _Head(nn.Module):
    def __init__(self, dim: int, out_features: int):
        super().__init__()
        self.head = _holder(out_features, dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)
        self.modulation_density = nn.Parameter(torch.randn(1, 1, dim) / dim ** 0.5)


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
        self.blocks = nn.ModuleList([_Block(dim, ffn_dim) for _ in range(num_layers)])
        self.head = _Head(dim, pt * ph * pw * out_dim)
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

    def engine(self) -> DiTEngine:
        if self._engine is None:
            self._engine = DiTEngine(self)
            if self._parallel is not None:
                self._engine.set_parallel(**self._parallel)
        return self._engine

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

    def enable_riflex(self, k=6, L_test=66, L_test_scale=4.886):
        self._riflex = (k, L_test, L_test_scale)
        self.freqs = self._rope_angles()
        if self._engine is not None:
            self._engine._angles = None

    def disable_riflex(self):
        self._riflex = None
        self.freqs = self._rope_angles()
        if self._engine is not None:
            self._engine._angles = None

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
            a2a = os.environ.get("FLEXAM_SP_MODE", "ulysses") == "ulysses" and self.num_heads % world == 0
            cfg_parallel = world % 2 == 0 and (world == 2 or not a2a)
        if cfg_parallel and world % 2:
            raise ValueError("cfg_parallel needs an even number of ranks")
        if cfg_parallel:
            sp = world // 2
            members = dist.get_process_group_ranks(group) if group is not None else list(range(world))
            halves = [dist.new_group(members[i * sp:(i + 1) * sp]) for i in range(2)]     # every rank creates both
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
    def _timestep_rows(t: torch.Tensor, batch: int, seq_len: int, ref_len: int):
        """Per-token timesteps -> (distinct rows [B*U], int32 row index [B*L], U).  t [B, Lt]."""
        if ref_len and t.size(1) < seq_len:                       # FX.py:900-904: ref tokens take the last token's t
            t = torch.cat([t[:, -1:].repeat(1, seq_len - t.size(1)), t], dim=1)
        if t.size(1) < seq_len:                                   # FX.py:930-934
            t = torch.cat([t, t[:, -1:].repeat(1, seq_len - t.size(1))], dim=1)
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
        cond = eng.set_conditioning(list(context), y, full_ref, additional_control, density, tuple(x.shape[1:]))
        L, ref_len = cond["L"], cond["ref_len"]
        if seq_len + ref_len < L:
            raise AssertionError(f"seq_len {seq_len} is shorter than the token sequence {L - ref_len}")
        t = t.to(eng.device)
        if t.dim() == 1:
            rows, index, U = t.float(), None, 1
        else:
            rows, index, U = self._timestep_rows(t, B, L, ref_len)
            index = index.to(eng.device)
        head_local = eng.run(x, rows, index, U, teacache=self.teacache, cond_flag=cond_flag)
        if self.teacache is not None and cond_flag:          # FX.py:1119-1122
            self.teacache.cnt += 1
            if self.teacache.cnt == self.teacache.num_steps:
                self.teacache.reset()
        tokens = eng.gather_tokens(head_local)
        from . import hip
        c, f, h, w = x.shape[1:]
        out = torch.empty(B_full, self.out_dim, f, h, w, device=eng.device, dtype=x.dtype if x.dtype in (F32, torch.bfloat16) else F32)
        for b in range(B_full):
            hip.unpatchify(tokens[b], ref_len, self.out_dim, f, h, w, out=out[b])
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
