"""HIP execution engine of the FlexAM DiT (Wan2.2-Fun-5B-FLEXAM) on one MI355X, optionally one
shard of a sequence-parallel group.

What the reference does per denoise step in wan_transformer3d_FlexAM.py:817-1123 is split here by
how often it changes:

  per clip   (`set_conditioning`)  cnn-block over the control/depth/cos latents (:869-881), the
             100 step-invariant input channels of the patch embedding (:883-885), ref_conv tokens
             (:895-899), text embedding (:958-964) and every block's cross-attention K/V
             (:364-365), density embedding (:950-955), RoPE rows (:137-164)
  per step   (`run`)  48-channel patchify + patch GEMM, the time-embedding MLP on the DISTINCT
             timesteps only (two rows in the sampler: frame-0 tokens have t = 0, PIPE.py:891-898),
             one AdaLN table for all blocks, 30 blocks, head

Data layout in HBM (per rank): residual stream x fp32 [B*Lc, C] (token-major, rows of 12 KiB);
GEMM operands bf16 row-major with K contiguous; q|k|v of a block in one [B*Lc, 3C] buffer (heads
packed along the row, so attention addresses head h at column h*128); AdaLN rows are looked up per
token through an int32 row index instead of the reference's materialised [B, L, 6, C] fp32 tensor.
All arithmetic runs in libflexam_hip.so (flexam_amd/hip.py); torch only allocates and slices.
"""
import math
import os
from contextlib import nullcontext
from typing import List, Optional

import torch

from . import hip
from .rope import rope_angle_table, rope_tables

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class _ConvCL:
    """A (1,kh,kw) convolution as an implicit GEMM over a padded channels-last image."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, cp_in: int, device):
        co, ci = weight.shape[0], weight.shape[1]
        kh, kw = weight.shape[-2], weight.shape[-1]
        w = weight.detach().to(device, F32).reshape(co, ci, kh, kw).permute(0, 2, 3, 1)        # [co, kh, kw, ci]
        wp = torch.zeros(co, kh, kw, cp_in, device=device, dtype=F32)
        wp[..., :ci] = w
        self.weight = wp.reshape(co, kh * kw * cp_in).to(BF16).contiguous()
        self.bias = bias.detach().to(device, F32).contiguous()
        self.kh, self.kw, self.cp_in, self.cout = kh, kw, cp_in, co
        self._koff = {}

    def koff(self, wp_img: int, device) -> torch.Tensor:
        if wp_img not in self._koff:
            offs = []
            for dh in range(self.kh):
                for dw in range(self.kw):
                    base = ((dh - self.kh // 2) * wp_img + (dw - self.kw // 2)) * self.cp_in
                    offs += [base + cb * 64 for cb in range(self.cp_in // 64)]
            self._koff[wp_img] = torch.tensor(offs, dtype=I64, device=device)
        return self._koff[wp_img]


class _Image:
    """Zero-padded channels-last bf16 image [F, H+2, W+2, Cp] with guard rows on both sides so that
    every 3x3 tap offset of every padded position stays inside the allocation."""

    def __init__(self, f, h, w, cp, device):
        self.f, self.h, self.w, self.cp = f, h, w, cp
        self.rows = f * (h + 2) * (w + 2)
        guard = (w + 2) + 1
        self.buf = torch.zeros((self.rows + 2 * guard) * cp, device=device, dtype=BF16)
        self.img = self.buf[guard * cp:(guard + self.rows) * cp].view(f, h + 2, w + 2, cp)
        self.mat = self.img.view(self.rows, cp)


class DiTEngine:
    _sage_warned = False

    def __init__(self, model):
        self.model = model
        c = model.config
        self.dim, self.ffn, self.nh, self.nl = c["dim"], c["ffn_dim"], c["num_heads"], c["num_layers"]
        self.hd = self.dim // self.nh
        self.eps = c["eps"]
        self.patch = tuple(c["patch_size"])
        self.out_dim, self.in_dim = c["out_dim"], c["in_dim"]
        self.text_len, self.freq_dim = c["text_len"], c["freq_dim"]
        if self.hd != 128:
            raise RuntimeError(f"flexam_amd: head_dim {self.hd} unsupported by the HIP attention kernel (128 only)")
        if self.patch != (1, 2, 2):
            raise RuntimeError("flexam_amd: only patch_size (1,2,2) is implemented")
        self.device = model.patch_embedding.weight.device
        if self.device.type != "cuda":
            raise RuntimeError("flexam_amd: the DiT runs only on a GPU through libflexam_hip.so "
                               "(no CPU or eager fallback); move the model to cuda first")
        hip.device_check()
        self.sp_group = None
        self.sp_rank, self.sp_size = 0, 1
        self.world_group, self.world_size = None, 1
        self.cfg_size, self.cfg_row = 1, 0          # cfg_size 2: this rank computes only CFG row `cfg_row`
        self._ws = {}
        self._ws_gen = 0                            # bumped whenever the activation buffers are dropped: recorded launch plans name their addresses
        self._angles = None
        self.cond = None
        self.n_conditioning = 0                     # set_conditioning runs so far (tests: the per-clip work is hoisted)
        # AdaLN tables of all layers are built in one launch per step ([layers, R, 6, C] fp32, R = distinct timesteps x batch);
        # beyond this many bytes (masks with fractional edges: hundreds of distinct timesteps) ONE [R, 6, C] table is rebuilt
        # per layer instead -- the reference's own footprint is [B, L, 6, C] per step (wan_transformer3d_FlexAM.py:944)
        self.table_limit = int(os.environ.get("FLEXAM_ADALN_TABLE_BYTES", str(1 << 30)))
        self.fp8 = False                            # BASELINE configs[4]: QKV / FFN GEMMs on fp8 MFMA (enable_fp8)
        self._fp8_w = None
        self.fp8_modules = False                    # fp8 GEMMs inside blocks that are called as modules (not the fused path): enable_fp8
        self._pack()

    # ------------------------------------------------------------------ weights
    def _pack(self):
        m, dev, d = self.model, self.device, self.dim
        bf = lambda t: t.detach().to(dev, BF16).contiguous()
        f32 = lambda t: t.detach().to(dev, F32).contiguous()
        small = lambda t: t.detach().to(dev).contiguous() if t.dtype in (BF16, F32) else f32(t)

        def pad_k(w2d):
            n, k = w2d.shape
            out = torch.zeros(n, _round_up(k, 64), device=dev, dtype=BF16)
            out[:, :k] = w2d.detach().to(dev, BF16)
            return out
        self.pe_w, self.pe_b = pad_k(m.patch_embedding.weight.flatten(1)), f32(m.patch_embedding.bias)
        self.ref_w = self.ref_b = None
        if m.ref_conv is not None:
            self.ref_w, self.ref_b = pad_k(m.ref_conv.weight.flatten(1)), f32(m.ref_conv.bias)
        self.head_w, self.head_b = bf(m.head.head.weight), f32(m.head.head.bias)
        self.txt = [(pad_k(m.text_embedding[0].weight), f32(m.text_embedding[0].bias)),
                    (bf(m.text_embedding[2].weight), f32(m.text_embedding[2].bias))]
        self.time = [(small(l.weight), f32(l.bias)) for l in (m.time_embedding[0], m.time_embedding[2], m.time_projection[1])]
        self.dens = [(small(l.weight), f32(l.bias)) for l in (m.density_embedding[0], m.density_embedding[2], m.density_projection[1])]
        # per-layer parameter packs live on the block modules (`_Block.packed()`: fused q|k|v and cross k|v buffers that the
        # parameters alias, softmax_scale * log2(e) folded into norm_q for the FLEXAM_ATTN_PRESCALED attention form); a block that
        # was replaced (`transformer.blocks[i] = wrapper`, comfyui_nodes.py:67-71) or whose forward / attention forward was
        # re-bound (wan_transformer3d_FlexAM.py:807-815) has no pack here and is CALLED as a module by run()
        self.blocks, self.block_modules = [], list(m.blocks)
        for blk in m.blocks:
            blk = getattr(blk, "_orig_mod", blk)          # torch.compile(block) wraps the same module: nothing to compile in a HIP-call block
            native = hasattr(blk, "packed") and hasattr(blk, "pristine") and blk.pristine()
            self.blocks.append(blk.packed() if native else None)
        self.fused = all(p is not None for p in self.blocks)
        nat = [p for p in self.blocks if p is not None]
        self.mod = torch.stack([p["mod"] for p in nat]) if self.fused else None                # [nl, 6, d]
        self.mdens = torch.stack([p["mdens"] for p in nat]) if self.fused else None           # [nl, 2, d]
        self.hmod, self.hmdens = f32(m.head.modulation), f32(m.head.modulation_density)       # [1,2,d], [1,1,d]
        self.cnn = None
        if m.cnn_conv1 is not None:
            cin = [_round_up(m.cnn_conv1[0].weight.shape[1], 64), 192, 192, 128]
            self.cnn = dict(
                convs=[_ConvCL(getattr(m, f"cnn_conv{i}")[0].weight, getattr(m, f"cnn_conv{i}")[0].bias, cin[i - 1], dev) for i in range(1, 5)],
                gn=[(f32(getattr(m, f"cnn_conv{i}")[1].weight), f32(getattr(m, f"cnn_conv{i}")[1].bias)) for i in range(1, 5)],
                conv5=_ConvCL(m.cnn_conv5.weight, m.cnn_conv5.bias, 128, dev), groups=[24, 24, 12, 12], cp=cin)

    def set_parallel(self, sp_group, sp_rank: int, sp_size: int, world_group=None, world_size: int = None, cfg_size: int = 1,
                     cfg_row: int = 0):
        """Parallel layout of this rank: token chunk `sp_rank` of `sp_size` inside `sp_group`; with cfg_size = 2 the
        world is two such groups, one per CFG row (world rank = cfg_row * sp_size + sp_rank)."""
        self.sp_group, self.sp_rank, self.sp_size = sp_group, sp_rank, sp_size
        mode = os.environ.get("FLEXAM_SP_MODE", "allgather")
        if mode not in ("ulysses", "allgather"):
            raise ValueError(f"FLEXAM_SP_MODE={mode!r}: expected 'ulysses' or 'allgather'")
        self.sp_mode = mode if (mode == "allgather" or self.nh % max(sp_size, 1) == 0) else "allgather"
        # FLEXAM_SP_OVERLAP.  K|V all-gather: 0 (default since r6) = ONE gather per block and CFG row, waited for, then ONE ordinary
        # attention call; 1 = head-group pieces with local-chunk-first partial attention + merge underneath them.  r5 measured the
        # overlap machinery at 6.7 ms of a 48 ms rank step at 8 GPUs (three partial calls parking 17 fp32 slots for a merge: 605 us of
        # attention per block against 342 for the one call; profiles/r5o_*): it pays only on links slow enough that hiding ~0.3 ms of
        # a block's gather is worth 0.22 ms of compute, which bench.py's layout probe measures per node -- the default is the form
        # that is fastest on compute.  All-to-all over heads: 1 (default) = a sample's blocks leave under the other sample's projection,
        # 2 = attention per sample as well, 0 = one exchange for the pair.
        ov = os.environ.get("FLEXAM_SP_OVERLAP")
        ov = ("1" if self.sp_mode == "ulysses" else "0") if ov is None else ov.strip().lower()
        self.sp_overlap_level = 0 if ov in ("0", "off", "false", "no", "") else (2 if ov == "2" else 1)      # anything else: on (1)
        self.sp_overlap = self.sp_overlap_level != 0
        # with the overlap on, the K|V gather is cut into `sp_pieces` groups of heads, one collective each: the attention of a group starts
        # when ITS piece has landed, the later pieces travel underneath it.  Default: 2 pieces from 4 chunks on (3+ peers: the gather
        # outlasts the local-chunk attention it hides under), 1 below and whenever the gather is waited for (two attention calls on half
        # the heads each fill 256 CUs worse than one: 44.2 against 41.4 ms per rank step, profiles/r5o_*)
        env = os.environ.get("FLEXAM_SP_PIECES")
        pieces = int(env) if env is not None else (2 if (sp_size >= 4 and self.nh % 2 == 0 and self.sp_overlap and self.sp_mode == "allgather") else 1)
        if self.sp_mode != "allgather":
            pieces = 1                              # (head-group pieces belong to the gather)
        if pieces < 1 or self.nh % pieces:
            raise ValueError(f"FLEXAM_SP_PIECES={pieces}: must divide the {self.nh} heads")
        self.sp_pieces = pieces if sp_size > 1 else 1
        self.sp_fused_qkv = os.environ.get("FLEXAM_SP_FUSED_QKV", "1") != "0"
        # FLEXAM_CU_BUDGET=<n>: plan the persistent grids for n CUs (a multiple of 8) while this layout runs collectives beside compute --
        # a kernel that owns every CU leaves a collective's own kernels nowhere to run until it ends.  Not set by default: on one GPU
        # the emulation's stand-in (a single delay wave) finds room beside the GEMMs, and 8 CUs cost 3 % of a rank's step
        # (profiles/r6s_*); what RCCL's kernels need on a real node is the first thing to measure there.
        if os.environ.get("FLEXAM_CU_BUDGET"):
            hip.set_cu_budget(int(os.environ["FLEXAM_CU_BUDGET"]) if sp_size > 1 else 0)
        self.world_group = world_group if cfg_size > 1 else sp_group
        self.world_size = world_size if world_size is not None else sp_size
        self.cfg_size, self.cfg_row = cfg_size, cfg_row
        self._ws.clear()
        self._ws_gen += 1

    def enable_fp8(self, on: bool = True):
        """QKV (self-attention q|k|v, cross-attention q) and FFN projections on the fp8 (OCP e4m3) MFMA path: weights quantised once per output channel, activations per
        row and per call (flexam_quantize_rows_fp8), fp32 accumulation, the same fused epilogues.  Attention, the output / cross
        projections, norms, modulation and the residual stream are unchanged.  BASELINE.json configs[4] ("fp8 MFMA QKV/FFN
        variant"); the reference's own fp8 mode only stores weights in fp8 (FlexAM/utils/fp8_optimization.py:1-57)."""
        if not self.fused:
            # some block is replaced / wrapped / re-bound: every block is CALLED as a module and carries the switch itself
            # (_Block.set_fp8, set by model.enable_fp8_gemm on every native block, also inside wrappers) -- the same GEMMs on the fp8 pipe
            # with absmax row scales; the engine only has to keep its own fused-path state out of the way
            for blk in self.model.modules():
                if hasattr(blk, "set_fp8"):
                    blk.set_fp8(on)
            self.fp8 = False
            self.fp8_modules = bool(on)
            return
        # wo / cwo are quantised too: FLEXAM_FP8_OPROJ=1 (read per forward; an experiment, not part of configs[4]'s "QKV/FFN") runs the
        # self-attention and cross-attention output projections on the fp8 pipe as well -- the attention output is row-quantised by one
        # more pass (flexam_quantize_rows_fp8)
        if on and self._fp8_w is None:
            self._fp8_w = []
            for p in self.blocks:
                q = {}
                for name in ("wqkv", "cwq", "w1", "w2", "wo", "cwo"):
                    q[name], q["s_" + name] = hip.quantize_rows_fp8(p[name])
                # bounds for the a-priori scale of FFN1's e4m3 output (flexam_ln_modulate_fp8, next_scale): the largest L2 norm of a
                # DEQUANTISED w1 row (what the MFMA multiplies) and the largest |bias|; two floats per layer, read back once
                deq = q["w1"].view(torch.float8_e4m3fn).float() * q["s_w1"][:, None]
                q["w1_norm"] = float(deq.norm(dim=1).max()) * 1.001
                q["b1_max"] = float(p["b1"].abs().max()) if p["b1"] is not None else 0.0
                del deq
                self._fp8_w.append(q)
        self.fp8 = bool(on)
        self._ws.clear()
        self._ws_gen += 1

    def _ln_fp8(self, xres, ws, hbuf, nxt=None, **kw):
        """LN + modulate as the fp8 GEMMs' A operand: one launch at widths the wave-per-row kernel covers (multiples of 512, the 5B
        model's 3072), the bf16 row kernel followed by the row quantiser otherwise.  nxt = (w_norm_max, bias_max) of the GEMM the
        rows feed: the launch then also writes the a-priori output scales of that GEMM into ws["so"] (fused form only; returns
        whether it did)."""
        d, m = self.dim, xres.shape[0]
        if d % 512 == 0 and d <= 4096:
            if nxt is not None:
                hip.ln_modulate_fp8(xres, ws["a8d"][:m], ws["sa"][:m], eps=self.eps, next_scale=ws["so"][:m], next_wnorm=nxt[0], next_bias=nxt[1], **kw)
                return ws["a8d"][:m], ws["sa"][:m], True
            a8, sa = hip.ln_modulate_fp8(xres, ws["a8d"][:m], ws["sa"][:m], eps=self.eps, **kw)
            return a8, sa, False
        hip.ln_modulate(xres, out=hbuf, eps=self.eps, **kw)
        a8, sa = hip.quantize_rows_fp8(hbuf, ws["a8d"][:m], ws["sa"][:m])
        return a8, sa, False

    def set_sequence_parallel(self, group, rank: int, size: int):
        self.set_parallel(group, rank, size)

    # ------------------------------------------------------------------ per-clip state
    def _cnn_block(self, control: torch.Tensor, additional: torch.Tensor) -> torch.Tensor:
        """control [Cc,F,H,W], additional [Ca,F,H,W] -> cnn output [Co,F,H,W] fp32 (FX.py:869-880)."""
        dev = self.device
        _, f, h, w = control.shape
        cn = self.cnn
        rows = f * (h + 2) * (w + 2)
        img = _Image(f, h, w, cn["cp"][0], dev)
        hip.pack_cl(control.contiguous(), img.img, 0)
        hip.pack_cl(additional.contiguous(), img.img, control.shape[0])
        prev = None
        for i, conv in enumerate(cn["convs"]):
            y = hip.gemm(img.mat, conv.weight, conv.bias, a_koff=conv.koff(w + 2, dev), m=rows, k=conv.weight.shape[1], out_dtype=F32)
            cp_out = cn["cp"][i + 1] if i + 1 < 4 else 128
            nxt = _Image(f, h, w, cp_out, dev)
            gamma, beta = cn["gn"][i]
            residual = img.img if i in (1, 3) else None                   # x2 = conv2(x1) + x1, x4 = conv4(x3) + x3
            hip.groupnorm_silu_cl(y, conv.cout, f, h, w, cn["groups"][i], gamma, beta, nxt.img, residual=residual)
            prev, img = img, nxt
        c5 = cn["conv5"]
        y = hip.gemm(img.mat, c5.weight, c5.bias, a_koff=c5.koff(w + 2, dev), m=rows, k=c5.weight.shape[1], out_dtype=F32)
        return hip.unpack_cl(y, c5.cout, f, h, w)

    def set_conditioning(self, context: List[torch.Tensor], y: Optional[torch.Tensor], full_ref: Optional[torch.Tensor],
                         additional_control: Optional[torch.Tensor], density: Optional[torch.Tensor], latent_shape,
                         shared: bool = False):
        """Computes everything that does not depend on the noisy latent or the timestep.
        context: list of B tensors [len_i, text_dim]; y [By, 100, F, H, W]; full_ref [By, 48, H, W];
        additional_control [By, 240, F, H, W]; density [B].  shared=True: conditioning tensors have
        batch 1 and are shared by all B rows (the sampler's CFG pair differs only in `context`)."""
        dev, d = self.device, self.dim
        B = len(context)
        cx, f, h, w = latent_shape
        lvid = f * (h // 2) * (w // 2)
        ref_len = (h // 2) * (w // 2) if (full_ref is not None and self.ref_w is not None) else 0
        L = lvid + ref_len
        nb = 1 if shared else B
        kpe = self.pe_w.shape[1]
        patch_a = torch.zeros(nb, lvid, kpe, device=dev, dtype=BF16)
        ref_tok = torch.empty(nb, ref_len, d, device=dev, dtype=F32) if ref_len else None
        if y is not None:
            for b in range(nb):
                yb = y[b].to(dev)
                if self.cnn is not None and additional_control is not None:
                    cnn_out = self._cnn_block(yb[:cx].float(), additional_control[b].to(dev).float())
                    hip.patchify(cnn_out, patch_a[b], col0=cx * 4)
                    rest = yb[cx:].float().contiguous()
                    hip.patchify(rest, patch_a[b], col0=(cx + cnn_out.shape[0]) * 4)
                else:
                    hip.patchify(yb.float().contiguous(), patch_a[b], col0=cx * 4)
        if ref_len:
            kr = self.ref_w.shape[1]
            for b in range(nb):
                ra = torch.zeros(ref_len, kr, device=dev, dtype=BF16)
                hip.patchify(full_ref[b].to(dev).float().unsqueeze(1).contiguous(), ra)
                # the reference's conv output is bf16 (autocast); keep that rounding for the tokens
                hip.gemm(ra, self.ref_w, self.ref_b, out=ref_tok[b], out_dtype=F32)
        # text -> context embedding -> per-block cross K (normalised) and V
        tdim = self.txt[0][0].shape[1]
        ctx_in = torch.zeros(B * self.text_len, tdim, device=dev, dtype=BF16)
        n_max = 0
        for b, u in enumerate(context):
            n = min(u.shape[0], self.text_len)
            n_max = max(n_max, n)
            ctx_in[b * self.text_len:b * self.text_len + n, :u.shape[1]] = u[:n].to(dev, BF16)
        # Rows n_max .. text_len - 1 of EVERY sample are zero before the text MLP (the reference pads the same way, FX.py:958-964),
        # so behind it they are one and the same context row and, per block, one and the same K / V row: cross-attention keeps the
        # first of them and counts it text_len - n_max times (flexam_attn_fwd_lastkey) instead of attending to 386 copies
        cross_lk, cross_mult = None, 1.0
        if n_max + 1 < self.text_len and os.environ.get("FLEXAM_CROSS_DEDUP", "1") != "0":
            cross_lk, cross_mult = n_max + 1, float(self.text_len - n_max)
        hmid = hip.gemm(ctx_in, self.txt[0][0], self.txt[0][1], epilogue=hip.EPI_GELU_TANH)
        ctx = hip.gemm(hmid, self.txt[1][0], self.txt[1][1])
        cross_kv = []
        for p in (self.blocks if self.fused else ()):
            kv = hip.gemm(ctx, p["cwkv"], p["cbkv"])
            hip.rmsnorm_rope(kv[:, :d], p["cnk"], eps=self.eps)
            cross_kv.append(kv.view(B, self.text_len, 2 * d))
        # density embedding
        dens_emb = dens0 = None
        if density is not None:
            s = hip.sinusoid_embed(density.to(dev, F32).reshape(-1), self.freq_dim)
            e1 = hip.small_linear(s, *self.dens[0])
            dens_emb = hip.small_linear(e1, *self.dens[1], silu_in=True)
            dens0 = hip.small_linear(dens_emb, *self.dens[2], silu_in=True).view(B, 2, d)
        # all samples carry the same density (the sampler's CFG pair does): part of what lets block 0 share its self-attention half
        dens_same = dens0 is None or B == 1 or bool((dens0 == dens0[:1]).all())
        grid = (f + (1 if ref_len else 0), h // 2, w // 2)
        if self._angles is None:
            self._angles = self.model._rope_angles()
        cos, sin = rope_tables(grid, L, self.hd, self._angles)
        self.cond = dict(B=B, nb=nb, L=L, lvid=lvid, ref_len=ref_len, latent_shape=(cx, f, h, w), patch_a=patch_a, ref_tok=ref_tok,
                         cross_kv=cross_kv, dens_emb=dens_emb, dens0=dens0, cos=cos.to(dev), sin=sin.to(dev),
                         ctx=ctx.view(B, self.text_len, d), grid=grid, dens_same=dens_same, cross_lk=cross_lk, cross_mult=cross_mult)
        self.n_conditioning += 1
        return self.cond

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, lc):
        """Activation buffers of one run.  Buffers of other shapes are dropped when a new shape arrives."""
        key = (B, lc)
        if key not in self._ws:
            dev, d, m = self.device, self.dim, B * lc
            self._ws = {}
            self._ws_gen += 1
            self._ws[key] = dict(
                x=torch.empty(m, d, device=dev, dtype=F32), h=torch.empty(m, d, device=dev, dtype=BF16),
                qkv=torch.empty(m, 3 * d, device=dev, dtype=BF16), ao=torch.empty(m, d, device=dev, dtype=BF16),
                ffn=torch.empty(m, self.ffn, device=dev, dtype=BF16),
                head=torch.empty(m, self.head_w.shape[0], device=dev, dtype=F32))
            if self.fp8:
                self._ws[key].update(a8=torch.empty(m, self.ffn, device=dev, dtype=torch.uint8), sa=torch.empty(m, device=dev, dtype=F32),
                                     a8d=torch.empty(m, d, device=dev, dtype=torch.uint8), so=torch.empty(m, device=dev, dtype=F32))
        return self._ws[key]

    def _attn8_buffers(self, nb, lc, heads=None):
        """MXFP8 operand buffers of the quantised self-attention (one set per (samples, tokens, heads) shape, reused by every block)."""
        cache = self.__dict__.setdefault("_attn8", {})
        heads = self.nh if heads is None else heads
        if (nb, lc, heads) not in cache:
            if any(k[1:] != (lc, heads) for k in cache):          # another token / head count: drop the old sets (as _workspace does)
                cache.clear()
            cache[(nb, lc, heads)] = hip.attn_fp8_buffers(nb, heads, lc, self.device)      # (block 0 may run one sample: its own set, not a re-allocation per step)
        return cache[(nb, lc, heads)]

    # ------------------------------------------------------------------ per-step
    def embed_time(self, t_rows: torch.Tensor):
        """t_rows [R] fp32 distinct timesteps -> e [R, d], e0 [R, 6, d] (fp32, FX.py:928-944)."""
        R, d = t_rows.numel(), self.dim
        s = hip.sinusoid_embed(t_rows.to(self.device, F32), self.freq_dim)
        e = torch.empty(R, d, device=self.device, dtype=F32)
        e0 = torch.empty(R, 6 * d, device=self.device, dtype=F32)
        step = 8 if R <= 8 else 32                 # rows per pass over the 113 MB projection weight (results do not depend on it)
        for i in range(0, R, step):
            sl = slice(i, min(i + step, R))
            e1 = hip.small_linear(s[sl], *self.time[0])
            hip.small_linear(e1, *self.time[1], silu_in=True, out=e[sl])
            hip.small_linear(e[sl], *self.time[2], silu_in=True, out=e0[sl])
        return e, e0.view(R, 6, d)

    def run(self, x: torch.Tensor, t_rows: torch.Tensor, row_index: Optional[torch.Tensor], rows_per_batch: int,
            only_row: Optional[int] = None, teacache=None, cond_flag: bool = True, rows_shared: bool = False) -> torch.Tensor:
        """x [Bx, 48, F, H, W] (Bx = B, or 1 when all rows share the latent); t_rows [R] distinct
        timesteps with R = B * rows_per_batch table rows (rows of batch b are b*rows_per_batch ..);
        row_index int32 [B * L] global table row per token, or None (then token (b, l) uses row b).
        Returns the head output tokens fp32 [B, Lc, 4*out_dim] of this rank's token chunk."""
        cd, dev, d = self.cond, self.device, self.dim
        B, L, lvid, ref_len = cd["B"], cd["L"], cd["lvid"], cd["ref_len"]
        # only_row: run a single conditioning row (cfg_skip: the unconditional row is dropped, cfg_optimization.py:5-37);
        # t_rows / row_index then describe that one row
        rsel = slice(None) if only_row is None else slice(only_row, only_row + 1)
        if only_row is not None:
            B = 1
        dens0 = cd["dens0"][rsel] if cd["dens0"] is not None else None
        dens_emb = cd["dens_emb"][rsel] if cd["dens_emb"] is not None else None
        cx, f, h, w = cd["latent_shape"]
        sp, rank = self.sp_size, self.sp_rank
        # A sequence that does not divide over the ranks is padded to the next multiple with zero tokens at its end, as the reference
        # does (FX.py:919-925); they are rows like any other in every token-local op, never keys of self-attention (the key ranges
        # below end at L), and the head gather drops them
        # VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION under the K|V gather (one gather, waited for): the ranks exchange their MXFP8 key / value
        # RECORDS (one per 64 keys) instead of bf16 rows, so every chunk is a whole number of 64-key tiles: the padding unit is 64 x ranks
        sage_asked = os.environ.get("VIDEOX_ATTENTION_TYPE", "FLASH_ATTENTION") == "SAGE_ATTENTION"
        sage_gather = (sage_asked and self.fused and sp > 1 and getattr(self, "sp_mode", None) == "allgather" and not self.sp_overlap
                       and self.sp_pieces == 1)
        unit = sp * 64 if sage_gather else sp
        Lp = -(-L // unit) * unit
        lc = Lp // sp
        tok0 = rank * lc
        if Lp > L and cd["cos"].shape[0] < Lp:             # RoPE rows of the pad tokens: the identity, like every token beyond the grid
            extra = Lp - cd["cos"].shape[0]
            cd["cos"] = torch.cat([cd["cos"], torch.ones(extra, cd["cos"].shape[1], device=dev)])
            cd["sin"] = torch.cat([cd["sin"], torch.zeros(extra, cd["sin"].shape[1], device=dev)])
        ws = self._workspace(B, lc)
        xres, hbuf, qkv, ao, ffn, head = ws["x"], ws["h"], ws["qkv"], ws["ao"], ws["ffn"], ws["head"]
        xr = xres.view(B, lc, d)

        # ---- stem: patch embedding of the noisy latent (+ cached static channels), ref tokens
        bx = x.shape[0]
        full = torch.empty(Lp, d, device=dev, dtype=F32) if sp > 1 else None
        if Lp > L:
            full[L:].zero_()
        for b in range(bx):
            pa = cd["patch_a"][b if cd["nb"] > 1 else 0]
            hip.patchify(x[b].to(dev).contiguous(), pa, col0=0)
            dst = full if sp > 1 else xr[b]
            hip.gemm(pa, self.pe_w, self.pe_b, out=dst[ref_len:L], out_dtype=F32)
            if ref_len:
                dst[:ref_len].copy_(cd["ref_tok"][b if cd["nb"] > 1 else 0])
            if sp > 1:
                xr[b].copy_(full[tok0:tok0 + lc])
        # CFG pair on one latent (PIPE.py:846-848 feeds `torch.cat([latents] * 2)`): until the first cross-attention the two samples
        # are the same tensor -- same tokens, same timestep rows, same density -- so block 0 runs LayerNorm, q|k|v, RoPE, self-
        # attention and the output projection ONCE and the second sample's residual stream is a copy (half of 1/30 of the
        # attention and projection work of a step; every later operation sees the text and runs per sample)
        share0 = (self.fused and B == 2 and bx == 1 and sp == 1 and rows_shared and cd.get("dens_same", False) and teacache is None
                  and os.environ.get("FLEXAM_SHARE_BLOCK0", "1") != "0")
        self.share0_taken = bool(share0)
        if not share0:
            for b in range(bx, B):
                xr[b].copy_(xr[0])

        # ---- timestep embedding on the distinct rows + AdaLN tables of all blocks and the head
        R = t_rows.numel()
        if rows_shared and R > rows_per_batch:     # every sample carries the same timestep rows (the sampler's CFG pair): embed once
            e1, e01 = self.embed_time(t_rows[:rows_per_batch])
            reps = R // rows_per_batch
            e, e0 = e1.repeat(reps, 1), e01.repeat(reps, 1, 1)
        else:
            e, e0 = self.embed_time(t_rows)
        per_layer = (not self.fused) or self.nl * R * 6 * d * 4 > self.table_limit
        dens0c = dens0.contiguous() if dens0 is not None else None
        # the tables live in the workspace (one set per row count): the blocks' launches name their addresses, and a recorded launch plan
        # (below) is only valid while they stay put
        tabs = ws.setdefault(("tabs", R, per_layer), {})
        if not tabs:
            tabs["blk"] = torch.empty(1 if per_layer else self.nl, R, 6, d, device=dev, dtype=F32)
            tabs["head"] = torch.empty(1, R, 2, d, device=dev, dtype=F32)
        if per_layer:
            tab, tab1 = None, tabs["blk"]
        else:
            tab = tabs["blk"]
            hip.mod_table(self.mod, e0, tab, rows_per_batch, 0b010010, self.mdens, dens0c, 0xFF1FF0 if dens0 is not None else -1)
        htab = tabs["head"]
        e2 = e.unsqueeze(1).expand(R, 2, d).contiguous()
        hd_dens = dens_emb.reshape(B, 1, d).contiguous() if dens_emb is not None else None
        hip.mod_table(self.hmod, e2, htab, rows_per_batch, 0b10, self.hmdens if hd_dens is not None else None, hd_dens,
                      0xF0 if hd_dens is not None else -1)
        calc = True
        if teacache is not None:
            calc = self._teacache_decide(teacache, e0, row_index, B, L, cond_flag)
        if row_index is not None:
            # the per-token row index of THIS rank's rows, copied (47 KB) into a buffer of the workspace: the blocks' launches then see ONE
            # address from step to step whoever built the index (the sampler keeps one tensor per clip, the reference-style forward()
            # builds a new one per call) -- what a recorded launch plan (below) needs
            if sp > 1:
                from .dist import shard_rows
                row_index = shard_rows(row_index, B, L, rank, sp, chunk=lc)
            buf = ws.get("row_index_buf")
            if buf is None or buf.numel() != row_index.numel():
                buf = ws["row_index_buf"] = torch.empty(row_index.numel(), device=dev, dtype=I32)
            buf.copy_(row_index.reshape(-1))
            row_index = buf
        rpb = lc                                   # used only when row_index is None: row = m // lc = b

        nh, hdim = self.nh, self.hd
        # the reference reads the switch at every attention call (attention_utils.py:195); quantised self-attention on one rank only
        # one rank: the fused producer (RMSNorm + RoPE write the MXFP8 operands).  Sequence parallel with the all-to-all over heads: every
        # rank ends up with ALL tokens of its heads in bf16, packs them and runs the MXFP8 kernel on them.  K|V all-gather in its default
        # form (one gather, waited for): each rank quantises ITS keys / values and the MXFP8 records are what is gathered (sage_gather,
        # above).  The overlapped gather forms (head-group pieces, partial softmaxes) keep the bf16 kernel -- said once per process
        sage = sage_asked and self.fused and (sp == 1 or (self.sp_mode == "ulysses" and Lp == L) or sage_gather)
        self.sage_taken = bool(sage)                       # what this forward DID (bench.py labels its line from it, like share0_taken)
        if sage_asked and not sage and not DiTEngine._sage_warned:
            DiTEngine._sage_warned = True
            import warnings
            warnings.warn("flexam_amd: VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION is ignored " +
                          "under sequence parallelism with the OVERLAPPED K|V all-gather (FLEXAM_SP_OVERLAP=1 / head-group pieces) or an all-to-all "
                          "over a padded sequence: self-attention runs the bf16 kernel", RuntimeWarning, stacklevel=2)
        fp8_oproj = self.fp8 and os.environ.get("FLEXAM_FP8_OPROJ", "0") == "1"
        q4 = qkv.view(B, lc, 3 * d)[:, :, 0:d].unflatten(2, (nh, hdim))
        k4 = qkv.view(B, lc, 3 * d)[:, :, d:2 * d].unflatten(2, (nh, hdim))
        v4 = qkv.view(B, lc, 3 * d)[:, :, 2 * d:].unflatten(2, (nh, hdim))
        ao4 = ao.view(B, lc, nh, hdim)
        if teacache is not None:
            key = "previous_residual_cond" if cond_flag else "previous_residual_uncond"
            if not calc:                                   # skipped step: x += residual of the last computed step (FX.py:1003-1006)
                res = getattr(teacache, key)
                n = xres.shape[0]                          # `previous_residual[-x.size(0):]`: a cfg-skipped (B = 1) forward takes the
                res = res[-n:] if res.shape[0] >= n else res.repeat(n // res.shape[0], 1)      # conditional row of a B = 2 residual
                hip.axpby(xres, 1.0, res.contiguous(), 1.0)
            else:
                ori = xres.clone()
        if calc and not self.fused:
            self._run_block_modules(xres, B, lc, e0, row_index, rows_per_batch, dens0, t_rows, rsel)

        def run_blocks():
            for i, p in enumerate(self.blocks if (calc and self.fused) else ()):
                if per_layer:                                  # one table, rebuilt per layer (bounded memory, see table_limit)
                    hip.mod_table(self.mod[i:i + 1], e0, tab1, rows_per_batch, 0b010010, self.mdens[i:i + 1], dens0c,
                                  0xFF1FF0 if dens0 is not None else -1)
                    T = tab1[0]
                else:
                    T = tab[i]
                fp8_here = self.fp8
                nb = 1 if (share0 and i == 0) else B               # samples that run the self-attention half of this block (share0: above)
                mb = nb * lc
                ri = row_index[:mb] if row_index is not None else None
                if fp8_here:                                   # LN + modulate written as e4m3 + row scales: the fp8 QKV GEMM's A operand
                    a8, sa, _ = self._ln_fp8(xres[:mb], ws, hbuf[:mb], shift=T[:, 0], scale=T[:, 1], row_index=ri, rows_per_batch=rpb)
                else:
                    hip.ln_modulate(xres[:mb], out=hbuf[:mb], eps=self.eps, shift=T[:, 0], scale=T[:, 1], row_index=ri, rows_per_batch=rpb)
                if sp > 1 and self.sp_mode == "ulysses":
                    # all tokens of H/sp heads per rank: q|k|v all-to-all -> attention -> all-to-all back; the o-projection reads the
                    # returned blocks in place (flexam_amd/dist.py)
                    a_o, koff_o = self._ulysses_attention(qkv, hbuf, fp8_here and (a8, sa), i, p, B, lc, tok0, sage=sage)
                    hip.gemm_gate_residual(a_o, p["wo"], p["bo"], xres, gate=T[:, 2], gate_row=row_index, rows_per_batch=rpb, a_koff=koff_o)
                elif sp > 1:
                    # K|V projection + K norm/RoPE first, written straight into the send buffer; their all-gather (RCCL over xGMI)
                    # runs under the Q projection, the Q norm/RoPE and the attention to the LOCAL chunk
                    # (FLEXAM_SP_FUSED_QKV=1, default: ONE q|k|v launch instead -- at a rank's few thousand rows two launches of 24 and 12 tile
                    #  columns quantise worse on 256 CUs than one of 36 (emulated rank of 8, profiles/r5*: 119 + 80 us against ~135), and the
                    #  gather starts ~15 us later, not ~80)
                    fused_qkv = self.sp_fused_qkv
                    if sage:                               # (sage_gather: the MXFP8 records travel; always one q|k|v launch)
                        self._proj(hbuf, fp8_here and (a8, sa), i, p, "wqkv", "bqkv", slice(None), qkv)
                        self._allgather_attention_mx(qkv, p, ao4, q4, k4, v4, B, lc, tok0)
                    else:
                        self._proj(hbuf, fp8_here and (a8, sa), i, p, "wqkv", "bqkv", slice(None) if fused_qkv else slice(d, None), qkv if fused_qkv else qkv[:, d:])
                        self._allgather_attention(qkv, hbuf, fp8_here and (a8, sa), i, p, ao4, q4, B, lc, tok0, q_done=fused_qkv)
                    hip.gemm_gate_residual(ao, p["wo"], p["bo"], xres, gate=T[:, 2], gate_row=row_index, rows_per_batch=rpb)
                else:
                    a8sa = fp8_here and (a8[:mb], sa[:mb])
                    self._proj(hbuf[:mb], a8sa, i, p, "wqkv", "bqkv", slice(None), qkv[:mb])
                    if sage and nh == 24 and hdim == 128 and os.environ.get("FLEXAM_SAGE_FUSED", "1") != "0":   # SAGE_ATTENTION: MXFP8 operands (csrc/attn_fp8.inc);
                        bufs = self._attn8_buffers(nb, lc)     # RMSNorm + RoPE write Q and K as operands directly, V is packed on its own
                        hip.rmsnorm_rope_mx(qkv[:mb, 0:d], p["nq"], qkv[:mb, d:2 * d], p["nk"], bufs, cd["cos"], cd["sin"], lc, tok0, eps=self.eps)
                        hip.attn_fp8_pack(None, None, v4[:nb], bufs)
                        hip.attn_fwd_fp8(bufs, lc, out=ao4[:nb])
                    else:
                        hip.rmsnorm_rope(qkv[:mb, 0:d], p["nq"], qkv[:mb, d:2 * d], p["nk"], eps=self.eps, rope_cos=cd["cos"], rope_sin=cd["sin"],
                                         tokens_per_batch=lc, token_offset=tok0, head_dim=hdim)
                        if sage:
                            bufs = self._attn8_buffers(nb, lc)
                            hip.attn_fp8_pack(q4[:nb], k4[:nb], v4[:nb], bufs)
                            hip.attn_fwd_fp8(bufs, lc, out=ao4[:nb])
                        else:
                            hip.attn_fwd(q4[:nb], k4[:nb], v4[:nb], out=ao4[:nb], prescaled=True)
                    if fp8_here and fp8_oproj:
                        a8o, sao = hip.quantize_rows_fp8(ao[:mb], ws["a8d"][:mb], ws["sa"][:mb])
                        hip.gemm_fp8_gate_residual(a8o, sao, self._fp8_w[i]["wo"], self._fp8_w[i]["s_wo"], p["bo"], xres[:mb], gate=T[:, 2], gate_row=ri,
                                                   rows_per_batch=rpb)
                    else:
                        hip.gemm_gate_residual(ao[:mb], p["wo"], p["bo"], xres[:mb], gate=T[:, 2], gate_row=ri, rows_per_batch=rpb)
                    if nb < B:
                        hip.host_op(lambda: xr[1].copy_(xr[0]))         # (a torch copy, not a library call: a host step of a recorded plan)
                # cross-attention on the text context (K/V precomputed per clip)
                qc = qkv[:, 0:d]
                if self.fp8:                                   # the Q projection of cross-attention on the fp8 pipe as well (its K|V are per clip)
                    a8, sa, _ = self._ln_fp8(xres, ws, hbuf, ln_w=p["n3w"], ln_b=p["n3b"])
                    hip.gemm_fp8(a8, sa, self._fp8_w[i]["cwq"], self._fp8_w[i]["s_cwq"], p["cbq"], out=qc)
                else:
                    hip.ln_modulate(xres, out=hbuf, eps=self.eps, ln_w=p["n3w"], ln_b=p["n3b"])
                    hip.gemm(hbuf, p["cwq"], p["cbq"], out=qc)
                hip.rmsnorm_rope(qc, p["cnq"], eps=self.eps)
                kv = cd["cross_kv"][i][rsel]
                if cd.get("cross_lk"):                          # the identical padded text rows as ONE weighted key
                    lk = cd["cross_lk"]
                    hip.attn_fwd_lastkey(q4, kv[:, :lk, 0:d].unflatten(2, (nh, hdim)), kv[:, :lk, d:].unflatten(2, (nh, hdim)), cd["cross_mult"],
                                         out=ao4, prescaled=True)
                else:
                    hip.attn_fwd(q4, kv[:, :, 0:d].unflatten(2, (nh, hdim)), kv[:, :, d:].unflatten(2, (nh, hdim)), out=ao4, prescaled=True)
                if self.fp8 and fp8_oproj:
                    a8o, sao = hip.quantize_rows_fp8(ao, ws["a8d"], ws["sa"])
                    hip.gemm_fp8_gate_residual(a8o, sao, self._fp8_w[i]["cwo"], self._fp8_w[i]["s_cwo"], p["cbo"], xres)
                else:
                    hip.gemm_gate_residual(ao, p["cwo"], p["cbo"], xres)
                # FFN
                if self.fp8:
                    w8 = self._fp8_w[i]
                    # FFN1 writes FFN2's e4m3 operand itself: its output row scales are known before it runs (a bound from the row's L2
                    # norm, written by the LN launch), so there is no absmax / quantise pass over the [M, 14336] intermediate
                    # (FLEXAM_FP8_FFN_APRIORI=0: the earlier form -- bf16 intermediate + an absmax row quantiser pass -- for checkpoints whose w1
                    #  has a few very large rows: the bound is set by the LARGEST row norm, so every ordinary row's outputs then sit lower in
                    #  e4m3's range.  One scale per output row has to cover all 14336 columns, so a per-tile bound cannot be used by FFN2.)
                    apriori = os.environ.get("FLEXAM_FP8_FFN_APRIORI", "1") != "0"
                    a8, sa, bound = self._ln_fp8(xres, ws, hbuf, nxt=(w8["w1_norm"], w8["b1_max"]) if apriori else None, shift=T[:, 3], scale=T[:, 4],
                                                 row_index=row_index, rows_per_batch=rpb)
                    if bound:
                        hip.gemm_fp8_gelu_q(a8, sa, w8["w1"], w8["s_w1"], p["b1"], ws["so"], ws["a8"])
                        a8, sa = ws["a8"], ws["so"]
                    else:                                      # widths the fused LN launch does not cover: bf16 intermediate + row quantiser
                        hip.gemm_fp8(a8, sa, w8["w1"], w8["s_w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH)
                        a8, sa = hip.quantize_rows_fp8(ffn, ws["a8"], ws["sa"])
                    hip.gemm_fp8_gate_residual(a8, sa, w8["w2"], w8["s_w2"], p["b2"], xres, gate=T[:, 5], gate_row=row_index, rows_per_batch=rpb)
                else:
                    hip.ln_modulate(xres, out=hbuf, eps=self.eps, shift=T[:, 3], scale=T[:, 4], row_index=row_index, rows_per_batch=rpb)
                    # (FFN1 -> FFN2 per row chunk, so that the [M, 14336] intermediate stays in the Infinity Cache: the clock rises with the
                    #  saved HBM traffic, but tile quantisation and launch ramps cost more: +0.5 / +1.6 / +7.1 % of a step at 2 / 4 / 7 chunks,
                    #  profiles/r4p_ffn_row_chunks.txt)
                    hip.gemm(hbuf, p["w1"], p["b1"], out=ffn, epilogue=hip.EPI_GELU_TANH)
                    hip.gemm_gate_residual(ffn, p["w2"], p["b2"], xres, gate=T[:, 5], gate_row=row_index, rows_per_batch=rpb)
        def run_head():
            H = htab[0]
            hip.ln_modulate(xres, out=hbuf, eps=self.eps, shift=H[:, 0], scale=H[:, 1], row_index=row_index, rows_per_batch=rpb)
            hip.gemm(hbuf, self.head_w, self.head_b, out=head)

        # ---- launch plan: the blocks and the head issue the same ~420 launches on the same buffers every step (only buffer CONTENTS
        # change), so the first step records them (hip.record: executed and appended to command lists) and every later step re-issues
        # the lists from C (flexam_replay, csrc/replay.hip: ~1 us per launch instead of 20-30 us of Python + ctypes).  Collectives,
        # waits and torch copies between them are host steps of the plan (hip.host_op).  The key names everything the recorded launches
        # depend on besides buffer contents; plans live in the per-clip state (new conditioning = new plans) and hold references to every
        # tensor whose address they carry.  Not with TeaCache (data-dependent skipping), per-layer tables or blocks called as modules.
        use_plan = (calc and self.fused and teacache is None and not per_layer and os.environ.get("FLEXAM_REPLAY", "1") != "0")
        self.replay_taken = False
        if use_plan:
            pkey = (self._ws_gen, B, lc, only_row, bool(share0), bool(sage), self.fp8, fp8_oproj, R, rows_per_batch,
                    row_index.data_ptr() if row_index is not None else 0, tabs["blk"].data_ptr(), cd["cos"].data_ptr(),
                    torch.cuda.current_stream().cuda_stream, hip.num_cus(), os.environ.get("FLEXAM_SAGE_FUSED", "1"), os.environ.get("FLEXAM_FP8_FFN_APRIORI", "1"),
                    sp, rank, getattr(self, "sp_mode", None), getattr(self, "sp_pieces", 1), getattr(self, "sp_overlap_level", 0),
                    getattr(self, "sp_fused_qkv", True), id(self.sp_group))
            plans = cd.setdefault("_plans", {})
            plan = plans.get(pkey)
            if plan is None:
                for k in [k for k in plans if k[0] != self._ws_gen]:      # plans of dropped activation buffers would keep those buffers alive
                    del plans[k]
                with hip.record() as plan:
                    run_blocks()
                    run_head()
                while len(plans) >= 6:                     # (cond / uncond rows of cfg_skip, the shared-block-0 form, ...: a handful per clip)
                    plans.pop(next(iter(plans)))
                plans[pkey] = plan
            else:
                plan.run()
                self.replay_taken = True
            self.plan_launches = plan.launches
        else:
            run_blocks()
        if teacache is not None and calc:                  # residual = x_after_blocks - x_before (FX.py:1048-1051), kept on the GPU
            hip.axpby(ori, 1.0, xres, -1.0)
            setattr(teacache, key, ori)
        if not use_plan:
            run_head()
        return head.view(B, lc, -1)

    # ------------------------------------------------------------------ block-level seam
    def _run_block_modules(self, xres, B, lc, e0, row_index, rows_per_batch, dens0, t_rows, rsel):
        """Some block is not a pristine native block (replaced, wrapped, or with a re-bound forward): every block is CALLED
        with the reference's block signature (wan_transformer3d_FlexAM.py:1053-1089) on [B, L, C] fp32 tensors.  The AdaLN
        input is materialised per token like the reference's e0 ([B, L, 6, C]); native blocks find its compact form in the
        `_flexam_rows` attribute."""
        cd, d = self.cond, self.dim
        if row_index is not None:
            e_full = e0[row_index.long()].view(B, lc, 6, d)
        else:
            e_full = e0.view(B, 6, d)
        e_full._flexam_rows = (e0, row_index, rows_per_batch)
        grid_sizes = torch.tensor([list(cd["grid"])] * B, dtype=torch.long)
        # Under sequence parallelism the blocks get this rank's token chunk and the GLOBAL lengths / grid, as the reference hands them
        # over (wan_transformer3d_FlexAM.py:970-975, 1075-1086); the exchange around self-attention belongs to `self_attn.forward`
        # (the reference re-binds it to a USP forward, :807-815): the native _SelfAttn.forward reads the context below and gathers K|V
        # itself, a caller's re-bound forward finds group / rank / token offset / RoPE tables in flexam_amd.dist.current_sp_context()
        from .dist import sequence_parallel_context
        sp = self.sp_size
        seq_lens = torch.tensor([cd["L"]] * B, dtype=torch.long)      # the REAL lengths (FX.py:917): under sequence parallelism lc * sp may be padded
        x3 = xres.view(B, lc, d)
        ctx = cd["ctx"][rsel]
        with sequence_parallel_context(self.sp_group, self.sp_rank, sp, self.sp_rank * lc, cd["L"], cd["cos"], cd["sin"]) if sp > 1 else nullcontext():
            for blk in self.block_modules:
                out = blk(x3, e=e_full, density_emb=dens0, seq_lens=seq_lens, grid_sizes=grid_sizes, freqs=self.model.freqs, context=ctx,
                          context_lens=None, dtype=BF16, t=t_rows)
                x3.copy_(out.view(B, lc, d))

    # ------------------------------------------------------------------ TeaCache
    @staticmethod
    def _teacache_decide(tc, e0, row_index, B, L, cond_flag: bool) -> bool:
        """Step-skipping decision of wan_transformer3d_FlexAM.py:977-1000 (host logic on the tiny AdaLN input).
        e0 [R, 6, C]; the reference looks at the LAST token's row (`e0[:, -1, :]`) or at e0 itself for 1-D t."""
        if not cond_flag:
            return tc.should_calc
        if row_index is not None:
            mod_inp = e0[row_index.view(B, L)[:, -1].long()]
        else:
            mod_inp = e0
        if tc.cnt < tc.num_skip_start_steps:
            calc = True
            tc.accumulated_rel_l1_distance = 0
        else:
            prev = tc.previous_modulated_input
            rel = ((mod_inp - prev).abs().mean() / prev.abs().mean()).item()
            tc.accumulated_rel_l1_distance += tc.rescale_func(rel)
            if tc.accumulated_rel_l1_distance < tc.rel_l1_thresh:
                calc = False
            else:
                calc = True
                tc.accumulated_rel_l1_distance = 0
        tc.previous_modulated_input = mod_inp.clone()
        tc.should_calc = calc
        return calc

    # ------------------------------------------------------------------ sequence parallel
    def _ulysses_attention(self, qkv, hbuf, a8sa, layer, p, B, lc, tok0, sage=False):
        """h [B*lc, C] (LayerNorm output of this rank's tokens) -> q|k|v projection -> exchange -> attention -> exchange back ->
        (A base view, per-K-block A offsets) of the attention output for the o-projection.  Head group j = heads j*H/sp .. goes
        to rank j.
        Send layout [B, sp, lc, 3*G] (G = H/sp * head_dim): written by the RMSNorm+RoPE launch itself (q, k normed + rotated, v
        copied), block (b, j) goes to rank j.  Receive layout [B, sp, lc, 3*G] = [B, L, 3*G]: rank-major blocks ARE the token
        order, so attention addresses it with plain strides.  Its output [B, L, G] is cut into the sp token chunks that go back;
        rank j's block returns to [j, B, lc, G], which the o-projection reads as A[m, j*G + c] through its K-block offsets.
        With several samples per rank (the CFG pair batched: pure N-way chunks) the samples are stages of the outbound exchange
        (FLEXAM_SP_OVERLAP=1, default): sample b's q|k|v leave as soon as ITS projection and norm are done and travel under the
        projection of sample b + 1; ONE attention call for the pair follows the last arrival (two calls of half the work units fill
        256 CUs a quarter worse than one), then the outputs return.  FLEXAM_SP_OVERLAP=2 pipelines the attention too: sample b's
        call runs while sample b + 1's blocks arrive and sample b - 1's output returns -- only the last return is not under compute;
        it pays when a link is slower than the ~0.1 ms the two smaller attention calls cost (about 35 GB/s at 8 GPUs)."""
        from .dist import all_to_all_blocks
        sp, nh, hd, d, dev = self.sp_size, self.nh, self.hd, self.dim, self.device
        hg = nh // sp
        G = hg * hd
        W = 3 * G
        ws = self._ws[(B, lc)]
        if "a2a_send" not in ws:
            ws["a2a_send"] = torch.empty(B, sp, lc, W, device=dev, dtype=BF16)
            ws["a2a_recv"] = torch.empty(B, sp, lc, W, device=dev, dtype=BF16)
            ws["a2a_out"] = torch.empty(B, sp * lc, hg, hd, device=dev, dtype=BF16)
            ws["a2a_recv2"] = torch.empty(sp, B, lc, G, device=dev, dtype=BF16)
            ws["a2a_koff"] = torch.tensor([(kb * 64 // G) * (B * lc * G) + (kb * 64) % G for kb in range(d // 64)], dtype=I64, device=dev)
        cd = self.cond
        send, recv, out, recv2 = ws["a2a_send"], ws["a2a_recv"], ws["a2a_out"], ws["a2a_recv2"]
        full = recv.view(B, sp * lc, 3, hg, hd)
        chunks = out.view(B, sp, lc, G)
        nk = cd["L"]                                   # keys: the real tokens (rows nk .. sp*lc - 1 are the reference's zero pads, FX.py:919-925)

        def attend(b0, nb):
            """Attention of samples b0 .. b0 + nb - 1 on this rank's heads over all tokens; SAGE_ATTENTION: the received bf16 q|k|v are
            packed into MXFP8 operands first (flexam_attn_fp8_pack) and the quantised kernel runs (run() only asks for it when nk = sp * lc)."""
            q_, k_, v_ = full[b0:b0 + nb, :, 0], full[b0:b0 + nb, :nk, 1], full[b0:b0 + nb, :nk, 2]
            if sage:
                bufs = self._attn8_buffers(nb, sp * lc, hg)
                hip.attn_fp8_pack(q_, k_, v_, bufs)
                hip.attn_fwd_fp8(bufs, sp * lc, out=out[b0:b0 + nb])
            else:
                hip.attn_fwd(q_, k_, v_, out=out[b0:b0 + nb], prescaled=True)

        def project_and_pack(rows, b0, nb):          # samples b0 .. b0 + nb - 1: rows of h -> q|k|v -> normed / rotated send blocks
            a8 = a8sa and (a8sa[0][rows], a8sa[1][rows])
            self._proj(hbuf[rows], a8, layer, p, "wqkv", "bqkv", slice(None), qkv[rows])
            flat = send[b0:b0 + nb].view(-1)
            hip.rmsnorm_rope_scatter(qkv[rows, 0:d], p["nq"], qkv[rows, d:2 * d], p["nk"], qkv[rows, 2 * d:], flat, flat[G:], flat[2 * G:],
                                     ld_out=W, out_bs=sp * lc * W, col_block=G, block_stride=lc * W, eps=self.eps, rope_cos=cd["cos"],
                                     rope_sin=cd["sin"], tokens_per_batch=lc, token_offset=tok0, head_dim=hd)

        # the exchanges and their waits are host steps (hip.host_op): run here, and again at this place by every replay of a recorded
        # launch plan; `st` carries the Work handles from the step that issues to the step that waits
        st = {"there": [None] * B, "back": [None] * B}

        if "a2a_lists" not in ws:                      # the per-peer block views, made once (a replayed step must not rebuild 32 views per block)
            ws["a2a_lists"] = [([recv[b, i] for i in range(sp)], [send[b, j] for j in range(sp)],
                                [recv2[j, b] for j in range(sp)], [chunks[b, i] for i in range(sp)]) for b in range(B)]
        lists = ws["a2a_lists"]

        def go_there(b, async_op):                     # packed = the same blocks as ONE tensor pair (a backend that copies can do it in one go)
            st["there"][b] = all_to_all_blocks(lists[b][0], lists[b][1], self.sp_group, async_op=async_op, packed=(recv[b], send[b]))

        def go_back(b, async_op):
            st["back"][b] = all_to_all_blocks(lists[b][2], lists[b][3], self.sp_group, async_op=async_op, packed=(recv2[:, b], chunks[b]))

        def wait(which, bs):
            for b in bs:
                if st[which][b] is not None:
                    st[which][b].wait()
        if B == 1 or not self.sp_overlap:
            project_and_pack(slice(None), 0, B)
            hip.host_op(lambda: [go_there(b, False) for b in range(B)])
            attend(0, B)
            hip.host_op(lambda: [go_back(b, False) for b in range(B)])
            return recv2.view(sp * B * lc, G), ws["a2a_koff"]
        for b in range(B):
            project_and_pack(slice(b * lc, (b + 1) * lc), b, 1)
            hip.host_op(lambda b=b: go_there(b, True))
        if self.sp_overlap_level < 2:
            hip.host_op(lambda: wait("there", range(B)))
            attend(0, B)
            hip.host_op(lambda: ([go_back(b, True) for b in range(B)], wait("back", range(B))))
            return recv2.view(sp * B * lc, G), ws["a2a_koff"]
        for b in range(B):
            hip.host_op(lambda b=b: wait("there", [b]))
            attend(b, 1)
            hip.host_op(lambda b=b: go_back(b, True))
        hip.host_op(lambda: wait("back", range(B)))
        return recv2.view(sp * B * lc, G), ws["a2a_koff"]

    @staticmethod
    def _splits_for(units: int, tiles: int, n_cu: int = 256) -> int:
        """Key ranges per work unit for a partial-attention call: fill whole rounds of the CUs, every range >= 8 key tiles."""
        best, best_cost = 1, float(-(-units // n_cu))
        for s in range(2, 9):
            if tiles // s < 8:
                break
            cost = -(-units * s // n_cu) / s + 0.04
            if cost < best_cost * 0.97:
                best, best_cost = s, cost
        return best

    def _proj(self, hbuf, a8sa, layer, p, wname, bname, rows, out):
        """out = h @ W[rows]^T + b[rows]: bf16 MFMA, or fp8 MFMA on the row-quantised h (`a8sa` = (bytes, row scales)) with the
        per-channel scales of the same weight rows."""
        if a8sa:
            w8 = self._fp8_w[layer]
            return hip.gemm_fp8(a8sa[0], a8sa[1], w8[wname][rows], w8["s_" + wname][rows], p[bname][rows], out=out)
        return hip.gemm(hbuf, p[wname][rows], p[bname][rows], out=out)

    def _allgather_attention(self, qkv, hbuf, a8sa, layer, p, ao4, q4, B, lc, tok0, q_done=False):
        """K|V of this rank's tokens are in qkv[:, C:] (projected, not yet normed).  The RMSNorm+RoPE launch writes K (normed,
        rotated) and V into the send buffer, cut into `sp_pieces` groups of heads: [G, B, lc, 2*C/G].  One all-gather per group
        and CFG row assembles [G, B, L, 2*C/G] in token order (the rank-major concatenation IS the token order: no re-layout
        pass), all of them issued at once.  DEFAULT (r6: FLEXAM_SP_OVERLAP=0, one piece): the gather is waited for and ONE ordinary
        attention call of the local queries over all L real keys follows -- the fastest form on compute.  FLEXAM_SP_OVERLAP=1
        (r2-r5's default, a layout-probe candidate): underneath the gather: Q projection, Q norm/RoPE, then the heads of group 0 attend to the LOCAL
        chunk (partial softmax, straight from the send buffer), to the chunks before / after it once piece 0 has landed, one
        merge; the heads of group g > 0 run one ordinary attention call on their gathered piece, which travelled while group
        g - 1 computed.  A peer chunk cannot arrive faster than its one xGMI link delivers it, and the chunks of one gather all
        land together; cutting along the heads gives pieces that are complete work for part of the kernel, so all links stay
        busy in every phase (reference call sites of the missing exchange: wan_transformer3d_FlexAM.py:801-815, 970-975)."""
        from .dist import all_gather_into_tensor, group_backend
        sp, nh, hd, d, dev = self.sp_size, self.nh, self.hd, self.dim, self.device
        ws = self._ws[(B, lc)]
        L = sp * lc                                    # rows of the gathered buffer (the padded sequence)
        Lr = self.cond["L"]                            # keys: the real tokens; rows Lr .. L - 1 are zero pads (FX.py:919-925) and end every key range
        n_loc = max(0, min(lc, Lr - tok0))             # real tokens of the local chunk
        G = self.sp_pieces
        cb, hg = d // G, nh // G
        if "kv_send" not in ws:
            ws["kv_send"] = torch.empty(G, B, lc, 2 * cb, device=dev, dtype=BF16)
            ws["kv_cat"] = torch.empty(G, B, L, 2 * cb, device=dev, dtype=BF16)
            units = B * hg * ((lc + 255) // 256)
            ranges = [n_loc, min(tok0, Lr), max(0, Lr - tok0 - lc)]                      # local, before, after
            ws["kv_splits"] = [self._splits_for(units, (n + 63) // 64) if n else 0 for n in ranges]
            ws["kv_part"] = hip.attn_partial_workspace(B, hg, lc, sum(hip.attn_effective_splits(n, s) for n, s in zip(ranges, ws["kv_splits"]) if n), dev)
        cd = self.cond
        send, cat = ws["kv_send"], ws["kv_cat"]
        flat = send.view(-1)
        hip.rmsnorm_rope_scatter(None, None, qkv[:, d:2 * d], p["nk"], qkv[:, 2 * d:], None, flat, flat[cb:], ld_out=2 * cb, out_bs=lc * 2 * cb,
                                 col_block=cb, block_stride=B * lc * 2 * cb, eps=self.eps, rope_cos=cd["cos"], rope_sin=cd["sin"],
                                 tokens_per_batch=lc, token_offset=tok0, head_dim=hd)
        # RCCL runs the pieces one after the other on its own stream, in the order they are issued here.  The host-staged backends of the
        # test runs (gloo) execute several in-flight collectives of one group on concurrent worker threads, which is not what is being
        # modelled (and delivered wrong chunks intermittently with 8 ranks on one device): there each gather completes before the next.
        overlapped = group_backend(self.sp_group) in ("nccl", "loopback")
        # collectives and waits are host steps (hip.host_op): run here, and again at this place by every replay of a recorded launch plan
        st = {}

        def issue():
            st["works"] = [[all_gather_into_tensor(cat[g, b], send[g, b], group=self.sp_group, async_op=overlapped) for b in range(B)] for g in range(G)]

        def wait(g):
            for w in st["works"][g]:
                if w is not None:
                    w.wait()
        hip.host_op(issue)
        if not q_done:
            self._proj(hbuf, a8sa, layer, p, "wqkv", "bqkv", slice(0, d), qkv[:, 0:d])
        hip.rmsnorm_rope(qkv[:, 0:d], p["nq"], eps=self.eps, rope_cos=cd["cos"], rope_sin=cd["sin"], tokens_per_batch=lc, token_offset=tok0,
                         head_dim=hd)
        heads = lambda t: t.unflatten(2, (hg, hd))
        for g in range(G):
            qg, og = q4[:, :, g * hg:(g + 1) * hg], ao4[:, :, g * hg:(g + 1) * hg]
            kc, vc = cat[g, :, :, 0:cb], cat[g, :, :, cb:]
            if g > 0 or not self.sp_overlap:
                hip.host_op(lambda g=g: wait(g))
                hip.attn_fwd(qg, heads(kc[:, :Lr]), heads(vc[:, :Lr]), out=og, prescaled=True)
                continue
            s_loc, s_before, s_after = ws["kv_splits"]
            n = 0
            if n_loc > 0:
                n = hip.attn_fwd_partial(qg, heads(send[0, :, :n_loc, 0:cb]), heads(send[0, :, :n_loc, cb:]), ws["kv_part"], 0, s_loc, prescaled=True)
            hip.host_op(lambda: wait(0))
            if tok0 > 0:
                n += hip.attn_fwd_partial(qg, heads(kc[:, :min(tok0, Lr)]), heads(vc[:, :min(tok0, Lr)]), ws["kv_part"], n, s_before, prescaled=True)
            if tok0 + lc < Lr:
                n += hip.attn_fwd_partial(qg, heads(kc[:, tok0 + lc:Lr]), heads(vc[:, tok0 + lc:Lr]), ws["kv_part"], n, s_after, prescaled=True)
            hip.attn_merge(og, ws["kv_part"], n, prescaled=True)

    def _allgather_attention_mx(self, qkv, p, ao4, q4, k4, v4, B, lc, tok0):
        """SAGE_ATTENTION under the K|V all-gather (r6; the reference's `sageattn` switch, attention_utils.py:195-203, with the exchange of
        the missing FlexAM/dist, wan_transformer3d_FlexAM.py:801-815): this rank's q, k (RMSNorm + RoPE at the chunk's global offset) and
        v become MXFP8 operands -- written by the RMSNorm + RoPE launch itself at the 5B width (flexam_rmsnorm_rope_mx; the V half by the
        pack kernel), else normed in bf16 and packed --, ONE all-gather moves the key / value RECORDS ([B, H, lc / 64] x 18 KiB per rank:
        288 bytes per key and head instead of 512 in bf16; the rank-major result is the chunk layout flexam_attn_fwd_fp8_chunked reads),
        and one attention call of the local queries over all L real keys follows.  lc is a multiple of 64 (run() pads to 64 x ranks)."""
        from .dist import all_gather_into_tensor
        sp, nh, hd, d = self.sp_size, self.nh, self.hd, self.dim
        cd = self.cond
        ws = self._ws[(B, lc)]
        bufs = self._attn8_buffers(B, lc)
        if "kv8_all" not in ws:
            ws["kv8_all"] = torch.empty(sp, *bufs[2].shape, device=self.device, dtype=torch.uint8)
        kv8_all = ws["kv8_all"]
        if nh == 24 and hd == 128 and os.environ.get("FLEXAM_SAGE_FUSED", "1") != "0":
            hip.rmsnorm_rope_mx(qkv[:, 0:d], p["nq"], qkv[:, d:2 * d], p["nk"], bufs, cd["cos"], cd["sin"], lc, tok0, eps=self.eps)
            hip.attn_fp8_pack(None, None, v4, bufs)
        else:
            hip.rmsnorm_rope(qkv[:, 0:d], p["nq"], qkv[:, d:2 * d], p["nk"], eps=self.eps, rope_cos=cd["cos"], rope_sin=cd["sin"],
                             tokens_per_batch=lc, token_offset=tok0, head_dim=hd)
            hip.attn_fp8_pack(q4, k4, v4, bufs)
        hip.host_op(lambda: all_gather_into_tensor(kv8_all.view(sp * B, *bufs[2].shape[1:]), bufs[2], group=self.sp_group))
        hip.attn_fwd_fp8_chunked(bufs[0], bufs[1], kv8_all, lc, cd["L"], out=ao4)

    def gather_tokens(self, head_local: torch.Tensor) -> torch.Tensor:
        """All-gather of the head output [B, Lc, 192] -> [B, L, 192] (the reference's one collective,
        wan_transformer3d_FlexAM.py:1103-1104)."""
        if self.world_size == 1:
            return head_local
        from .dist import all_gather_seq
        Lr = self.cond["L"]                            # the pad tokens of a sequence that does not divide over the ranks end here
        if self.cfg_size == 1:
            return all_gather_seq(head_local, self.sp_group)[:, :Lr]
        # world rank = cfg_row * sp + sp_rank, one local row each: the rank-major gather IS [2, sp, Lc, n]
        from .dist import all_gather_into_tensor
        bl, lc, n = head_local.shape
        if bl != 1:
            raise RuntimeError("cfg-parallel ranks carry exactly one CFG row")
        out = torch.empty(self.world_size * lc, n, device=head_local.device, dtype=head_local.dtype)   # rank-major concat
        all_gather_into_tensor(out, head_local.reshape(lc, n).contiguous(), group=self.world_group)
        return out.view(self.cfg_size, self.sp_size * lc, n)[:, :Lr]
