"""MI355X drop-in for `AutoencoderKLWan3_8` (Wan2.2 3D-VAE, z = 48, 4x16x16) -- decode and encode paths.

Reference: FlexAM/models/wan_vae3_8.py (:892-1079 wrapper, :739-870 chunked model, :621-728 decoder).
State-dict keys are the reference's (`model.encoder...`, `model.conv1...`, `model.decoder...`,
`model.conv2...`; a full Wan2.2_VAE.pth loads as VAE.py:1073-1077 does).  `decode(z).sample` and
`encode(x).latent_dist` run entirely in libflexam_hip.so:

  * activations are channels-last; every causal 3x3x3 / 3x3 convolution is ONE flexam_gemm_bf16
    launch over a zero-bordered bf16 image with a per-K-block tap-offset table (implicit GEMM, no
    im2col), 27 taps x Cin/64 K-blocks, fp32 accumulate on MFMA;
  * the chunk-causal cache of the reference (feat_cache/feat_idx, CACHE_T = 2) is a 2-frame history
    kept at the front of each conv's input image and rolled after the conv -- same arithmetic, see
    DESIGN.md "VAE chunk cache";
  * RMS_norm + SiLU, nearest-2x upsample (+ frame de-interleave), DupUp3D shortcut, the middle
    attention (head_dim = C, as three GEMMs + a row softmax) and unpatchify + clamp are one-pass
    bandwidth kernels (csrc/vae.hip); the residual stream between blocks stays fp32;
  * encode (VAE.py:788-818, Encoder3d :505-618): patchify writes the 12-channel conv1 image directly; the
    stride-2 spatial convs run as unit-stride implicit GEMMs over a space-to-depth image (the 9 taps address
    (row, col, sub-pixel channel group)); the stride-2 temporal conv is one GEMM per output frame over the
    same history image; AvgDown3D shortcuts are a fused gather-mean-add; the final 1x1 conv and the latent
    normalisation are folded into the head conv's weights (exact in fp32, one GEMM fewer).
"""
import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import hip
from .wan_transformer3d_FlexAM import ModelConfig

BF16, F32, I64 = torch.bfloat16, torch.float32, torch.int64

# latent normalisation statistics of the Wan2.2 VAE (published constants; VAE.py:906-1010)
LATENT_MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891,
               0.1570, -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502,
               -0.2055, -0.0322, 0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723,
               -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
LATENT_STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901,
              0.6885, 0.6165, 0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659,
              0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944,
              0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class DiagonalGaussianDistribution:
    """The posterior object `encode` returns (diffusers' class of the same name, used by VAE.py:1034):
    parameters = [mean | logvar] along dim 1, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


def _round_up(v, m):
    return (v + m - 1) // m * m


def decoder_param_shapes(z_dim=48, dec_dim=256, dim_mult=(1, 2, 4, 4), temporal_up=(True, True, False)) -> Dict[str, tuple]:
    """Parameter inventory of conv2 + Decoder3d under the reference's names (VAE.py:621-675)."""
    dims = [dec_dim * m for m in [dim_mult[-1]] + list(dim_mult[::-1])]
    s: Dict[str, tuple] = {}

    def conv(name, co, ci, k):
        s[name + ".weight"], s[name + ".bias"] = (co, ci, *k), (co,)

    def res(name, ci, co):
        s[name + ".residual.0.gamma"] = (ci, 1, 1, 1)
        conv(name + ".residual.2", co, ci, (3, 3, 3))
        s[name + ".residual.3.gamma"] = (co, 1, 1, 1)
        conv(name + ".residual.6", co, co, (3, 3, 3))
        if ci != co:
            conv(name + ".shortcut", co, ci, (1, 1, 1))
    conv("conv2", z_dim, z_dim, (1, 1, 1))
    conv("decoder.conv1", dims[0], z_dim, (3, 3, 3))
    res("decoder.middle.0", dims[0], dims[0])
    s["decoder.middle.1.norm.gamma"] = (dims[0], 1, 1)
    conv("decoder.middle.1.to_qkv", dims[0] * 3, dims[0], (1, 1))
    conv("decoder.middle.1.proj", dims[0], dims[0], (1, 1))
    res("decoder.middle.2", dims[0], dims[0])
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        c_in = ci
        for j in range(3):
            res(f"decoder.upsamples.{i}.upsamples.{j}", c_in, co)
            c_in = co
        if i != len(dims) - 2:
            conv(f"decoder.upsamples.{i}.upsamples.3.resample.1", co, co, (3, 3))
            if temporal_up[i]:
                conv(f"decoder.upsamples.{i}.upsamples.3.time_conv", co * 2, co, (3, 1, 1))
    s["decoder.head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv("decoder.head.2", 12, dims[-1], (3, 3, 3))
    return s


def encoder_param_shapes(z_dim=48, dim=160, dim_mult=(1, 2, 4, 4), temporal_down=(False, True, True)) -> Dict[str, tuple]:
    """Parameter inventory of conv1 + Encoder3d under the reference's names (VAE.py:505-562, :757-758)."""
    dims = [dim * m for m in [1] + list(dim_mult)]
    s: Dict[str, tuple] = {}

    def conv(name, co, ci, k):
        s[name + ".weight"], s[name + ".bias"] = (co, ci, *k), (co,)

    def res(name, ci, co):
        s[name + ".residual.0.gamma"] = (ci, 1, 1, 1)
        conv(name + ".residual.2", co, ci, (3, 3, 3))
        s[name + ".residual.3.gamma"] = (co, 1, 1, 1)
        conv(name + ".residual.6", co, co, (3, 3, 3))
        if ci != co:
            conv(name + ".shortcut", co, ci, (1, 1, 1))
    conv("conv1", z_dim * 2, z_dim * 2, (1, 1, 1))
    conv("encoder.conv1", dims[0], 12, (3, 3, 3))
    n_stage = len(dims) - 1
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        q = f"encoder.downsamples.{i}.downsamples"
        res(f"{q}.0", ci, co)
        res(f"{q}.1", co, co)
        if i != n_stage - 1:
            conv(f"{q}.2.resample.1", co, co, (3, 3))
            if i < len(temporal_down) and temporal_down[i]:
                conv(f"{q}.2.time_conv", co, co, (3, 1, 1))
    res("encoder.middle.0", dims[-1], dims[-1])
    s["encoder.middle.1.norm.gamma"] = (dims[-1], 1, 1)
    conv("encoder.middle.1.to_qkv", dims[-1] * 3, dims[-1], (1, 1))
    conv("encoder.middle.1.proj", dims[-1], dims[-1], (1, 1))
    res("encoder.middle.2", dims[-1], dims[-1])
    s["encoder.head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv("encoder.head.2", z_dim * 2, dims[-1], (3, 3, 3))
    return s


class _ParamTree(nn.Module):
    """Holds parameters under dotted names ('decoder.middle.0.residual.2.weight') as nested modules."""

    def add(self, dotted: str, shape):
        head, _, rest = dotted.partition(".")
        if not rest:
            self.register_parameter(head, nn.Parameter(torch.zeros(shape)))
            return
        if head not in self._modules:
            self.add_module(head, _ParamTree())
        self._modules[head].add(rest, shape)


class _Conv:
    """One (causal) convolution as an implicit GEMM: packed bf16 weight [Cout, taps*Cp], fp32 bias,
    a padded channels-last input image with `hist` leading history frames, tap-offset tables."""

    def __init__(self, weight, bias, device, t_cap: int):
        w = weight.detach().to(device, F32)
        if w.dim() == 4:
            w = w.unsqueeze(2)
        co, ci, kt, kh, kw = w.shape
        self.co, self.ci, self.kt, self.kh, self.kw = co, ci, kt, kh, kw
        # Run packing (channel counts that are not multiples of 64: the encoder's 160-channel stage and its 12-channel input): the
        # image keeps `ci` channels per pixel (rounded to 8: 16-byte pixels) instead of padding each pixel to 64, and the kw taps of one
        # image row -- kw * cp CONTIGUOUS elements starting at pixel w - 1 -- are one K run, padded to 64 as a whole (3 * 160 = 480 ->
        # 512 instead of 3 * 192 = 576; 3 * 16 = 48 -> 64 instead of 192).  The run's last K block reads a few channels of pixel w + 2:
        # finite activations against zero weights.  FLEXAM_VAE_RUNPACK=0 restores per-pixel padding (A/B only).
        self.run_pack = kw == 3 and ci % 64 != 0 and os.environ.get("FLEXAM_VAE_RUNPACK", "1") != "0"
        self.k_rowmajor = os.environ.get("FLEXAM_VAE_KORDER", "row") != "tap"
        if self.run_pack:
            self.cp = _round_up(ci, 8)
            self.krun = _round_up(kw * self.cp, 64)
            taps = torch.zeros(co, kt, kh, kw, self.cp, device=device, dtype=F32)
            taps[..., :ci] = w.permute(0, 2, 3, 4, 1)
            wp = torch.zeros(co, kt, kh, self.krun, device=device, dtype=F32)
            wp[..., :kw * self.cp] = taps.reshape(co, kt, kh, kw * self.cp)
            self.weight = wp.reshape(co, kt * kh * self.krun).to(BF16).contiguous()
        else:
            self.cp = _round_up(ci, 64)
            wp = torch.zeros(co, kt, kh, kw, self.cp, device=device, dtype=F32)
            wp[..., :ci] = w.permute(0, 2, 3, 4, 1)
            # K order (dt, dh, channel block, dw, 64 channels): the kw taps of one image row are consecutive K blocks, and they read
            # the same 64-channel slice of rows shifted by ONE position -- the second and third hit the L2 lines the first just
            # brought in.  (With the tap-major order (dt, dh, dw, channel block) a shifted re-read comes cp/64 K blocks later, after
            # the XCD's 32 workgroups have pulled 32 x cp/64 x 32 KiB through its 4 MiB L2: the 3x3x3 convs at 256 x 448 then fetch
            # every activation ~9 times over the fabric.)  FLEXAM_VAE_KORDER=tap restores the tap-major order (A/B only).
            if self.k_rowmajor:
                wp = wp.view(co, kt, kh, kw, self.cp // 64, 64).permute(0, 1, 2, 4, 3, 5)
            self.weight = wp.reshape(co, kt * kh * kw * self.cp).to(BF16).contiguous()
        self.bias = bias.detach().to(device, F32).contiguous()
        self.hist = kt - 1
        self.t_cap = t_cap
        self.device = device
        self.shape = None
        self._koff = None

    # Causal convs keep their input frames in a ring of RING chunks: the window [history | current chunk] slides forward by the chunk
    # length after every run, so the last `hist` frames of a chunk ARE the history of the next one where they lie; only when the
    # window reaches the end of the buffer are those frames copied to its start (every RING-th chunk instead of after every chunk:
    # the copies were 3 % of a decode and 5 % of an encode).  Border positions are zero everywhere and never written.
    RING = max(1, int(os.environ.get("FLEXAM_VAE_RING", "4")))

    def image(self, h, w):
        if self.shape != (h, w):
            hp, wp = h + 2, w + 2
            ring = max(1, min(self.RING, 16 // max(self.t_cap, 1))) if self.hist else 1      # long chunks: fewer of them, and a short ring (memory)
            frames = self.hist + self.t_cap * ring
            guard = _round_up((wp + 1) * self.cp + 64, 8)       # + the overrun of a packed run's last K block
            self.buf = torch.zeros(guard * 2 + frames * hp * wp * self.cp, device=self.device, dtype=BF16)
            self.all = self.buf[guard:guard + frames * hp * wp * self.cp].view(frames, hp, wp, self.cp)
            self.cur = 0                                         # first frame of the window
            offs = []
            tap = lambda dt, dh, dw: (dt * hp * wp + (dh - self.kh // 2) * wp + (dw - self.kw // 2)) * self.cp
            for dt in range(self.kt):
                for dh in range(self.kh):
                    if self.run_pack:
                        offs += [tap(dt, dh, 0) + 64 * blk for blk in range(self.krun // 64)]
                    elif self.k_rowmajor:
                        offs += [tap(dt, dh, dw) + cb * 64 for cb in range(self.cp // 64) for dw in range(self.kw)]
                    else:
                        offs += [tap(dt, dh, dw) + cb * 64 for dw in range(self.kw) for cb in range(self.cp // 64)]
            self._koff = torch.tensor(offs, dtype=I64, device=self.device)
            self.shape = (h, w)
            self.img = self.all[:self.hist + self.t_cap]
        return self.img

    def reset(self):
        if self.shape is not None and self.hist:
            self.cur = 0
            self.img = self.all[:self.hist + self.t_cap]
            self.img[:self.hist].zero_()

    def run(self, t, h, w, out_dtype=F32, residual_into=None):
        """Convolve the `t` current frames (window frames hist..hist+t); then slide the window."""
        rows = t * (h + 2) * (w + 2)
        a = self.img.view(-1, self.cp)
        if residual_into is not None:
            out = hip.gemm_gate_residual(a, self.weight, self.bias, residual_into, a_koff=self._koff)
        else:
            out = hip.gemm(a, self.weight, self.bias, a_koff=self._koff, m=rows, k=self.weight.shape[1], out_dtype=out_dtype)
        self.roll(t)
        return out

    def roll(self, t):
        """The last `hist` frames of the chunk become the history of the next one: the window moves on by t frames."""
        if self.hist:
            win = self.hist + self.t_cap
            self.cur += t
            if self.cur + win > self.all.shape[0]:               # end of the ring: bring the history to the front
                src = self.all[self.cur:self.cur + self.hist]
                self.all[:self.hist].copy_(src.clone() if self.cur < self.hist else src)
                self.cur = 0
            self.img = self.all[self.cur:self.cur + win]

    def run_time_stride2(self, t, h, w, out_dtype=F32):
        """(3,1,1) conv with temporal stride 2 and one cached frame (Resample downsample3d, VAE.py:162-174):
        output frame j reads [prev | x][2j .. 2j+2] = image frames 1+2j .. 3+2j; one GEMM per output frame."""
        if t % 2 or (self.kt, self.kh, self.kw) != (3, 1, 1):
            raise RuntimeError("run_time_stride2: needs an even frame count and a (3,1,1) kernel")
        rows = (h + 2) * (w + 2)
        out = torch.empty(t // 2 * rows, self.co, device=self.device, dtype=out_dtype)
        for j in range(t // 2):
            hip.gemm(self.img[1 + 2 * j:].reshape(-1, self.cp), self.weight, self.bias, a_koff=self._koff, m=rows, k=self.weight.shape[1],
                     out=out[j * rows:(j + 1) * rows])
        self.roll(t)
        return out


class _ConvFold(_Conv):
    """A causal convolution with few output channels (the decoder head, 256 -> 12) as a per-tap product + gather: ONE plain GEMM over
    the input pixels, Y[pixel, tap * Cout + o] = W[o, :, tap] . x[pixel, :] (K = Cin: every activation is read once; N = 27 * 12 = 324),
    then flexam_tapsum_cl adds, for every output pixel, the 27 products of its neighbours.  As an implicit GEMM the same convolution has
    K = 27 Cin -- every activation re-read 27 times -- for 12 useful columns of a 160-wide tile: 0.73 ms per 4 frames at 256 x 448, 2.4 %
    of a decode.  Y is recomputed for the two history frames of a chunk (they are the chunk before's last frames)."""

    def __init__(self, weight, bias, device, t_cap: int):
        w = weight.detach().to(device, F32)
        co, ci, kt, kh, kw = w.shape
        assert (kh, kw) == (3, 3)
        self.co, self.ci, self.kt, self.kh, self.kw = co, ci, kt, kh, kw
        self.run_pack, self.k_rowmajor = False, True
        self.cp = _round_up(ci, 64)
        wf = torch.zeros(kt * 9 * co, self.cp, device=device, dtype=F32)
        wf[:, :ci] = w.permute(2, 3, 4, 0, 1).reshape(kt * 9 * co, ci)              # row (dt*9 + dh*3 + dw) * co + o
        self.weight = wf.to(BF16).contiguous()
        self.bias = bias.detach().to(device, F32).contiguous()
        self.hist, self.t_cap, self.device, self.shape, self._koff, self._y = kt - 1, t_cap, device, None, None, None

    def run(self, t, h, w, out_dtype=F32, residual_into=None):
        assert out_dtype == F32 and residual_into is None
        hp, wp = h + 2, w + 2
        rows_all = (self.hist + t) * hp * wp
        if self._y is None or self._y.shape[0] < rows_all:
            self._y = torch.empty((self.hist + self.t_cap) * hp * wp, self.weight.shape[0], device=self.device, dtype=F32)
        y = hip.gemm(self.img.view(-1, self.cp)[:rows_all], self.weight, None, out=self._y[:rows_all])
        out = torch.empty(t * hp * wp, self.co, device=self.device, dtype=F32)      # border rows: never read
        hip.tapsum_cl(y, t, h, w, self.kt, self.co, self.bias, out)
        self.roll(t)
        return out


class _ConvS2D:
    """ZeroPad2d((0,1,0,1)) + Conv2d(3x3, stride 2) (Resample downsample2d/3d, VAE.py:104-113) as a unit-stride
    implicit GEMM over a space-to-depth image [t, H/2+2, W/2+2, 4*Cs]: tap (dh, dw) of the strided conv is
    (row dh>>1, col dw>>1, channel group (dh&1)*2 + (dw&1)) of that image."""

    def __init__(self, weight, bias, device, t_cap: int):
        w = weight.detach().to(device, F32)
        co, ci, kh, kw = w.shape
        assert (kh, kw) == (3, 3)
        self.co, self.ci, self.cs = co, ci, _round_up(ci, 64)
        wp = torch.zeros(co, 3, 3, self.cs, device=device, dtype=F32)
        wp[..., :ci] = w.permute(0, 2, 3, 1)
        self.k_rowmajor = os.environ.get("FLEXAM_VAE_KORDER", "row") != "tap"      # see _Conv: (dh, channel block, dw, 64)
        if self.k_rowmajor:
            wp = wp.view(co, 3, 3, self.cs // 64, 64).permute(0, 1, 3, 2, 4)
        self.weight = wp.reshape(co, 9 * self.cs).to(BF16).contiguous()
        self.bias = bias.detach().to(device, F32).contiguous()
        self.t_cap, self.device, self.shape = t_cap, device, None

    def image(self, h2, w2):
        """h2, w2: OUTPUT resolution."""
        if self.shape != (h2, w2):
            hp, wp, c4 = h2 + 2, w2 + 2, 4 * self.cs
            guard = (wp + 2) * c4
            self.buf = torch.zeros(self.t_cap * hp * wp * c4 + guard, device=self.device, dtype=BF16)
            self.img = self.buf[:self.t_cap * hp * wp * c4].view(self.t_cap, hp, wp, c4)
            offs = []
            tap = lambda dh, dw: (((dh >> 1) * wp + (dw >> 1)) * 4 + (dh & 1) * 2 + (dw & 1)) * self.cs
            for dh in range(3):
                if self.k_rowmajor:
                    offs += [tap(dh, dw) + cb * 64 for cb in range(self.cs // 64) for dw in range(3)]
                else:
                    offs += [tap(dh, dw) + cb * 64 for dw in range(3) for cb in range(self.cs // 64)]
            self._koff = torch.tensor(offs, dtype=I64, device=self.device)
            self.shape = (h2, w2)
        return self.img

    def run(self, t, h2, w2, out_dtype=F32, residual_into=None):
        rows = t * (h2 + 2) * (w2 + 2)
        if residual_into is not None:                  # residual_into[rows, Cout] += conv (fp32, in the GEMM epilogue)
            return hip.gemm_gate_residual(self.img.view(-1, 4 * self.cs), self.weight, self.bias, residual_into[:rows], a_koff=self._koff)
        return hip.gemm(self.img.view(-1, 4 * self.cs), self.weight, self.bias, a_koff=self._koff, m=rows, k=self.weight.shape[1], out_dtype=out_dtype)


class _ConvUp2x:
    """Nearest-exact 2x spatial upsample followed by Conv2d(3x3, padding 1) (Resample upsample2d / upsample3d, VAE.py:76-99) WITHOUT the
    upsampled image.  Output pixel (2y + a, 2x + b) sees, through the 3x3 window on the upsampled image, only the 2x2 block of
    LOW-resolution pixels (y - 1 + a + r, x - 1 + b + c), r, c in {0, 1}: rows 2y - 1 | 2y, 2y + 1 of the upsampled image are rows
    y - 1 | y, y of the input (a = 0) and rows 2y, 2y + 1 | 2y + 2 are rows y, y | y + 1 (a = 1).  So every parity (a, b) is a 2x2
    convolution of the low-resolution image with taps that are SUMS of the original ones (fp32 sums, rounded to bf16 once):
        W'[a][r] = { a=0: (W[-1], W[0] + W[1]);  a=1: (W[-1] + W[0], W[1]) }   (the same along x)
    -- four implicit GEMMs with K = 4 Cin over the low-resolution rows instead of one with K = 9 Cin over four times as many: 16
    instead of 36 tap products per low-resolution pixel (2.25x fewer FLOPs: these convolutions were 10 % of a decode), a quarter of the
    image bytes, and the zero border of the low-resolution image IS the convolution's zero padding (row 2H of the upsampled image
    does not exist, neither does row H of the input).  The four outputs [4, rows, Cout] fp32 are interleaved into the high-
    resolution residual stream by flexam_phase_dupup_cl together with the DupUp3D shortcut."""
    SETS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}        # (parity, r) -> original tap indices (dy + 1)

    def __init__(self, weight, bias, device, t_cap: int):
        w = weight.detach().to(device, F32)
        co, ci, kh, kw = w.shape
        assert (kh, kw) == (3, 3)
        self.co, self.ci, self.cp = co, ci, _round_up(ci, 64)
        self.weights = []
        for a in (0, 1):
            for b in (0, 1):
                wp = torch.zeros(co, 2, 2, self.cp, device=device, dtype=F32)
                for r in (0, 1):
                    for c in (0, 1):
                        wp[:, r, c, :ci] = sum(w[:, :, i, j] for i in self.SETS[(a, r)] for j in self.SETS[(b, c)])
                # K order (r, channel block, c, 64): the two column taps of a row are consecutive K blocks on the same 64 channels (see _Conv)
                wp = wp.view(co, 2, 2, self.cp // 64, 64).permute(0, 1, 3, 2, 4)
                self.weights.append(wp.reshape(co, 4 * self.cp).to(BF16).contiguous())
        self.bias = bias.detach().to(device, F32).contiguous()
        self.t_cap, self.device, self.shape = t_cap, device, None
        self.hist = 0

    def image(self, h, w):
        """h, w: INPUT (low) resolution.  Zero-bordered [t_cap, h + 2, w + 2, cp] with guard rows on both sides (the taps of the first
        and last padded rows reach one row past the image)."""
        if self.shape != (h, w):
            hp, wp, cp = h + 2, w + 2, self.cp
            guard = _round_up((wp + 1) * cp + 64, 8)
            self.buf = torch.zeros(2 * guard + self.t_cap * hp * wp * cp, device=self.device, dtype=BF16)
            self.img = self.buf[guard:guard + self.t_cap * hp * wp * cp].view(self.t_cap, hp, wp, cp)
            self._koff = []
            for a in (0, 1):
                for b in (0, 1):
                    offs = [((a + r - 1) * wp + (b + c - 1)) * cp + cb * 64 for r in (0, 1) for cb in range(cp // 64) for c in (0, 1)]
                    self._koff.append(torch.tensor(offs, dtype=I64, device=self.device))
            self.shape = (h, w)
            self._ph = None
        return self.img

    def run(self, t, h, w):
        """-> phases [4, t*(h+2)*(w+2), Cout] fp32 (rows of the padded LOW-resolution image; border rows hold garbage, never read)."""
        rows = t * (h + 2) * (w + 2)
        if self._ph is None or self._ph.shape[1] < rows:
            self._ph = torch.empty(4, self.t_cap * (h + 2) * (w + 2), self.co, device=self.device, dtype=F32)
        a = self.img.view(-1, self.cp)
        for p in range(4):
            hip.gemm(a, self.weights[p], self.bias, a_koff=self._koff[p], m=rows, k=4 * self.cp, out=self._ph[p, :rows])
        return self._ph[:, :rows]

    def reset(self):
        pass


class _EngineBase:
    """Blocks shared by the decoder and the encoder: ResidualBlock and the middle AttentionBlock."""

    def _setup(self, vae):
        self.sd = sd = {k: v for k, v in vae.model.state_dict().items()}
        self.device = dev = next(vae.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("flexam_amd: the VAE runs only on a GPU through libflexam_hip.so (no CPU fallback)")
        hip.device_check()
        self._scratch = {}
        return sd, dev

    @staticmethod
    def _f32(t, dev):
        return t.detach().to(dev, F32).reshape(-1).contiguous()

    def _mk_conv(self, name, t_cap=1):
        return _Conv(self.sd[name + ".weight"], self.sd[name + ".bias"], self.device, t_cap)

    def _mk_res(self, name, t_cap):
        sd, dev = self.sd, self.device
        d = dict(g0=self._f32(sd[name + ".residual.0.gamma"], dev), c1=self._mk_conv(name + ".residual.2", t_cap),
                 g3=self._f32(sd[name + ".residual.3.gamma"], dev), c2=self._mk_conv(name + ".residual.6", t_cap))
        d["short"] = self._mk_conv(name + ".shortcut", t_cap) if (name + ".shortcut.weight") in sd else None
        return d

    def _mk_attn(self, name, c):
        sd, dev = self.sd, self.device
        wqkv = sd[name + ".to_qkv.weight"].detach().to(dev, F32).reshape(3 * c, c)
        bqkv = self._f32(sd[name + ".to_qkv.bias"], dev)
        return dict(c=c, gamma=self._f32(sd[name + ".norm.gamma"], dev), wqkv=wqkv.to(BF16).contiguous(), bqkv=bqkv,
                    wv=wqkv[2 * c:].to(BF16).contiguous(), bv=bqkv[2 * c:].contiguous(),
                    wproj=sd[name + ".proj.weight"].detach().to(dev, BF16).reshape(c, c).contiguous(),
                    bproj=self._f32(sd[name + ".proj.bias"], dev))

    def _plain_image(self, key, frames, h, w, cp):
        k = (key, frames, h, w, cp)
        if k not in self._scratch:
            self._scratch[k] = torch.zeros(frames, h + 2, w + 2, cp, device=self.device, dtype=BF16)
        return self._scratch[k]

    def _res(self, r, x, t, h, w):
        """ResidualBlock (VAE.py:198-240) on rows x [t*(h+2)*(w+2), Cin] fp32 -> [.., Cout] fp32
        (in place when there is no shortcut conv)."""
        c1, c2 = r["c1"], r["c2"]
        hip.vae_prep_cl(x, c1.ci, t, h, w, c1.image(h, w), mode=2, gamma=r["g0"], t0=c1.hist)
        t1 = c1.run(t, h, w, out_dtype=BF16)
        hip.vae_prep_cl(t1, c2.ci, t, h, w, c2.image(h, w), mode=2, gamma=r["g3"], t0=c2.hist)
        if r["short"] is not None:
            sc = r["short"]
            xb = self._plain_image(("short", sc.ci), t, h, w, sc.cp)
            hip.vae_prep_cl(x, sc.ci, t, h, w, xb, mode=0)
            x = hip.gemm(xb.view(-1, sc.cp), sc.weight, sc.bias, out_dtype=F32)
        c2.run(t, h, w, residual_into=x)
        return x

    def _attention(self, a, x, t, h, w):
        """AttentionBlock (VAE.py:243-282): per frame, one head with head_dim = C."""
        c, dev = a["c"], self.device
        n = h * w
        n4, kp = _round_up(n, 4), _round_up(n, 64)                                         # GEMM N granularity / K granularity
        rows = (h + 2) * (w + 2)
        for f in range(t):
            xf = x[f * rows:(f + 1) * rows]
            xn = torch.zeros(n4, c, device=dev, dtype=BF16)
            hip.vae_prep_cl(xf, c, 1, h, w, xn, mode=1, gamma=a["gamma"], compact=True)
            qk = hip.gemm(xn, a["wqkv"][:2 * c], a["bqkv"][:2 * c])                       # [n4, 2c]
            s = hip.gemm(qk[:n, :c], qk[:, c:], out_dtype=F32)                             # q k^T  [n, n4]; softmax over the first n
            p = torch.empty(n, kp, device=dev, dtype=BF16)
            hip.softmax_rows(s, c ** -0.5, p, n)
            vt = torch.zeros(c, kp, device=dev, dtype=BF16)
            hip.gemm(a["wv"], xn, out=vt[:, :n4])                                          # V^T (bias folded below: rows of P sum to 1)
            o = hip.gemm(p, vt, a["bv"])                                                   # [n, c]
            y = hip.gemm(o, a["wproj"], a["bproj"])
            hip.scatter_add_cl(xf, y, c, 1, h, w)
        return x


class _DecoderEngine(_EngineBase):
    def __init__(self, vae):
        sd, dev = self._setup(vae)
        cfg = vae._arch
        self.z_dim, self.temporal_up = cfg["z_dim"], tuple(cfg["temporal_up"])
        dims = [cfg["dec_dim"] * m for m in [cfg["dim_mult"][-1]] + list(cfg["dim_mult"][::-1])]
        self.dims = dims
        # Latent frames per chunk after the first (1-frame) chunk.  The reference decodes one latent frame at a time (VAE.py:1046-1052) to
        # bound memory; the convolutions are causal over their own input history, so the output does not depend on the chunk length.
        # Several latent frames per chunk fill the 32 x 56 / 64 x 112 stages (1972 / 15048 GEMM rows per latent frame: pure split-K
        # launches at a third of the matrix rate) and cut the launch count.  FLEXAM_VAE_DEC_CHUNK=1 is the reference's walk.
        self.chunk = n = max(1, int(os.environ.get("FLEXAM_VAE_DEC_CHUNK", "2")))
        tmul = [n]
        for up in self.temporal_up:
            tmul.append(tmul[-1] * (2 if up else 1))                       # frames per chunk entering stage i
        conv, res = self._mk_conv, self._mk_res
        self.phase_up = os.environ.get("FLEXAM_VAE_UPCONV", "phase") != "image"
        self.conv2 = conv("conv2")
        self.conv1 = conv("decoder.conv1", n)
        self.mid = [res("decoder.middle.0", n), None, res("decoder.middle.2", n)]
        self.attn = self._mk_attn("decoder.middle.1", dims[0])
        self.stages = []
        n_stage = len(dims) - 1
        for i in range(n_stage):
            p = f"decoder.upsamples.{i}.upsamples"
            st = dict(res=[res(f"{p}.{j}", tmul[i]) for j in range(3)], up=i != n_stage - 1, cout=dims[i + 1])
            if st["up"]:
                st["temporal"] = bool(self.temporal_up[i])
                t_rs = tmul[i] * (2 if st["temporal"] else 1)
                # FLEXAM_VAE_UPCONV=image: the earlier form (upsampled image + one 3x3 convolution over it), A/B and cross-check only
                st["resample"] = (_ConvUp2x(sd[f"{p}.3.resample.1.weight"], sd[f"{p}.3.resample.1.bias"], dev, t_rs) if self.phase_up
                                  else conv(f"{p}.3.resample.1", t_rs))
                st["time_conv"] = conv(f"{p}.3.time_conv", tmul[i]) if st["temporal"] else None
            self.stages.append(st)
        self.head_gamma = self._f32(sd["decoder.head.0.gamma"], dev)
        # FLEXAM_VAE_HEADCONV=implicit: the head as an implicit GEMM like every other convolution (A/B and cross-check)
        if os.environ.get("FLEXAM_VAE_HEADCONV", "fold") != "implicit" and sd["decoder.head.2.weight"].shape[0] % 4 == 0:
            self.head_conv = _ConvFold(sd["decoder.head.2.weight"], sd["decoder.head.2.bias"], dev, tmul[-1])
        else:
            self.head_conv = conv("decoder.head.2", tmul[-1])
        self.mean = torch.tensor(vae.latent_mean, device=dev, dtype=F32)
        self.std = torch.tensor(vae.latent_std, device=dev, dtype=F32)
        self._grid_cache = {}
        del self.sd

    def _all_convs(self):
        out = [self.conv1, self.head_conv]
        blocks = [self.mid[0], self.mid[2]] + [r for st in self.stages for r in st["res"]]
        for r in blocks:
            out += [r["c1"], r["c2"]]
        for st in self.stages:
            if st["up"] and st["time_conv"] is not None:
                out.append(st["time_conv"])
        return out

    def _chunk(self, src_rows, h, w, first, video, f0, stripe=None, t=1):
        """Decoder3d.forward on `t` latent frames (VAE.py:677-728; the first chunk is always the single first frame); writes 1 or 4 t
        frames into `video`.  stripe = {stage: (a, b, ca, cb)}: rows [a, b) x columns [ca, cb) (relative to what it holds at that point) of
        the activation ENTERING that stage are kept and everything after works on that tile (`video` is then the tile's buffer); see
        stripe_plan."""
        c1 = self.conv1
        hip.vae_prep_cl(src_rows, c1.ci, t, h, w, c1.image(h, w), mode=0, t0=c1.hist)
        x = c1.run(t, h, w, out_dtype=F32)
        x = self._res(self.mid[0], x, t, h, w)
        x = self._attention(self.attn, x, t, h, w)
        x = self._res(self.mid[2], x, t, h, w)
        for si, st in enumerate(self.stages):
            if stripe is not None and si in stripe:
                a, b, ca, cb = stripe[si]
                band = torch.zeros(t, b - a + 2, cb - ca + 2, x.shape[1], device=self.device, dtype=F32)
                band[:, 1:-1, 1:-1] = x.view(t, h + 2, w + 2, -1)[:, a + 1:b + 1, ca + 1:cb + 1]
                x, h, w = band.view(-1, x.shape[1]), b - a, cb - ca
            x_in, cin = x, x.shape[1]
            # the residual blocks update their input in place unless the first one has a shortcut convolution (a new tensor): only then
            # does x_in survive without a copy for the DupUp3D shortcut below
            main = x.clone() if (st["up"] and st["res"][0]["short"] is None) else x
            for r in st["res"]:
                main = self._res(r, main, t, h, w)
            if not st["up"]:
                x = main
                continue
            co = st["cout"]
            rs = st["resample"]
            ft = 2 if st["temporal"] else 1
            y = None
            if st["temporal"] and not first:
                tc = st["time_conv"]
                hip.vae_prep_cl(main, co, t, h, w, tc.image(h, w), mode=0, t0=tc.hist)
                y = tc.run(t, h, w, out_dtype=BF16)                                        # [rows, 2*co]: frames 2i | 2i + 1 side by side
            t2 = 2 * t if y is not None else t
            if self.phase_up:                     # 2x2 phase convolutions on the low-resolution frames (_ConvUp2x), then interleave + shortcut
                if y is not None:
                    hip.deinterleave_cl(y, co, t, h, w, rs.image(h, w))
                else:
                    hip.vae_prep_cl(main, co, t, h, w, rs.image(h, w), mode=0, t0=0)
                ph = rs.run(t2, h, w)
                out = torch.empty(t2 * (2 * h + 2) * (2 * w + 2), co, device=self.device, dtype=F32)      # border rows: never read
                hip.phase_dupup_cl(ph, out, co, t2, 2 * h, 2 * w, x_in, cin, ft, (ft - 1) if first else 0)
            else:
                if y is not None:
                    hip.upsample2x_cl(y, co, t, h, w, rs.image(2 * h, 2 * w), interleave=True)
                else:
                    hip.upsample2x_cl(main, co, t, h, w, rs.image(2 * h, 2 * w), interleave=False)
                out = rs.run(t2, 2 * h, 2 * w, out_dtype=F32)
                hip.dupup_add_cl(out, co, t2, 2 * h, 2 * w, x_in, cin, ft, (ft - 1) if first else 0)
            x, t, h, w = out, t2, 2 * h, 2 * w
        hc = self.head_conv
        hip.vae_prep_cl(x, hc.ci, t, h, w, hc.image(h, w), mode=2, gamma=self.head_gamma, t0=hc.hist)
        y = hc.run(t, h, w, out_dtype=F32)                                                 # [rows, 12]
        hip.vae_unpatchify_clamp(y, t, h, w, video, f0)
        return t

    def _axis_need(self, n: int, part: int, parts: int):
        """One axis of stripe_plan: [lo, hi) of the activation ENTERING each stage that part `part` of `parts` needs for its share of the
        2 n 2^ups output rows (or columns) to come out exact, and that share [r0, r1) itself."""
        n_st = len(self.stages)
        N = [n]
        for st in self.stages:
            N.append(N[-1] * (2 if st["up"] else 1))
        n_out = 2 * N[-1]                                      # unpatchify doubles once more
        if n_out % parts:
            raise ValueError(f"{n_out} output rows / columns do not divide over {parts} parts")
        r0, r1 = part * n_out // parts, (part + 1) * n_out // parts
        lo, hi = r0 // 2 - 1, -(-r1 // 2) + 1                  # what the head conv's 3x3 window touches
        need = [None] * n_st
        for si in range(n_st - 1, -1, -1):
            st = self.stages[si]
            lo, hi = max(0, lo), min(N[si + 1], hi)
            if st["up"]:                                       # resample conv: 3x3 at the upsampled resolution = (y - 1) // 2 .. (y + 1) // 2 of the low one
                lo, hi = (lo - 1) // 2, -(-(hi + 1) // 2)
            lo, hi = lo - 2 * len(st["res"]), hi + 2 * len(st["res"])
            need[si] = (max(0, lo), min(N[si], hi))
        return need, N, r0, r1

    def _stage_cost(self):
        """Relative matrix work of a stage per pixel of ITS resolution and latent frame (weights of its convolutions x the frames the
        temporal upsamples before it have made): what band_grid weighs the stages with."""
        out, frames = [], 1
        for st in self.stages:
            n = sum(r[c].weight.numel() for r in st["res"] for c in ("c1", "c2")) + sum(r["short"].weight.numel() for r in st["res"] if r["short"] is not None)
            if st["up"]:
                rs = st["resample"]
                n += sum(wt.numel() for wt in rs.weights) if hasattr(rs, "weights") else 4 * rs.weight.numel()
                if st.get("time_conv") is not None:
                    n += st["time_conv"].weight.numel()
            out.append(n * frames)
            if st["up"] and st["temporal"]:
                frames *= 2
        return out

    def band_grid(self, h: int, w: int, world: int):
        """(rows, columns) of the tile grid the parallel decode cuts the output into: the factorisation of `world` whose SLOWEST tile does
        the least matrix work (area of what it holds in every stage x that stage's work per pixel).  A 97 x 512 x 896 clip on 8 ranks:
        2 x 4 (0.72 of the work of 8 row bands: a tile's halo is a fixed number of rows / columns, so squarer tiles carry less of it)."""
        key = (h, w, world)
        if key in self._grid_cache:
            return self._grid_cache[key]
        cost = self._stage_cost()
        best, best_c = None, None
        for gr in range(world, 0, -1):                         # row bands first: another grid must beat them by 2 %
            if world % gr:
                continue
            gc = world // gr
            try:
                worst = 0.0
                for ri in range(gr):
                    nr, _, _, _ = self._axis_need(h, ri, gr)
                    for ci in range(gc):
                        nc, _, _, _ = self._axis_need(w, ci, gc)
                        c = 0.0
                        for si, st in enumerate(self.stages):
                            c += cost[si] * (nr[si][1] - nr[si][0]) * (nc[si][1] - nc[si][0])
                        worst = max(worst, c)
            except ValueError:
                continue
            if best is None or worst < best_c * 0.98:
                best, best_c = (gr, gc), worst
        if best is None:
            raise ValueError(f"a [{16 * h}, {16 * w}] frame does not divide into {world} equal tiles")
        self._grid_cache[key] = best
        return best

    def stripe_plan(self, h: int, w: int, rank: int, world: int):
        """Tiles of the parallel decode for `rank` of `world`: which rows x columns of the activation ENTERING each stage this
        rank keeps so that its tile of the output comes out EXACT with no exchange.  Walking back from the output, what a stage must
        deliver grows by the receptive field of what follows: 1 for the head conv, 1 (at the upsampled resolution) for a resample conv,
        2 per residual block (two 3x3(x3) convs); a 2x upsample halves the range (_axis_need, the same walk for both axes).  conv1 and
        the middle block (global attention) always run on full frames.  r6: the tile is re-cropped at EVERY stage where that removes
        >= 1/8 of what is held (r1-r5 cropped once, entering stage 2, row bands only: 0.47 of a whole decode per rank at 8 ranks; per-stage
        row bands 0.35; the 2 x 4 grid band_grid picks for the 512 x 896 clip ~0.26).
        Returns (crops, (lo, hi, clo, chi), (gr, gc)): crops = {stage: (a, b, ca, cb)} relative to what is held when entering that stage;
        the tile's video has its own pixels at [lo, hi) x [clo, chi); rank = row * gc + column of the grid."""
        gr, gc = self.band_grid(h, w, world)
        ri, ci = divmod(rank, gc)
        nr, _, r0, r1 = self._axis_need(h, ri, gr)
        nc, _, c0, c1 = self._axis_need(w, ci, gc)
        crops, cur, ccur = {}, (0, h), (0, w)
        for si, st in enumerate(self.stages):
            (a, b), (ca, cb) = nr[si], nc[si]
            held, kept = (cur[1] - cur[0]) * (ccur[1] - ccur[0]), (b - a) * (cb - ca)
            if held - kept >= max(1, held // 8):
                crops[si] = (a - cur[0], b - cur[0], ca - ccur[0], cb - ccur[0])
                cur, ccur = (a, b), (ca, cb)
            if st["up"]:
                cur, ccur = (2 * cur[0], 2 * cur[1]), (2 * ccur[0], 2 * ccur[1])
        return crops, (r0 - 2 * cur[0], r1 - 2 * cur[0], c0 - 2 * ccur[0], c1 - 2 * ccur[0]), (gr, gc)

    @staticmethod
    def assemble_tiles(tiles, grid):
        """Tiles [3, F, rows, cols] of all ranks (rank = row * gc + column) -> the frame."""
        gr, gc = grid
        return torch.cat([torch.cat(list(tiles[r * gc:(r + 1) * gc]), dim=3) for r in range(gr)], dim=2)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, stripe=None) -> torch.Tensor:
        """z [zc, T, H, W] -> video [3, 1 + 4(T-1), 16H, 16W] fp32 in [-1, 1].
        stripe = (rank, world): only this rank's tile of the output, [3, F, 16H/gr, 16W/gc] with (gr, gc) = band_grid(H, W, world)."""
        zc, tz, h, w = z.shape
        for c in self._all_convs():
            c.reset()
        zi = self._plain_image("z", tz, h, w, self.conv2.cp)
        hip.pack_affine_cl(z.to(self.device, F32), self.std, self.mean, zi)                # z / (1/std) + mean
        x0 = hip.gemm(zi.view(-1, self.conv2.cp), self.conv2.weight, self.conv2.bias, out_dtype=F32)
        rows = (h + 2) * (w + 2)
        scale = 2 ** (len(self.stages) - 1)
        tfac = 2 ** sum(self.temporal_up)
        frames = 1 + tfac * (tz - 1)
        if stripe is None or stripe[1] == 1:
            video = torch.empty(3, frames, h * scale * 2, w * scale * 2, device=self.device, dtype=F32)
            self._walk(x0, rows, tz, h, w, video, None)
            return video
        crops, (lo, hi, clo, chi), _ = self.stripe_plan(h, w, *stripe)
        held, cheld = h, w                                     # rows / columns the tile holds at the output resolution
        for si, st in enumerate(self.stages):
            if si in crops:
                held, cheld = crops[si][1] - crops[si][0], crops[si][3] - crops[si][2]
            held, cheld = held * (2 if st["up"] else 1), cheld * (2 if st["up"] else 1)
        band = torch.empty(3, frames, 2 * held, 2 * cheld, device=self.device, dtype=F32)
        self._walk(x0, rows, tz, h, w, band, crops)
        return band[:, :, lo:hi, clo:chi].contiguous()

    def _walk(self, x0, rows, tz, h, w, video, stripe):
        """The chunk walk over the latent frames: frame 0 alone, then `self.chunk` frames at a time."""
        f0, i = 0, 0
        while i < tz:
            t = 1 if i == 0 else min(self.chunk, tz - i)
            f0 += self._chunk(x0[i * rows:(i + t) * rows], h, w, i == 0, video, f0, stripe=stripe, t=t)
            i += t


class _EncoderEngine(_EngineBase):
    """Encoder3d + conv1 + latent normalisation (VAE.py:505-618, :788-818), chunked 1 + 4 + 4 + ... frames."""

    def __init__(self, vae):
        sd, dev = self._setup(vae)
        cfg = vae._arch
        z2 = 2 * cfg["z_dim"]
        self.z2 = z2
        self.temporal_down = tuple(cfg["temporal_down"])
        dims = [cfg["enc_dim"] * m for m in [1] + list(cfg["dim_mult"])]
        self.dims = dims
        # Frames per chunk after the first (1-frame) chunk.  The reference walks the clip in chunks of 4 frames (VAE.py:1029-1037) to bound
        # memory; every convolution here is causal over its own input history, so the result does not depend on where the chunks are cut
        # (only on their being multiples of 4, which keeps the stride-2 time convolutions' frame pairs together).  Long chunks fill the
        # GPU in the low-resolution stages -- at 4 frames the 64 x 112 stage launches 150 tiles and the 32 x 56 stage 24 on 256 CUs --
        # and cut the launch count; 288 GB of HBM hold a whole 97-frame clip's activations.  FLEXAM_VAE_ENC_CHUNK=4 is the reference's walk.
        self.chunk = max(4, int(os.environ.get("FLEXAM_VAE_ENC_CHUNK", "48")) // 4 * 4)
        tcap = [self.chunk]
        for down in self.temporal_down:
            tcap.append(max(1, tcap[-1] // (2 if down else 1)))             # frames per chunk entering stage i
        tcap += [tcap[-1]] * (len(dims) - len(tcap))
        conv, res = self._mk_conv, self._mk_res
        self.conv1 = conv("encoder.conv1", tcap[0])
        self.stages = []
        n_stage = len(dims) - 1
        for i in range(n_stage):
            p = f"encoder.downsamples.{i}.downsamples"
            st = dict(res=[res(f"{p}.{j}", tcap[i]) for j in range(2)], down=i != n_stage - 1, cout=dims[i + 1], temporal=False)
            if st["down"]:
                st["temporal"] = bool(self.temporal_down[i]) if i < len(self.temporal_down) else False
                st["resample"] = _ConvS2D(sd[f"{p}.2.resample.1.weight"], sd[f"{p}.2.resample.1.bias"], dev, tcap[i])
                st["time_conv"] = conv(f"{p}.2.time_conv", tcap[i]) if st["temporal"] else None
            self.stages.append(st)
        t_last = tcap[n_stage - 1] if n_stage - 1 < len(tcap) else 1
        self.mid = [res("encoder.middle.0", t_last), None, res("encoder.middle.2", t_last)]
        self.attn = self._mk_attn("encoder.middle.1", dims[-1])
        self.head_gamma = self._f32(sd["encoder.head.0.gamma"], dev)
        # head conv (C -> 2z) followed by conv1 (1x1x1, 2z -> 2z) and (mu - mean) / std: one linear map, folded in fp32
        w1 = sd["conv1.weight"].detach().to(dev, F32).reshape(z2, z2)
        b1 = sd["conv1.bias"].detach().to(dev, F32)
        wh = sd["encoder.head.2.weight"].detach().to(dev, F32)
        bh = sd["encoder.head.2.bias"].detach().to(dev, F32)
        inv_std = torch.ones(z2, device=dev, dtype=F32)
        mean = torch.zeros(z2, device=dev, dtype=F32)
        inv_std[:z2 // 2] = 1.0 / torch.tensor(vae.latent_std, device=dev, dtype=F32)
        mean[:z2 // 2] = torch.tensor(vae.latent_mean, device=dev, dtype=F32)
        wf = torch.einsum("oc,cikhw->oikhw", w1, wh) * inv_std.view(-1, 1, 1, 1, 1)
        bf = (w1 @ bh + b1 - mean) * inv_std
        self.head_conv = _Conv(wf, bf, dev, t_last)
        del self.sd

    def _all_convs(self):
        out = [self.conv1, self.head_conv]
        blocks = [self.mid[0], self.mid[2]] + [r for st in self.stages for r in st["res"]]
        for r in blocks:
            out += [r["c1"], r["c2"]]
        out += [st["time_conv"] for st in self.stages if st.get("time_conv") is not None]
        return out

    def _chunk(self, video, f0, t, h, w, first):
        """Encoder3d.forward (VAE.py:564-618) on frames f0..f0+t of `video`; returns rows [t'*(h'+2)*(w'+2), 2z] fp32."""
        c1 = self.conv1
        hip.vae_patchify_cl(video, f0, t, c1.image(h, w), t0=c1.hist)
        x = c1.run(t, h, w, out_dtype=F32)
        for st in self.stages:
            x_in, cin, t_in = x, x.shape[1], t
            co = st["cout"]
            # Stage 0 (same width in and out, spatial downsample only): the AvgDown3D shortcut is taken FIRST, into the buffer the
            # stride-2 convolution then adds its result to (fp32 residual epilogue) -- the residual blocks can overwrite x in place, and
            # neither a copy of x (0.3 GB per 4 frames at 256 x 448) nor a second pass over the downsampled rows is needed
            fold = st["down"] and not st["temporal"] and st["res"][0]["short"] is None and cin == co
            if fold:
                short = torch.zeros(t * (h // 2 + 2) * (w // 2 + 2), co, device=self.device, dtype=F32)
                hip.avgdown_add_cl(short, co, t, h // 2, w // 2, x_in, cin, t_in, 1, 2)
                main = x
            else:
                main = x.clone() if st["res"][0]["short"] is None else x
            for r in st["res"]:
                main = self._res(r, main, t, h, w)
            if st["down"]:
                ds = st["resample"]
                h, w = h // 2, w // 2
                hip.space_to_depth_cl(main, co, t, 2 * h, 2 * w, ds.image(h, w), ds.cs)
                if fold:
                    x = ds.run(t, h, w, residual_into=short)
                    continue
                main = ds.run(t, h, w, out_dtype=F32)
                if st["temporal"]:
                    tc = st["time_conv"]
                    hip.vae_prep_cl(main, co, t, h, w, tc.image(h, w), mode=0, t0=tc.hist)
                    if first:
                        tc.roll(t)                                          # first chunk: the frame is only cached (VAE.py:165-166)
                    else:
                        main = tc.run_time_stride2(t, h, w, out_dtype=F32)
                        t //= 2
            hip.avgdown_add_cl(main, co, t, h, w, x_in, cin, t_in, 2 if st["temporal"] else 1, 2 if st["down"] else 1)
            x = main
        x = self._res(self.mid[0], x, t, h, w)
        x = self._attention(self.attn, x, t, h, w)
        x = self._res(self.mid[2], x, t, h, w)
        hc = self.head_conv
        hip.vae_prep_cl(x, hc.ci, t, h, w, hc.image(h, w), mode=2, gamma=self.head_gamma, t0=hc.hist)
        return hc.run(t, h, w, out_dtype=F32), t, h, w

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [3, 1 + 4k, H, W] in [-1, 1] -> [2z, 1 + k, H/16, W/16] fp32: normalised mu | log_var."""
        c, frames, hh, ww = x.shape
        n_down = sum(1 for st in self.stages if st["down"])
        if c != 3 or hh % (2 << n_down) or ww % (2 << n_down):
            raise ValueError(f"encode: expected [3, F, H, W] with H, W multiples of {2 << n_down}, got {tuple(x.shape)}")
        for cv in self._all_convs():
            cv.reset()
        video = x.to(self.device, F32).contiguous()
        h, w = hh // 2, ww // 2
        frames = 1 + (frames - 1) // 4 * 4                 # trailing frames past 1 + 4k are not encoded (VAE.py:1029: iter_ = 1 + (t - 1) // 4)
        outs, t_out, f0 = [], 0, 0
        while f0 < frames:
            t = 1 if f0 == 0 else min(self.chunk, frames - f0)
            rows, to, ho, wo = self._chunk(video, f0, t, h, w, f0 == 0)
            outs.append(rows)
            t_out += to
            f0 += t
        return hip.unpack_cl(torch.cat(outs) if len(outs) > 1 else outs[0], self.z2, t_out, ho, wo)


class AutoencoderKLWan3_8(nn.Module):
    def __init__(self, latent_channels=48, c_dim=160, vae_pth=None, dim_mult=[1, 2, 4, 4], temperal_downsample=[False, True, True],
                 temporal_compression_ratio=4, spatial_compression_ratio=8, dec_dim=256):
        super().__init__()
        self.config = ModelConfig(latent_channels=latent_channels, c_dim=c_dim, vae_pth=vae_pth, dim_mult=list(dim_mult),
                                  temperal_downsample=list(temperal_downsample), temporal_compression_ratio=temporal_compression_ratio,
                                  spatial_compression_ratio=spatial_compression_ratio)
        self.latent_channels = latent_channels
        self.temporal_compression_ratio = temporal_compression_ratio
        self.spatial_compression_ratio = spatial_compression_ratio
        self.latent_mean, self.latent_std = LATENT_MEAN[:latent_channels], LATENT_STD[:latent_channels]
        if latent_channels != len(LATENT_MEAN):
            self.latent_mean, self.latent_std = [0.0] * latent_channels, [1.0] * latent_channels
        temporal_up = list(temperal_downsample)[::-1]
        self._arch = dict(z_dim=latent_channels, dec_dim=dec_dim, enc_dim=c_dim, dim_mult=tuple(dim_mult), temporal_up=tuple(temporal_up),
                          temporal_down=tuple(temperal_downsample))
        self.model = _ParamTree()
        for name, shape in encoder_param_shapes(latent_channels, c_dim, tuple(dim_mult), tuple(temperal_downsample)).items():
            self.model.add(name, shape)
        for name, shape in decoder_param_shapes(latent_channels, dec_dim, tuple(dim_mult), tuple(temporal_up)).items():
            self.model.add(name, shape)
        self.supports_encode = True
        self._engine: Optional[_DecoderEngine] = None
        self._enc_engine: Optional[_EncoderEngine] = None
        self._parallel_group, self._parallel = None, False

    def _apply(self, fn, *a, **k):
        self._engine = self._enc_engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = self._enc_engine = None
        return super().load_state_dict(*a, **k)

    @property
    def dtype(self):
        return next(self.model.parameters()).dtype

    @property
    def device(self):
        return next(self.model.parameters()).device

    def engine(self) -> _DecoderEngine:
        if self._engine is None:
            self._engine = _DecoderEngine(self)
        return self._engine

    def enable_parallel_decode(self, group=None):
        """Tiled decode over the ranks of `group` (replaces the reference's missing `parallel_magvit_vae`,
        FlexAM/models/__init__.py:36-38): every rank decodes 1/N of the output pixels -- a tile of a rows x columns grid -- exactly
        (see _DecoderEngine.stripe_plan) and one all-gather assembles the clip."""
        self._parallel_group, self._parallel = group, True

    def disable_parallel_decode(self):
        self._parallel_group, self._parallel = None, False

    def _decode_one(self, eng, u):
        import torch.distributed as dist
        world = dist.get_world_size(self._parallel_group) if (self._parallel and dist.is_initialized()) else 1
        if world == 1:
            return eng.decode(u)
        band = eng.decode(u, stripe=(dist.get_rank(self._parallel_group), world))
        bands = [torch.empty_like(band) for _ in range(world)]
        dist.all_gather(bands, band, group=self._parallel_group)
        grid = eng.band_grid(u.shape[-2], u.shape[-1], world) if hasattr(eng, "band_grid") else (world, 1)
        return _DecoderEngine.assemble_tiles(bands, grid)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """VAE.py:1041-1056: per sample chunked decode, clamp(-1, 1)."""
        eng = self.engine()
        out = torch.stack([self._decode_one(eng, u) for u in z])
        if z.dtype == BF16:
            out = out.to(BF16)
        return DecoderOutput(out) if return_dict else (out,)

    def encoder_engine(self) -> _EncoderEngine:
        if self._enc_engine is None:
            self._enc_engine = _EncoderEngine(self)
        return self._enc_engine

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """VAE.py:1021-1039: per sample chunked encode -> DiagonalGaussianDistribution([normalised mu | log_var])."""
        eng = self.encoder_engine()
        h = torch.stack([eng.encode(u) for u in x])
        if x.dtype == BF16:
            h = h.to(BF16)
        posterior = DiagonalGaussianDistribution(h)
        return AutoencoderKLOutput(latent_dist=posterior) if return_dict else (posterior,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs={}):
        """VAE.py:1059-1079: raw Wan2.2_VAE.pth keys are prefixed with 'model.'; strict=False."""
        import inspect
        allowed = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in additional_kwargs.items() if k in allowed})
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            state = load_file(pretrained_model_path)
        else:
            state = torch.load(pretrained_model_path, map_location="cpu")
        m, u = model.load_state_dict({"model." + k: v for k, v in state.items()}, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        return model
