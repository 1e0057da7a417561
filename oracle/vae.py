"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the Wan2.2 3D-VAE decode.

Restates /root/reference/FlexAM/models/wan_vae3_8.py ("VAE.py"):
AutoencoderKLWan3_8.decode (:1041-1056) -> AutoencoderKLWan2_2_.decode (:820-849)
-> Decoder3d.forward (:677-728) with CausalConv3d (:22-47), RMS_norm (:50-64),
Resample upsample2d/3d (:76-160), ResidualBlock (:198-240), AttentionBlock
(:243-282), DupUp3D (:375-417), Up_ResidualBlock (:460-502), unpatchify (:304-318).

The reference threads a positional `feat_cache` list through every module and
special-cases short caches; this restatement keeps, per causal conv, a rolling
history of the last two *input* frames initialised to zeros, which is the same
arithmetic (derivation in DESIGN.md "VAE chunk cache"): a 1-frame cache in the
reference is front-padded with one zero frame (VAE.py:41-45), and the "Rep"
marker of upsample3d (VAE.py:120-152) means: chunk 0 skips the time conv and
leaves its history at zero.

Works on a state dict with the reference's key names (prefix "model." as loaded by
VAE.py:1073-1077).  Pinned against the reference module by golden G7/G10 and by
the live comparison test (tests/test_oracle_vs_reference.py).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Latent normalisation constants of AutoencoderKLWan3_8 (VAE.py:906-1010): published data
# of the Wan2.2 VAE (z = 48), not code.
LATENT_MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891,
               0.1570, -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502,
               -0.2055, -0.0322, 0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723,
               -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
LATENT_STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901,
              0.6885, 0.6165, 0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659,
              0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944,
              0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]


def rms_norm_cf(x: Tensor, gamma: Tensor) -> Tensor:
    """RMS_norm, VAE.py:50-64: L2-normalise over channels (dim 1), times sqrt(C) * gamma."""
    c = x.shape[1]
    return F.normalize(x, dim=1) * (c ** 0.5) * gamma.view(1, c, *([1] * (x.dim() - 2)))


class _Hist:
    """Rolling two-frame input history per causal conv (replaces feat_cache/feat_idx)."""

    def __init__(self):
        self.h: Dict[str, Tensor] = {}

    def conv(self, sd, name: str, x: Tensor, update: bool = True) -> Tensor:
        """CausalConv3d with chunk cache (VAE.py:22-47 + callers :219-237)."""
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        kt, kh, kw = w.shape[2:]
        if kt == 1:
            return F.conv3d(x, w, b, padding=(0, kh // 2, kw // 2))
        prev = self.h.get(name)
        if prev is None:
            prev = x.new_zeros(x.shape[0], x.shape[1], kt - 1, *x.shape[3:])
        xin = torch.cat([prev, x], dim=2)
        if update:
            self.h[name] = xin[:, :, -(kt - 1):].clone()
        return F.conv3d(xin, w, b, padding=(0, kh // 2, kw // 2))


def residual_block(sd, p: str, x: Tensor, hist: _Hist) -> Tensor:
    """ResidualBlock, VAE.py:198-240."""
    h = hist.conv(sd, p + ".shortcut", x) if (p + ".shortcut.weight") in sd else x
    y = F.silu(rms_norm_cf(x, sd[p + ".residual.0.gamma"]))
    y = hist.conv(sd, p + ".residual.2", y)
    y = F.silu(rms_norm_cf(y, sd[p + ".residual.3.gamma"]))
    y = hist.conv(sd, p + ".residual.6", y)
    return y + h


def attention_block(sd, p: str, x: Tensor) -> Tensor:
    """AttentionBlock, VAE.py:243-282: per frame, single head with head_dim = C."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm_cf(y, sd[p + ".norm.gamma"])
    qkv = F.conv2d(y, sd[p + ".to_qkv.weight"], sd[p + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, c * 3, h * w).transpose(1, 2).chunk(3, dim=-1)      # [bt, hw, c]
    a = torch.softmax(q @ k.transpose(1, 2) / (c ** 0.5), dim=-1) @ v
    y = a.transpose(1, 2).reshape(b * t, c, h, w)
    y = F.conv2d(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return y.view(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def dup_up3d(x: Tensor, out_c: int, ft: int, fs: int, first_chunk: bool) -> Tensor:
    """DupUp3D, VAE.py:375-417."""
    b, c, t, h, w = x.shape
    rep = out_c * ft * fs * fs // c
    y = x.repeat_interleave(rep, dim=1).view(b, out_c, ft, fs, fs, t, h, w)
    y = y.permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(b, out_c, t * ft, h * fs, w * fs)
    return y[:, :, ft - 1:] if first_chunk else y


def resample_up(sd, p: str, x: Tensor, hist: _Hist, temporal: bool, first_chunk: bool) -> Tensor:
    """Resample upsample2d / upsample3d, VAE.py:117-160."""
    b, c, t, h, w = x.shape
    if temporal and not first_chunk:                         # chunk 0: "Rep", no time conv (VAE.py:122-124)
        y = hist.conv(sd, p + ".time_conv", x)               # [b, 2c, t, h, w]
        y = y.reshape(b, 2, c, t, h, w)
        x = torch.stack((y[:, 0], y[:, 1]), dim=3).reshape(b, c, t * 2, h, w)
        t = t * 2
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").to(x.dtype)
    y = F.conv2d(y, sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], padding=1)
    return y.view(b, t, c, h * 2, w * 2).permute(0, 2, 1, 3, 4)


def decoder_chunk(sd, p: str, x: Tensor, hist: _Hist, first_chunk: bool, temporal_up, taps: Optional[dict] = None) -> Tensor:
    """Decoder3d.forward on one latent frame, VAE.py:677-728."""
    x = hist.conv(sd, p + ".conv1", x)
    x = residual_block(sd, p + ".middle.0", x, hist)
    x = attention_block(sd, p + ".middle.1", x)
    x = residual_block(sd, p + ".middle.2", x, hist)
    if taps is not None:
        taps["middle"] = x.clone()
    n_stage = len(temporal_up) + 1
    for i in range(n_stage):
        q = f"{p}.upsamples.{i}"
        up = i != n_stage - 1
        main = x
        for j in range(3):
            main = residual_block(sd, f"{q}.upsamples.{j}", main, hist)
        if up:
            t_up = bool(temporal_up[i])
            main = resample_up(sd, f"{q}.upsamples.3", main, hist, t_up, first_chunk)
            out_c = main.shape[1]
            x = main + dup_up3d(x, out_c, 2 if t_up else 1, 2, first_chunk)
        else:
            x = main
        if taps is not None:
            taps[f"up{i}"] = x.clone()
    x = F.silu(rms_norm_cf(x, sd[p + ".head.0.gamma"]))
    return hist.conv(sd, p + ".head.2", x)


def unpatchify2(x: Tensor) -> Tensor:
    """unpatchify(patch_size=2), VAE.py:304-318: 'b (c r q) f h w -> b c f (h q) (w r)'."""
    b, crq, f, h, w = x.shape
    c = crq // 4
    y = x.view(b, c, 2, 2, f, h, w)                # [b, c, r, q, f, h, w]
    return y.permute(0, 1, 4, 5, 3, 6, 2).reshape(b, c, f, h * 2, w * 2)


def vae_decode(sd: Dict[str, Tensor], z: Tensor, temporal_up=(True, True, False), mean=None, std=None,
               prefix: str = "model.", taps: Optional[list] = None) -> Tensor:
    """AutoencoderKLWan3_8.decode(z).sample for z [B, zc, T, H, W] -> [B, 3, 1+4(T-1), 16H, 16W].
    `taps` (optional list) receives one dict of intermediate tensors per latent frame."""
    outs = []
    for u in z:
        u = u.unsqueeze(0)
        if mean is not None:
            m = torch.as_tensor(mean, dtype=u.dtype).view(1, -1, 1, 1, 1)
            s = torch.as_tensor(std, dtype=u.dtype).view(1, -1, 1, 1, 1)
            u = u / (1.0 / s) + m                                    # VAE.py:825 with scale = [mean, 1/std]
        x = F.conv3d(u, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
        hist = _Hist()
        frames = []
        for i in range(x.shape[2]):
            chunk_taps = {} if taps is not None else None
            frames.append(decoder_chunk(sd, prefix + "decoder", x[:, :, i:i + 1], hist, i == 0, temporal_up, chunk_taps))
            if taps is not None:
                taps.append(chunk_taps)
        out = unpatchify2(torch.cat(frames, dim=2))
        outs.append(out.clamp(-1, 1).squeeze(0))
    return torch.stack(outs)


def vae_decoder_param_shapes(z_dim: int = 48, dec_dim: int = 256, dim_mult=(1, 2, 4, 4), temporal_up=(True, True, False),
                             prefix: str = "model.") -> Dict[str, tuple]:
    """State-dict inventory of the decode path (conv2 + Decoder3d); names as produced by the
    reference modules (VAE.py:621-675, 198-217, 243-258, 76-113)."""
    dims = [dec_dim * m for m in [dim_mult[-1]] + list(dim_mult[::-1])]
    s = {}

    def conv(name, co, ci, k):
        s[name + ".weight"] = (co, ci, *k)
        s[name + ".bias"] = (co,)

    def res(name, ci, co):
        s[name + ".residual.0.gamma"] = (ci, 1, 1, 1)
        conv(name + ".residual.2", co, ci, (3, 3, 3))
        s[name + ".residual.3.gamma"] = (co, 1, 1, 1)
        conv(name + ".residual.6", co, co, (3, 3, 3))
        if ci != co:
            conv(name + ".shortcut", co, ci, (1, 1, 1))
    conv(prefix + "conv2", z_dim, z_dim, (1, 1, 1))
    d = prefix + "decoder"
    conv(d + ".conv1", dims[0], z_dim, (3, 3, 3))
    res(d + ".middle.0", dims[0], dims[0])
    s[d + ".middle.1.norm.gamma"] = (dims[0], 1, 1)
    conv(d + ".middle.1.to_qkv", dims[0] * 3, dims[0], (1, 1))
    conv(d + ".middle.1.proj", dims[0], dims[0], (1, 1))
    res(d + ".middle.2", dims[0], dims[0])
    n_stage = len(dims) - 1
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        q = f"{d}.upsamples.{i}"
        c_in = ci
        for j in range(3):
            res(f"{q}.upsamples.{j}", c_in, co)
            c_in = co
        if i != n_stage - 1:
            conv(f"{q}.upsamples.3.resample.1", co, co, (3, 3))
            if temporal_up[i]:
                conv(f"{q}.upsamples.3.time_conv", co * 2, co, (3, 1, 1))
    s[d + ".head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv(d + ".head.2", 12, dims[-1], (3, 3, 3))
    return s


# ============================================================================= encode path
# AutoencoderKLWan3_8.encode (VAE.py:1021-1039) -> AutoencoderKLWan2_2_.encode (:788-818) -> Encoder3d (:505-618)
# with Down_ResidualBlock (:420-457), AvgDown3D (:321-372), Resample downsample2d/3d (:104-113,162-174), patchify (:285-301).

def patchify2(x: Tensor) -> Tensor:
    """patchify(patch_size=2): 'b c f (h q) (w r) -> b (c r q) f h w'."""
    b, c, f, hq, wr = x.shape
    y = x.view(b, c, f, hq // 2, 2, wr // 2, 2)          # [b, c, f, h, q, w, r]
    return y.permute(0, 1, 6, 4, 2, 3, 5).reshape(b, c * 4, f, hq // 2, wr // 2)


def avg_down3d(x: Tensor, out_c: int, ft: int, fs: int) -> Tensor:
    """AvgDown3D, VAE.py:340-372 (front zero-pad in time to a multiple of ft)."""
    pad_t = (ft - x.shape[2] % ft) % ft
    x = F.pad(x, (0, 0, 0, 0, pad_t, 0))
    b, c, t, h, w = x.shape
    y = x.view(b, c, t // ft, ft, h // fs, fs, w // fs, fs).permute(0, 1, 3, 5, 7, 2, 4, 6)
    y = y.reshape(b, c * ft * fs * fs, t // ft, h // fs, w // fs)
    g = c * ft * fs * fs // out_c
    return y.view(b, out_c, g, t // ft, h // fs, w // fs).mean(dim=2)


class _EncHist:
    """Per-conv chunk caches of the encoder.  3x3x3 causal convs: rolling two-frame history (as in the decoder).
    Strided time conv (3,1,1)/(2,1,1) of downsample3d: caches ONE frame and is skipped on the first chunk
    (VAE.py:162-174: the first chunk only stores its frame)."""

    def __init__(self):
        self.h: Dict[str, Tensor] = {}

    def conv(self, sd, name, x):
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        kt, kh, kw = w.shape[2:]
        if kt == 1:
            return F.conv3d(x, w, b, padding=(0, kh // 2, kw // 2))
        prev = self.h.get(name)
        if prev is None:
            prev = x.new_zeros(x.shape[0], x.shape[1], kt - 1, *x.shape[3:])
        xin = torch.cat([prev, x], dim=2)
        self.h[name] = xin[:, :, -(kt - 1):].clone()
        return F.conv3d(xin, w, b, padding=(0, kh // 2, kw // 2))

    def time_down(self, sd, name, x):
        prev = self.h.get(name)
        self.h[name] = x[:, :, -1:].clone()
        if prev is None:                                   # first chunk: no temporal conv
            return x
        return F.conv3d(torch.cat([prev, x], dim=2), sd[name + ".weight"], sd[name + ".bias"], stride=(2, 1, 1))


def _enc_res(sd, p, x, hist):
    h = hist.conv(sd, p + ".shortcut", x) if (p + ".shortcut.weight") in sd else x
    y = F.silu(rms_norm_cf(x, sd[p + ".residual.0.gamma"]))
    y = hist.conv(sd, p + ".residual.2", y)
    y = F.silu(rms_norm_cf(y, sd[p + ".residual.3.gamma"]))
    y = hist.conv(sd, p + ".residual.6", y)
    return y + h


def encoder_chunk(sd, p: str, x: Tensor, hist: _EncHist, temporal_down) -> Tensor:
    """Encoder3d.forward on one chunk (1 frame, then 4 frames), VAE.py:564-618."""
    x = hist.conv(sd, p + ".conv1", x)
    n_stage = len(temporal_down) + 1
    for i in range(n_stage):
        q = f"{p}.downsamples.{i}"
        down = i != n_stage - 1
        t_down = bool(temporal_down[i]) if i < len(temporal_down) else False
        x_in = x
        for j in range(2):
            x = _enc_res(sd, f"{q}.downsamples.{j}", x, hist)
        if down:
            b, c, t, h, w = x.shape
            y = F.pad(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), (0, 1, 0, 1))
            y = F.conv2d(y, sd[f"{q}.downsamples.2.resample.1.weight"], sd[f"{q}.downsamples.2.resample.1.bias"], stride=2)
            x = y.view(b, t, c, h // 2, w // 2).permute(0, 2, 1, 3, 4)
            if t_down:
                x = hist.time_down(sd, f"{q}.downsamples.2.time_conv", x)
        out_c = x.shape[1]
        x = x + avg_down3d(x_in, out_c, 2 if t_down else 1, 2 if down else 1)
    x = _enc_res(sd, p + ".middle.0", x, hist)
    x = attention_block(sd, p + ".middle.1", x)
    x = _enc_res(sd, p + ".middle.2", x, hist)
    x = F.silu(rms_norm_cf(x, sd[p + ".head.0.gamma"]))
    return hist.conv(sd, p + ".head.2", x)


def vae_encode(sd: Dict[str, Tensor], x: Tensor, temporal_down=(False, True, True), mean=None, std=None, prefix: str = "model.") -> Tensor:
    """AutoencoderKLWan3_8.encode(x)[0].mode() for x [B, 3, 1+4k, H, W] in [-1, 1] -> mu [B, zc, 1+k, H/16, W/16]
    (already normalised: (mu - mean) / std, VAE.py:810-815)."""
    outs = []
    for u in x:
        u = patchify2(u.unsqueeze(0))
        t = u.shape[2]
        hist = _EncHist()
        chunks = []
        for i in range(1 + (t - 1) // 4):
            sl = u[:, :, :1] if i == 0 else u[:, :, 1 + 4 * (i - 1):1 + 4 * i]
            chunks.append(encoder_chunk(sd, prefix + "encoder", sl, hist, temporal_down))
        out = torch.cat(chunks, dim=2)
        mu = F.conv3d(out, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"]).chunk(2, dim=1)[0]
        if mean is not None:
            m = torch.as_tensor(mean, dtype=mu.dtype).view(1, -1, 1, 1, 1)
            s = torch.as_tensor(std, dtype=mu.dtype).view(1, -1, 1, 1, 1)
            mu = (mu - m) * (1.0 / s)
        outs.append(mu.squeeze(0))
    return torch.stack(outs)


def vae_encoder_param_shapes(z_dim: int = 48, dim: int = 160, dim_mult=(1, 2, 4, 4), temporal_down=(False, True, True),
                             prefix: str = "model.") -> Dict[str, tuple]:
    dims = [dim * m for m in [1] + list(dim_mult)]
    s = {}

    def conv(name, co, ci, k):
        s[name + ".weight"] = (co, ci, *k)
        s[name + ".bias"] = (co,)

    def res(name, ci, co):
        s[name + ".residual.0.gamma"] = (ci, 1, 1, 1)
        conv(name + ".residual.2", co, ci, (3, 3, 3))
        s[name + ".residual.3.gamma"] = (co, 1, 1, 1)
        conv(name + ".residual.6", co, co, (3, 3, 3))
        if ci != co:
            conv(name + ".shortcut", co, ci, (1, 1, 1))
    conv(prefix + "conv1", z_dim * 2, z_dim * 2, (1, 1, 1))
    e = prefix + "encoder"
    conv(e + ".conv1", dims[0], 12, (3, 3, 3))
    n_stage = len(dims) - 1
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        q = f"{e}.downsamples.{i}.downsamples"
        res(f"{q}.0", ci, co)
        res(f"{q}.1", co, co)
        if i != n_stage - 1:
            conv(f"{q}.2.resample.1", co, co, (3, 3))
            if i < len(temporal_down) and temporal_down[i]:
                conv(f"{q}.2.time_conv", co, co, (3, 1, 1))
    res(e + ".middle.0", dims[-1], dims[-1])
    s[e + ".middle.1.norm.gamma"] = (dims[-1], 1, 1)
    conv(e + ".middle.1.to_qkv", dims[-1] * 3, dims[-1], (1, 1))
    conv(e + ".middle.1.proj", dims[-1], dims[-1], (1, 1))
    res(e + ".middle.2", dims[-1], dims[-1])
    s[e + ".head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv(e + ".head.2", z_dim * 2, dims[-1], (3, 3, 3))
    return s
