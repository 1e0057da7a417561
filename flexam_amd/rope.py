"""Host-side RoPE tables for the 3-axis rotary embedding of the Wan DiT.

The reference keeps a complex128 table `freqs` [1024, head_dim/2] (three axis tables side by
side, FlexAM/models/wan_transformer3d_FlexAM.py:45-52, 655-665) and, per block and per sample,
gathers/expands it to every token on the host before a complex multiply (:137-164).  Here the
per-token cos/sin rows are built once per latent shape ([L, head_dim/2] fp32, 6 MB at L = 11648)
and the HIP kernel flexam_rmsnorm_rope reads one row per token; tokens beyond F*H*W get the
identity rotation (the reference passes them through, :160).
"""
from typing import Optional, Sequence, Tuple

import torch


def axis_split(head_dim: int) -> Tuple[int, int, int]:
    """Complex pairs per axis (frame, height, width): 22/21/21 for head_dim 128."""
    c = head_dim // 2
    return c - 2 * (c // 3), c // 3, c // 3


def rope_angle_table(max_len: int, head_dim: int, theta: float = 10000.0, riflex_k: Optional[int] = None,
                     riflex_l_test: Optional[int] = None, riflex_scale: Optional[float] = None) -> torch.Tensor:
    """Angles [max_len, head_dim/2] fp64: position * theta^(-2i/axis_dim) per axis, with the
    optional RIFLEx change of one temporal frequency (wan_transformer3d_FlexAM.py:57-113, 774-788)."""
    d = head_dim
    cols = []
    for ax, axis_dim in enumerate((d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6))):
        inv = 1.0 / torch.pow(theta, torch.arange(0, axis_dim, 2, dtype=torch.float64) / axis_dim)
        if ax == 0 and riflex_k is not None:
            inv[riflex_k - 1] = 0.9 * 2 * torch.pi / riflex_l_test
            if riflex_scale is not None:
                inv[riflex_k - 1] = inv[riflex_k - 1] / riflex_scale
        cols.append(torch.outer(torch.arange(max_len, dtype=torch.float64), inv))
    return torch.cat(cols, dim=1)


def rope_tables(grid: Sequence[int], seq_len: int, head_dim: int, angles: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos, sin [seq_len, head_dim/2] fp32 for tokens laid out (f, h, w) row-major over `grid`."""
    f, h, w = (int(v) for v in grid)
    if angles is None:
        angles = rope_angle_table(max(f, h, w, 1), head_dim)
    cf, ch, cw = axis_split(head_dim)
    a_f, a_h, a_w = angles.split([cf, ch, cw], dim=1)
    ang = torch.cat([a_f[:f].view(f, 1, 1, cf).expand(f, h, w, cf),
                     a_h[:h].view(1, h, 1, ch).expand(f, h, w, ch),
                     a_w[:w].view(1, 1, w, cw).expand(f, h, w, cw)], dim=-1).reshape(f * h * w, head_dim // 2)
    n = f * h * w
    if n > seq_len:
        raise ValueError(f"grid {tuple(grid)} has {n} tokens > seq_len {seq_len}")
    cos = torch.ones(seq_len, head_dim // 2, dtype=torch.float32)
    sin = torch.zeros(seq_len, head_dim // 2, dtype=torch.float32)
    cos[:n] = ang.cos().float()
    sin[:n] = ang.sin().float()
    return cos, sin
