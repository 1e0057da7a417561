"""`cfg_skip`: drop the unconditional CFG row for the last `cfg_skip_ratio` of the denoise steps.

Same contract as FlexAM/utils/cfg_optimization.py:5-37 (a decorator factory applied to the DiT's
`forward`): when active, only the second half of every batched argument is forwarded and the result
is duplicated so the caller's `chunk(2)` still works."""
import functools

import numpy as np
import torch

_BATCHED = (torch.Tensor, list, tuple, np.ndarray)


def cfg_skip():
    def decorate(forward):
        @functools.wraps(forward)
        def wrapped(self, x, *args, **kwargs):
            n = len(x)
            ratio = getattr(self, "cfg_skip_ratio", None)
            active = (n >= 2 and ratio is not None and self.num_inference_steps is not None
                      and self.current_steps >= self.num_inference_steps * (1 - ratio))
            if not active:
                return forward(self, x, *args, **kwargs)
            half = n // 2
            cut = lambda v: v[half:] if isinstance(v, _BATCHED) else v
            out = forward(self, x[half:], *[cut(a) for a in args], **{k: cut(v) for k, v in kwargs.items()})
            return torch.cat([out, out], dim=0)
        return wrapped
    return decorate
