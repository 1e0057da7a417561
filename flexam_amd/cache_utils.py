"""TeaCache bookkeeping (host logic only) with the reference's field names and thresholds
(FlexAM/models/cache_utils.py:21-76).  The decision and the residual re-use inside the DiT forward
(wan_transformer3d_FlexAM.py:977-1051) live in DiTEngine._teacache_decide / DiTEngine.run; residuals stay on
the GPU (the reference's `offload=True` moves them to the host every computed step)."""
import numpy as np
import torch

# polynomial rescale coefficients published with TeaCache for the Wan family (data, not code)
_COEFFICIENTS = {
    "1.3b": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
    "t2v-14b": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
    "i2v-14b-480p": [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01],
    "default-14b": [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02],
}


def get_teacache_coefficients(model_name: str):
    n = model_name.lower()
    if any(k in n for k in ("wan2.1-t2v-1.3b", "wan2.1-fun-1.3b", "wan2.1-fun-v1.1-1.3b", "wan2.1-vace-1.3b")):
        return _COEFFICIENTS["1.3b"]
    if "wan2.1-t2v-14b" in n:
        return _COEFFICIENTS["t2v-14b"]
    if "wan2.1-i2v-14b-480p" in n:
        return _COEFFICIENTS["i2v-14b-480p"]
    if any(k in n for k in ("wan2.1-i2v-14b-720p", "wan2.1-fun-14b", "wan2.2-fun", "wan2.2-i2v-a14b", "wan2.2-t2v-a14b",
                            "wan2.2-ti2v-5b", "wan2.2-s2v", "wan2.1-vace-14b", "wan2.2-vace-fun")):
        return _COEFFICIENTS["default-14b"]
    print(f"The model {model_name} is not supported by TeaCache.")
    return None


class TeaCache:
    def __init__(self, coefficients, num_steps: int, rel_l1_thresh: float = 0.0, num_skip_start_steps: int = 0, offload: bool = True):
        if num_steps < 1:
            raise ValueError(f"`num_steps` must be greater than 0 but is {num_steps}.")
        if rel_l1_thresh < 0:
            raise ValueError(f"`rel_l1_thresh` must be greater than or equal to 0 but is {rel_l1_thresh}.")
        if not 0 <= num_skip_start_steps <= num_steps:
            raise ValueError(f"`num_skip_start_steps` must be in [0, {num_steps}] but is {num_skip_start_steps}.")
        self.coefficients, self.num_steps, self.rel_l1_thresh = coefficients, num_steps, rel_l1_thresh
        self.num_skip_start_steps, self.offload = num_skip_start_steps, offload
        self.rescale_func = np.poly1d(coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev: torch.Tensor, cur: torch.Tensor) -> float:
        return ((cur - prev).abs().mean() / prev.abs().mean()).item()

    def reset(self):
        self.cnt = 0
        self.should_calc = True
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_residual_cond = None
        self.previous_residual_uncond = None
