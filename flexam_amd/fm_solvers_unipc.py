"""MI355X drop-in for `FlexAM.utils.fm_solvers_unipc.FlowUniPCMultistepScheduler` (flow-matching UniPC).

Reference: FlexAM/utils/fm_solvers_unipc.py:20-799 (constructor :75-132, set_timesteps :159-224, predictor :349-478,
corrector :480-615, step :640-724).  Same constructor arguments, `set_timesteps(num_inference_steps, device, sigmas,
mu, shift)`, `step(model_output, timestep, sample, return_dict)`, `.sigmas`, `.timesteps` (int64, truncated),
`.step_index`, `.config`.

Every UniPC update is a linear combination of the sample and the stored x0 predictions whose scalar coefficients
depend only on the sigma schedule, so `step` computes the coefficients on the host in float64 (a handful of
log/expm1 and at most a 3x3 solve) and runs ONE `flexam_lincomb_f32` launch per conversion / corrector / predictor
on the fp32 latents -- no torch arithmetic on device tensors.  Supported configuration: predict_x0, flow_prediction,
bh1 / bh2, no thresholding, no solver_p, no dynamic shifting (what the reference's configs use); anything else raises.
"""
import math
from types import SimpleNamespace
from typing import List, Optional, Union

import numpy as np
import torch

from . import hip

F32 = torch.float32


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


def _lam(sigma: float) -> float:
    a = 1.0 - sigma
    return (math.log(a) if a > 0 else -math.inf) - (math.log(sigma) if sigma > 0 else -math.inf)


def _expm1(x: float) -> float:
    return -1.0 if x == -math.inf else math.expm1(x)


def _base_sigmas(num_train_timesteps: int, shift: float) -> torch.Tensor:
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    sig = torch.from_numpy(1.0 - alphas).to(F32)
    return shift * sig / (1 + (shift - 1) * sig)


def _flow_timesteps(cfg, sigma_max, sigma_min, num_inference_steps, sigmas, shift):
    """fm_solvers_unipc.py:182-206 / fm_solvers.py:249-275: float32 sigmas with a final 0, int64 timesteps."""
    if sigmas is None:
        sigmas = np.linspace(sigma_max, sigma_min, num_inference_steps + 1).copy()[:-1]
    sigmas = np.asarray(sigmas, dtype=np.float64)
    if shift is None:
        shift = cfg.shift
    sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
    timesteps = torch.from_numpy(sigmas * cfg.num_train_timesteps).to(torch.int64)
    return torch.from_numpy(np.concatenate([sigmas, [0.0]]).astype(np.float32)), timesteps


def _as_f32(t: torch.Tensor) -> torch.Tensor:
    if t.device.type != "cuda":
        raise RuntimeError("flexam_amd samplers run on the GPU through libflexam_hip.so (no CPU fallback)")
    return t.to(F32).contiguous()


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting=False, thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector: List[int] = [], solver_p=None,
                 timestep_spacing: str = "linspace", steps_offset: int = 0, final_sigmas_type: Optional[str] = "zero"):
        if solver_type not in ("bh1", "bh2"):
            if solver_type in ("midpoint", "heun", "logrho"):
                solver_type = "bh2"
            else:
                raise NotImplementedError(f"{solver_type} is not implemented for {self.__class__}")
        if prediction_type != "flow_prediction" or not predict_x0 or thresholding or solver_p is not None or use_dynamic_shifting \
                or final_sigmas_type != "zero":
            raise NotImplementedError("FlowUniPCMultistepScheduler (HIP): only predict_x0 / flow_prediction / final sigma zero, "
                                      "without thresholding, solver_p or dynamic shifting")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type,
                                      shift=shift, use_dynamic_shifting=False, thresholding=False, predict_x0=True,
                                      solver_type=solver_type, lower_order_final=lower_order_final,
                                      disable_corrector=list(disable_corrector), final_sigmas_type="zero")
        self.predict_x0 = True
        self.sigmas = _base_sigmas(num_train_timesteps, shift)
        self.timesteps = self.sigmas * num_train_timesteps
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self.num_inference_steps = None
        self.disable_corrector = list(disable_corrector)
        self._reset()

    def _reset(self):
        self.model_outputs = [None] * self.config.solver_order
        self.timestep_list = [None] * self.config.solver_order
        self.lower_order_nums, self.last_sample, self.this_order = 0, None, 1
        self._step_index = self._begin_index = None

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: Union[int, None] = None, device=None, sigmas=None, mu=None, shift=None):
        self.sigmas, ts = _flow_timesteps(self.config, self.sigma_max, self.sigma_min, num_inference_steps, sigmas, shift)
        self.timesteps = ts.to(device) if device is not None else ts
        self.num_inference_steps = len(ts)
        self._reset()

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    # ------------------------------------------------------------------ coefficients (host, float64)
    def _bh(self, sig_t, sig_s0, order, prev_sigmas):
        lam_s0 = _lam(sig_s0)
        h = _lam(sig_t) - lam_s0
        rks = [(_lam(s) - lam_s0) / h for s in prev_sigmas] + [1.0]
        hh = -h
        h_phi_1 = _expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = hh if self.config.solver_type == "bh1" else _expm1(hh)
        rows, b, fact = [], [], 1
        for i in range(1, order + 1):
            rows.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, h_phi_1, b_h, np.array(rows, dtype=np.float64), np.array(b, dtype=np.float64)

    def _predictor_terms(self, x, order):
        i, sg = self._step_index, self.sigmas
        sig_t, sig_s0 = float(sg[i + 1]), float(sg[i])
        alpha_t = 1.0 - sig_t
        rks, h_phi_1, b_h, R, b = self._bh(sig_t, sig_s0, order, [float(sg[i - k]) for k in range(1, order)])
        c_m0 = -alpha_t * h_phi_1
        extra = []
        if order > 1:
            rhos = [0.5] if order == 2 else list(np.linalg.solve(R[:-1, :-1], b[:-1]))
            for k in range(order - 1):                                  # rho_k * D1_k, D1_k = (m_k - m0) / rk
                c = -alpha_t * b_h * rhos[k] / rks[k]
                extra.append((c, self.model_outputs[-(k + 2)]))
                c_m0 -= c
        return [(sig_t / sig_s0, x), (c_m0, self.model_outputs[-1])] + extra

    def _corrector_terms(self, x0_t, order):
        i, sg = self._step_index, self.sigmas
        sig_t, sig_s0 = float(sg[i]), float(sg[i - 1])
        alpha_t = 1.0 - sig_t
        rks, h_phi_1, b_h, R, b = self._bh(sig_t, sig_s0, order, [float(sg[i - (k + 1)]) for k in range(1, order)])
        rhos = [0.5] if order == 1 else list(np.linalg.solve(R, b))
        c_m0 = -alpha_t * h_phi_1
        extra = []
        for k in range(order - 1):
            c = -alpha_t * b_h * rhos[k] / rks[k]
            extra.append((c, self.model_outputs[-(k + 2)]))
            c_m0 -= c
        c_t = -alpha_t * b_h * rhos[-1]                                  # rho_last * (x0_t - m0)
        return [(sig_t / sig_s0, self.last_sample), (c_m0 - c_t, self.model_outputs[-1])] + extra + [(c_t, x0_t)]

    # ------------------------------------------------------------------ step
    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        t = int(timestep)
        idx = (self.timesteps.cpu() == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0]) if len(idx) else 0

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        i = self._step_index
        out_dtype = sample.dtype
        v, sample = _as_f32(model_output), _as_f32(sample)
        x0 = hip.lincomb(torch.empty_like(sample), [(1.0, sample), (-float(self.sigmas[i]), v)])      # convert_model_output
        if i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None:
            sample = hip.lincomb(torch.empty_like(sample), self._corrector_terms(x0, self.this_order))
        self.model_outputs = self.model_outputs[1:] + [x0]
        self.timestep_list = self.timestep_list[1:] + [timestep]
        this_order = min(self.config.solver_order, len(self.timesteps) - i) if self.config.lower_order_final else self.config.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = hip.lincomb(torch.empty_like(sample), self._predictor_terms(sample, self.this_order))
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        prev = prev.to(out_dtype)
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def __len__(self):
        return self.config.num_train_timesteps
