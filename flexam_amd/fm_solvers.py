"""MI355X drop-in for `FlexAM.utils.fm_solvers`: `FlowDPMSolverMultistepScheduler` (flow-matching DPM-Solver++),
`get_sampling_sigmas`, `retrieve_timesteps`.

Reference: FlexAM/utils/fm_solvers.py:22-66 (helpers), :69-856 (scheduler: set_timesteps :226-290, first / second /
third order updates :415-677, step :706-798).  Supported configuration = everything the reference class itself can run:
algorithm_type "dpmsolver++" (orders 1-3) and "sde-dpmsolver++" (orders 1-2; its noise is drawn from the caller's generator the way
the reference draws it), solver_type midpoint | heun, flow_prediction, final sigma zero, with or without dynamic thresholding.
"dpmsolver" / "sde-dpmsolver" raise NotImplementedError here; in the reference they die in set_timesteps (their mandatory
final_sigmas_type "sigma_min" reads an `alphas_cumprod` the flow scheduler never defines, fm_solvers.py:148-175, 249-275).

As for UniPC (fm_solvers_unipc.py) every update is a linear combination of the sample and the stored x0
predictions: coefficients on the host in float64, ONE `flexam_lincomb_f32` launch per conversion and per update.
"""
import inspect
from types import SimpleNamespace
from typing import Optional, Union

import numpy as np
import math

import torch

from . import hip
from .fm_solvers_unipc import SchedulerOutput, _as_f32, _base_sigmas, _expm1, _flow_timesteps, _lam

F32 = torch.float32


def get_sampling_sigmas(sampling_steps, shift):
    """fm_solvers.py:22-26."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """fm_solvers.py:29-66 (same errors)."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    params = set(inspect.signature(scheduler.set_timesteps).parameters.keys())
    if timesteps is not None:
        if "timesteps" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
    elif sigmas is not None:
        if "sigmas" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" sigmas schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        return scheduler.timesteps, num_inference_steps
    return scheduler.timesteps, len(scheduler.timesteps)


class FlowDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting=False, thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0, algorithm_type: str = "dpmsolver++",
                 solver_type: str = "midpoint", lower_order_final: bool = True, euler_at_final: bool = False,
                 final_sigmas_type: Optional[str] = "zero", lambda_min_clipped: float = -float("inf"),
                 variance_type: Optional[str] = None, invert_sigmas: bool = False):
        if algorithm_type == "deis":
            algorithm_type = "dpmsolver++"
        if solver_type in ("logrho", "bh1", "bh2"):
            solver_type = "midpoint"
        # What the reference class itself can run (fm_solvers.py:148-175, 249-275): "dpmsolver" / "sde-dpmsolver" demand
        # final_sigmas_type "sigma_min", whose branch reads an `alphas_cumprod` the flow scheduler never defines (AttributeError in
        # set_timesteps), so the x0-prediction forms are the whole usable surface: dpmsolver++ (orders 1-3) and sde-dpmsolver++
        # (orders 1-2: the reference's third-order update has no SDE form), midpoint / heun, with or without dynamic thresholding.
        if algorithm_type not in ("dpmsolver++", "sde-dpmsolver++") or solver_type not in ("midpoint", "heun") \
                or prediction_type != "flow_prediction" or use_dynamic_shifting or final_sigmas_type != "zero" \
                or solver_order not in (1, 2, 3) or (algorithm_type == "sde-dpmsolver++" and solver_order == 3):
            raise NotImplementedError("FlowDPMSolverMultistepScheduler (HIP): dpmsolver++ (orders 1-3) / sde-dpmsolver++ (orders 1-2) with "
                                      "midpoint / heun, flow_prediction, final sigma zero (the forms the reference class can run)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type,
                                      shift=shift, use_dynamic_shifting=False, thresholding=bool(thresholding),
                                      dynamic_thresholding_ratio=dynamic_thresholding_ratio, sample_max_value=sample_max_value,
                                      algorithm_type=algorithm_type, solver_type=solver_type, lower_order_final=lower_order_final,
                                      euler_at_final=euler_at_final, final_sigmas_type="zero")
        self.sigmas = _base_sigmas(num_train_timesteps, shift)
        self.timesteps = self.sigmas * num_train_timesteps
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self.num_inference_steps = None
        self._reset()

    def _reset(self):
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
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

    def _terms(self, sample, noise=None):
        i, n, sg, cfg = self._step_index, len(self.timesteps), self.sigmas, self.config
        final = i == n - 1                                               # final_sigmas_type == "zero": last step is first order
        second = i == n - 2 and cfg.lower_order_final and n < 15
        sig_t, sig_s0 = float(sg[i + 1]), float(sg[i])
        alpha_t = 1.0 - sig_t
        lam_s0 = _lam(sig_s0)
        h = _lam(sig_t) - lam_s0
        m = self.model_outputs
        if cfg.algorithm_type == "sde-dpmsolver++":                      # fm_solvers.py:473-477, 568-580
            if sig_t == 0.0:                                             # last step: h = inf, exp(-h) = 0 -> x_t = x0, no noise left
                return [(1.0, m[-1])]
            em, e2 = math.exp(-h), -_expm1(-2.0 * h)                     # exp(-h), 1 - exp(-2h)
            out = [(sig_t / sig_s0 * em, sample), (sig_t * math.sqrt(e2), noise)]
            if cfg.solver_order == 1 or self.lower_order_nums < 1 or final:
                return out + [(alpha_t * e2, m[-1])]
            r0 = (lam_s0 - _lam(float(sg[i - 1]))) / h
            c1 = 0.5 * alpha_t * e2 if cfg.solver_type == "midpoint" else alpha_t * (e2 / (-2.0 * h) + 1.0)
            return out + [(alpha_t * e2 + c1 / r0, m[-1]), (-c1 / r0, m[-2])]
        e = _expm1(-h)
        base = [(sig_t / sig_s0, sample)]
        if cfg.solver_order == 1 or self.lower_order_nums < 1 or final:
            return base + [(-alpha_t * e, m[-1])]
        lam_s1 = _lam(float(sg[i - 1]))
        r0 = (lam_s0 - lam_s1) / h
        if cfg.solver_order == 2 or self.lower_order_nums < 2 or second:
            c1 = -0.5 * alpha_t * e if cfg.solver_type == "midpoint" else alpha_t * (e / h + 1.0)
            return base + [(-alpha_t * e + c1 / r0, m[-1]), (-c1 / r0, m[-2])]
        r1 = (lam_s1 - _lam(float(sg[i - 2]))) / h
        c_d1 = alpha_t * (e / h + 1.0)
        c_d2 = -alpha_t * ((e + h) / h ** 2 - 0.5)
        w0 = c_d1 * (1 + r0 / (r0 + r1)) + c_d2 / (r0 + r1)
        w1 = -c_d1 * r0 / (r0 + r1) - c_d2 / (r0 + r1)
        return base + [(-alpha_t * e + w0 / r0, m[-1]), (-w0 / r0 + w1 / r1, m[-2]), (-w1 / r1, m[-3])]

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        idx = (self.timesteps.cpu() == int(timestep)).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0]) if len(idx) else 0

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, variance_noise=None,
             return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._init_step_index(timestep)
        out_dtype = model_output.dtype
        v, sample = _as_f32(model_output), _as_f32(sample)
        x0 = hip.lincomb(torch.empty_like(sample), [(1.0, sample), (-float(self.sigmas[self._step_index]), v)])
        if self.config.thresholding:
            x0 = self._threshold_sample(x0)
        self.model_outputs = self.model_outputs[1:] + [x0]
        noise = None
        if self.config.algorithm_type == "sde-dpmsolver++":              # fm_solvers.py:761-771: drawn on the generator's device
            if variance_noise is not None:
                noise = _as_f32(variance_noise.to(sample.device))
            else:
                gdev = generator.device if generator is not None else sample.device
                noise = torch.randn(sample.shape, generator=generator, device=gdev, dtype=F32).to(sample.device)
        prev = hip.lincomb(torch.empty_like(sample), self._terms(sample, noise))
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        prev = prev.to(out_dtype)
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def _threshold_sample(self, x0: torch.Tensor) -> torch.Tensor:
        """Dynamic thresholding of the x0 prediction (fm_solvers.py:291-326): per batch item, s = the `dynamic_thresholding_ratio`
        quantile of |x0| clamped to [1, sample_max_value]; x0 <- clamp(x0, -s, s) / s.  A handful of PyTorch tensor operations on
        the latent (the quantile is a sort of 0.1 M values), once per step of an option the demo pipelines never select."""
        b = x0.shape[0]
        flat = x0.reshape(b, -1)
        s = torch.quantile(flat.abs(), self.config.dynamic_thresholding_ratio, dim=1)
        s = torch.clamp(s, min=1, max=self.config.sample_max_value).unsqueeze(1)
        return (torch.clamp(flat, -s, s) / s).reshape(x0.shape).contiguous()

    def __len__(self):
        return self.config.num_train_timesteps
