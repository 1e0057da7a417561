"""Flow-matching Euler scheduler used by the FlexAM sampler.

The reference takes `diffusers.FlowMatchEulerDiscreteScheduler` (pipelines.py:1146-1148, yaml
scheduler_kwargs; used at pipeline_wan2_2_fun_control_FlexAM.py:604-605,931).  diffusers is a
third-party dependency with an unpinned version (requirements.txt: diffusers>=0.30.1) that is not
installed here, so this is a restatement of its published algorithm for the options the reference
sets (shift, num_train_timesteps; use_dynamic_shifting=False) -- PARITY UNPINNED, see DESIGN.md.
"""
from types import SimpleNamespace

import numpy as np
import torch


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False, **unused):
        if use_dynamic_shifting:
            raise NotImplementedError("use_dynamic_shifting is false in config/wan2.2/wan_civitai_5b_FlexAM.yaml")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=False)
        sig = np.arange(num_train_timesteps, 0, -1, dtype=np.float32) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_max, self.sigma_min = float(sig[0]), float(sig[-1])
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = self.sigmas * num_train_timesteps
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int = None, device=None, sigmas=None, mu=None, timesteps=None):
        n = self.config.num_train_timesteps
        custom_t = None
        if sigmas is None:
            if timesteps is not None:                             # custom timesteps: sigmas = t / n, the given timesteps are kept
                custom_t = np.asarray([float(v) for v in timesteps], dtype=np.float64)
                sigmas = custom_t / n
            else:
                if num_inference_steps is None:
                    raise ValueError("set_timesteps needs num_inference_steps, sigmas or timesteps")
                sigmas = np.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps) / n
        sigmas = np.asarray(sigmas, dtype=np.float64)
        s = self.config.shift
        sigmas = s * sigmas / (1 + (s - 1) * sigmas)              # shifted again at inference (double shift)
        sig = torch.from_numpy(sigmas).to(torch.float32)
        ts = torch.from_numpy(custom_t).to(torch.float32) if custom_t is not None else sig * n
        self.timesteps = ts.to(device) if device is not None else ts
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._step_index = None

    def step(self, model_output, timestep, sample, return_dict: bool = False, **unused):
        if self._step_index is None:
            self._step_index = 0
        dt = float(self.sigmas[self._step_index + 1] - self.sigmas[self._step_index])
        prev = (sample.to(torch.float32) + dt * model_output).to(model_output.dtype)
        self._step_index += 1
        return (prev,)

    def sigma_step(self, i: int) -> float:
        """sigma_{i+1} - sigma_i of the current schedule (what the fused HIP step consumes)."""
        return float(self.sigmas[i + 1] - self.sigmas[i])
