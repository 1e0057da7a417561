"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's vendored flow-matching multistep samplers
(SURVEY 8 f3):

  FlowUniPCMultistepScheduler       FlexAM/utils/fm_solvers_unipc.py:20-799   (bh1 / bh2, predict_x0, flow_prediction)
  FlowDPMSolverMultistepScheduler   FlexAM/utils/fm_solvers.py:69-856         (dpmsolver++, midpoint / heun, orders 1-3)

Both are linear multistep methods: every update is a linear combination of the current sample and the stored
x0-predictions, with scalar coefficients that depend only on the sigma schedule.  The restatement computes
the coefficients in float64 Python (the reference uses float32 torch scalars) and applies them to tensors;
it is pinned against the reference classes by golden G10 (oracle/make_golden.py) and live in
tests/test_oracle_vs_reference.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Terms = List[Tuple[float, Tensor]]


def flow_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000, init_shift: float = 1.0,
                sigmas: Optional[Sequence[float]] = None, apply_shift: bool = True):
    """set_timesteps of both classes (fm_solvers_unipc.py:159-224, fm_solvers.py:226-290): float32 sigmas with a
    final 0 and int64 (truncated) timesteps.  sigma_max / sigma_min come from the constructor's schedule
    (linspace of alphas, shifted by the constructor's `shift`, unipc.py:105-132)."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    base = torch.from_numpy(1.0 - alphas).to(torch.float32)
    base = init_shift * base / (1 + (init_shift - 1) * base)
    sigma_min, sigma_max = base[-1].item(), base[0].item()
    if sigmas is None:
        sig = np.linspace(sigma_max, sigma_min, num_steps + 1).copy()[:-1]
    else:
        sig = np.asarray(sigmas, dtype=np.float64)
    if apply_shift:
        sig = shift * sig / (1 + (shift - 1) * sig)
    timesteps = torch.from_numpy(sig * num_train_timesteps).to(torch.int64)
    return torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32)), timesteps


def sampling_sigmas(num_steps: int, shift: float) -> np.ndarray:
    """get_sampling_sigmas, fm_solvers.py:22-26 (the DPM++ path of PIPE.py:609-614 passes these as `sigmas`)."""
    s = np.linspace(1, 0, num_steps + 1)[:num_steps]
    return shift * s / (1 + (shift - 1) * s)


def _lam(sigma: float) -> float:
    """lambda = log(alpha) - log(sigma) with alpha = 1 - sigma; +inf at sigma = 0 as in torch."""
    a = 1.0 - sigma
    la = math.log(a) if a > 0 else -math.inf
    ls = math.log(sigma) if sigma > 0 else -math.inf
    return la - ls


def _expm1(x: float) -> float:
    return -1.0 if x == -math.inf else math.expm1(x)


def apply(terms: Terms) -> Tensor:
    out = None
    for c, t in terms:
        out = c * t if out is None else out + c * t
    return out


class UniPC:
    """step(): fm_solvers_unipc.py:640-724; predictor :349-478, corrector :480-615."""

    def __init__(self, sigmas: Tensor, solver_order: int = 2, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: Sequence[int] = ()):
        self.sigmas = [float(s) for s in sigmas]
        self.n = len(self.sigmas) - 1
        self.order, self.solver_type, self.lower_order_final = solver_order, solver_type, lower_order_final
        self.disable_corrector = list(disable_corrector)
        self.m: List[Optional[Tensor]] = [None] * solver_order
        self.lower_order_nums, self.last_sample, self.i, self.this_order = 0, None, 0, 1

    def _common(self, sig_t, sig_s0, order, prev_sigmas):
        lam_t, lam_s0 = _lam(sig_t), _lam(sig_s0)
        h = lam_t - lam_s0
        rks = [(_lam(s) - lam_s0) / h for s in prev_sigmas] + [1.0]
        hh = -h
        h_phi_1 = _expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = hh if self.solver_type == "bh1" else _expm1(hh)
        rows, b, fact = [], [], 1
        for i in range(1, order + 1):
            rows.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, h_phi_1, b_h, np.array(rows, dtype=np.float64), np.array(b, dtype=np.float64)

    def predictor_terms(self, x: Tensor, order: int) -> Terms:
        i = self.i
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        alpha_t = 1.0 - sig_t
        rks, h_phi_1, b_h, R, b = self._common(sig_t, sig_s0, order, [self.sigmas[i - k] for k in range(1, order)])
        m0 = self.m[-1]
        terms = {"x": sig_t / sig_s0, "m0": -alpha_t * h_phi_1}
        if order > 1:
            rhos = [0.5] if order == 2 else list(np.linalg.solve(R[:-1, :-1], b[:-1]))
            for k in range(order - 1):                                  # D1_k = (m_k - m0) / rk
                c = -alpha_t * b_h * rhos[k] / rks[k]
                terms[f"m{k + 1}"] = terms.get(f"m{k + 1}", 0.0) + c
                terms["m0"] -= c
        out = [(terms["x"], x), (terms["m0"], m0)]
        out += [(terms[f"m{k + 1}"], self.m[-(k + 2)]) for k in range(order - 1)]
        return out

    def corrector_terms(self, x0_t: Tensor, order: int) -> Terms:
        i = self.i
        sig_t, sig_s0 = self.sigmas[i], self.sigmas[i - 1]
        alpha_t = 1.0 - sig_t
        rks, h_phi_1, b_h, R, b = self._common(sig_t, sig_s0, order, [self.sigmas[i - (k + 1)] for k in range(1, order)])
        rhos = [0.5] if order == 1 else list(np.linalg.solve(R, b))
        m0 = self.m[-1]
        c_m0 = -alpha_t * h_phi_1
        extra = []
        for k in range(order - 1):
            c = -alpha_t * b_h * rhos[k] / rks[k]
            extra.append((c, self.m[-(k + 2)]))
            c_m0 -= c
        c_t = -alpha_t * b_h * rhos[-1]                                  # * (x0_t - m0)
        c_m0 -= c_t
        return [(sig_t / sig_s0, self.last_sample), (c_m0, m0)] + extra + [(c_t, x0_t)]

    def step(self, model_output: Tensor, sample: Tensor) -> Tensor:
        i = self.i
        x0 = sample - self.sigmas[i] * model_output                       # convert_model_output :310-331
        if i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None:
            sample = apply(self.corrector_terms(x0, self.this_order))
        self.m = self.m[1:] + [x0]
        this_order = min(self.order, self.n - i) if self.lower_order_final else self.order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = apply(self.predictor_terms(sample, self.this_order))
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.i += 1
        return prev


class DPMSolverPP:
    """dpmsolver++ (deterministic) and sde-dpmsolver++ of fm_solvers.py: step :706-798, updates :415-677; dynamic thresholding of
    the x0 prediction :291-326.  (The class's other two algorithm types cannot run in the reference: they demand
    final_sigmas_type "sigma_min", whose branch of set_timesteps reads an attribute the flow scheduler never defines.)"""

    def __init__(self, sigmas: Tensor, solver_order: int = 2, solver_type: str = "midpoint", lower_order_final: bool = True,
                 euler_at_final: bool = False, algorithm_type: str = "dpmsolver++", thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0):
        self.sigmas = [float(s) for s in sigmas]
        self.n = len(self.sigmas) - 1
        self.order, self.solver_type = solver_order, solver_type
        self.lower_order_final, self.euler_at_final = lower_order_final, euler_at_final
        self.sde = algorithm_type == "sde-dpmsolver++"
        self.thresholding, self.ratio, self.max_value = thresholding, dynamic_thresholding_ratio, sample_max_value
        self.m: List[Optional[Tensor]] = [None] * solver_order
        self.lower_order_nums, self.i = 0, 0

    def terms(self, sample: Tensor, noise: Optional[Tensor] = None) -> Terms:
        i, n = self.i, self.n
        final = i == n - 1                                               # final_sigmas_type == "zero" always lowers the last step
        second = i == n - 2 and self.lower_order_final and n < 15
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        alpha_t = 1.0 - sig_t
        h = _lam(sig_t) - _lam(sig_s0)
        m0 = self.m[-1]
        if self.sde:                                                     # :473-477 (first order), :568-580 (second order)
            if sig_t == 0.0:                                             # h = inf: every factor of x and of the noise vanishes
                return [(1.0, m0)]
            em, e2 = math.exp(-h), -_expm1(-2.0 * h)                     # exp(-h), 1 - exp(-2h)
            out = [(sig_t / sig_s0 * em, sample), (sig_t * math.sqrt(e2), noise)]
            if self.order == 1 or self.lower_order_nums < 1 or final:
                return out + [(alpha_t * e2, m0)]
            r0 = (_lam(sig_s0) - _lam(self.sigmas[i - 1])) / h
            c1 = 0.5 * alpha_t * e2 if self.solver_type == "midpoint" else alpha_t * (e2 / (-2.0 * h) + 1.0)
            return out + [(alpha_t * e2 + c1 / r0, m0), (-c1 / r0, self.m[-2])]
        e = _expm1(-h)                                                   # exp(-h) - 1
        base = [(sig_t / sig_s0, sample)]
        if self.order == 1 or self.lower_order_nums < 1 or final:
            return base + [(-alpha_t * e, m0)]
        lam_s0, lam_s1 = _lam(sig_s0), _lam(self.sigmas[i - 1])
        r0 = (lam_s0 - lam_s1) / h
        m1 = self.m[-2]
        if self.order == 2 or self.lower_order_nums < 2 or second:
            c1 = -0.5 * alpha_t * e if self.solver_type == "midpoint" else alpha_t * (e / h + 1.0)
            return base + [(-alpha_t * e + c1 / r0, m0), (-c1 / r0, m1)]  # D1 = (m0 - m1) / r0
        lam_s2 = _lam(self.sigmas[i - 2])
        r1 = (lam_s1 - lam_s2) / h
        m2 = self.m[-3]
        # D1_0 = (m0 - m1)/r0, D1_1 = (m1 - m2)/r1, D1 = D1_0 + r0/(r0+r1) (D1_0 - D1_1), D2 = (D1_0 - D1_1)/(r0+r1)
        c_d1 = alpha_t * (e / h + 1.0)
        c_d2 = -alpha_t * ((e + h) / h ** 2 - 0.5)
        w0 = c_d1 * (1 + r0 / (r0 + r1)) + c_d2 / (r0 + r1)             # coefficient of D1_0
        w1 = -c_d1 * r0 / (r0 + r1) - c_d2 / (r0 + r1)                  # coefficient of D1_1
        return base + [(-alpha_t * e + w0 / r0, m0), (-w0 / r0 + w1 / r1, m1), (-w1 / r1, m2)]

    def step(self, model_output: Tensor, sample: Tensor, generator=None) -> Tensor:
        x0 = sample - self.sigmas[self.i] * model_output
        if self.thresholding:                                            # :291-326
            b = x0.shape[0]
            flat = x0.reshape(b, -1)
            s_ = torch.clamp(torch.quantile(flat.abs(), self.ratio, dim=1), min=1, max=self.max_value).unsqueeze(1)
            x0 = (torch.clamp(flat, -s_, s_) / s_).reshape(x0.shape)
        self.m = self.m[1:] + [x0]
        noise = torch.randn(sample.shape, generator=generator, dtype=torch.float32) if self.sde else None      # :761-766
        prev = apply(self.terms(sample, noise))
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.i += 1
        return prev


class MultistepSchedule:
    """Adapter with the interface oracle.sampler.denoise_loop expects (`set_timesteps(n) -> timesteps`, `step(v, x)`),
    building the schedule the way the reference pipeline does for each family (PIPE.py:606-614)."""

    def __init__(self, kind: str, shift: float = 5.0, generator=None, **kw):
        self.kind, self.shift, self.kw, self.generator = kind, shift, kw, generator
        self.solver = None

    def set_timesteps(self, num_steps: int) -> Tensor:
        if self.kind == "unipc":
            self.sigmas, ts = flow_sigmas(num_steps, self.shift)
            self.solver = UniPC(self.sigmas, **self.kw)
        else:
            self.sigmas, ts = flow_sigmas(num_steps, 1.0, sigmas=sampling_sigmas(num_steps, self.shift))
            self.solver = DPMSolverPP(self.sigmas, **self.kw)
        return ts

    def step(self, v: Tensor, x: Tensor) -> Tensor:
        return self.solver.step(v, x, generator=self.generator) if self.kind != "unipc" else self.solver.step(v, x)
