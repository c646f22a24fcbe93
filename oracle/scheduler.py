"""Oracle scheduler: numpy float64 restatement of diffusers==0.24.0
`DPMSolverMultistepScheduler` (dpmsolver++, order 2, midpoint, epsilon prediction) as the
reference configures it at /root/reference/train.py:806-808, plus the DDPM forward-noising
used by /root/reference/utils/common.py:32-48.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written independently of the product
scheduler (animate_anything_amd/schedulers.py) so the two cross-check each other.
"""
from __future__ import annotations

import numpy as np
import torch


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "scaled_linear":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32)
    else:
        raise NotImplementedError(beta_schedule)
    return np.cumprod(1.0 - betas.astype(np.float64))


def ddpm_add_noise(x0, noise, t, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                   beta_schedule="scaled_linear"):
    """`DDPMScheduler.add_noise`: sqrt(abar_t) x0 + sqrt(1-abar_t) eps."""
    acp = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
    a = float(acp[int(t)])
    return (a ** 0.5) * x0 + ((1.0 - a) ** 0.5) * noise


class DPMSolverMultistepScheduler:
    order = 1  # diffusers' `scheduler.order` attribute read by pipeline.py:158

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", solver_order=2, timestep_spacing="leading", steps_offset=1,
                 lower_order_final=True):
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, solver_order=solver_order,
                           timestep_spacing=timestep_spacing, steps_offset=steps_offset,
                           lower_order_final=lower_order_final)
        self.acp = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.N = num_train_timesteps
        self.solver_order = solver_order
        self.spacing, self.offset, self.lower_order_final = timestep_spacing, steps_offset, lower_order_final
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        if self.spacing == "linspace":
            ts = np.linspace(0, self.N - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.spacing == "leading":
            ratio = self.N // (n + 1)
            ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + self.offset
        else:
            raise NotImplementedError(self.spacing)
        sig_all = ((1 - self.acp) / self.acp) ** 0.5
        sig = np.interp(ts, np.arange(self.N), sig_all)
        self.sigmas = np.concatenate([sig, [sig_all[0]]]).astype(np.float32).astype(np.float64)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self._x0 = [None] * self.solver_order
        self._lower = 0
        self._idx = None

    def scale_model_input(self, sample, t=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1.0 / (sigma * sigma + 1.0) ** 0.5
        return alpha, sigma * alpha

    def add_noise(self, x0, noise, timesteps):
        ts = self.timesteps.tolist()
        idx = [ts.index(int(t)) for t in timesteps]
        out = []
        for b, i in enumerate(idx):
            a, s = self._alpha_sigma(self.sigmas[i])
            out.append(a * x0[b] + s * noise[b])
        return torch.stack(out)

    def step(self, model_output, timestep, sample):
        ts = self.timesteps.tolist()
        if self._idx is None:
            self._idx = ts.index(int(timestep))
        i, n = self._idx, len(ts)
        final_low = (i == n - 1) and self.lower_order_final and n < 15
        a_s, s_s = self._alpha_sigma(self.sigmas[i])
        x0 = (sample.double() - s_s * model_output.double()) / a_s
        self._x0 = self._x0[1:] + [x0]
        a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
        lam_t, lam_s = np.log(a_t) - np.log(s_t), np.log(a_s) - np.log(s_s)
        h = lam_t - lam_s
        coef = a_t * (np.exp(-h) - 1.0)
        if self.solver_order == 1 or self._lower < 1 or final_low:
            prev = (s_t / s_s) * sample.double() - coef * x0
        else:
            a_p, s_p = self._alpha_sigma(self.sigmas[i - 1])
            lam_p = np.log(a_p) - np.log(s_p)
            r0 = (lam_s - lam_p) / h
            d1 = (self._x0[-1] - self._x0[-2]) / r0
            prev = (s_t / s_s) * sample.double() - coef * self._x0[-1] - 0.5 * coef * d1
        if self._lower < self.solver_order:
            self._lower += 1
        self._idx += 1
        return prev.to(sample.dtype)
