"""Host-side samplers for the denoising loop.

`DPMSolverMultistepScheduler` mirrors the diffusers==0.24.0 scheduler the reference installs at
/root/reference/train.py:806-808 (`DPMSolverMultistepScheduler.from_config(pipeline.scheduler.config)`:
dpmsolver++, order 2, midpoint, epsilon prediction, lower_order_final) and
`DDPMScheduler.add_noise` as used by /root/reference/utils/common.py:32-48.

The schedule (timesteps, sigmas) and the per-step scalar coefficients are host float64 math; the
tensor update itself is either the torch expression of `step()` (API parity with diffusers) or the
fused HIP kernel `aa_cfg_dpm_step` fed by `coefficients()` (what the pipeline uses).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


class SchedulerOutput(SimpleNamespace):
    pass


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "scaled_linear":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32).astype(np.float64) ** 2
    if beta_schedule == "linear":
        return np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float32).astype(np.float64)
    raise NotImplementedError(f"{beta_schedule} is not implemented")


class DDPMScheduler:
    """Only what the eval path needs: the forward-noising `add_noise`."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", **_):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule)
        self.alphas_cumprod = torch.from_numpy(np.cumprod(1.0 - _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)))

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(original_samples.device)[timesteps.to(original_samples.device).long()]
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        a = acp.sqrt().reshape(shape).to(original_samples.dtype)
        s = (1 - acp).sqrt().reshape(shape).to(original_samples.dtype)
        return a * original_samples + s * noise


class DPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 solver_order=2, prediction_type="epsilon", algorithm_type="dpmsolver++", solver_type="midpoint",
                 lower_order_final=True, timestep_spacing="leading", steps_offset=1, **_):
        if prediction_type != "epsilon" or algorithm_type != "dpmsolver++" or solver_type != "midpoint" or solver_order > 2:
            raise NotImplementedError("only epsilon / dpmsolver++ / midpoint / order<=2 (the reference's configuration)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, solver_order=solver_order,
                                      prediction_type=prediction_type, algorithm_type=algorithm_type,
                                      solver_type=solver_type, lower_order_final=lower_order_final,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        self._acp = np.cumprod(1.0 - _betas(num_train_timesteps, beta_start, beta_end, beta_schedule))
        self.alphas_cumprod = torch.from_numpy(self._acp)
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.sigmas = None

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(vars(config)) if isinstance(config, SimpleNamespace) else dict(config)
        cfg.update(overrides)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps, device=None):
        c, N = self.config, self.config.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, N - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ts = (np.arange(0, num_inference_steps + 1) * (N // (num_inference_steps + 1))).round()[::-1][:-1].copy().astype(np.int64)
            ts = ts + c.steps_offset
        else:
            raise NotImplementedError(c.timestep_spacing)
        sig = ((1 - self._acp) / self._acp) ** 0.5
        sigmas = np.concatenate([np.interp(ts, np.arange(N), sig), [sig[0]]]).astype(np.float32)
        self.sigmas = sigmas.astype(np.float64)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)
        self._ts = [int(t) for t in ts]
        self.num_inference_steps = num_inference_steps
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1.0 / math.sqrt(sigma * sigma + 1.0)
        return alpha, sigma * alpha

    def add_noise(self, original_samples, noise, timesteps):
        idx = [self._ts.index(int(t)) for t in timesteps]
        a = torch.tensor([self._alpha_sigma(self.sigmas[i])[0] for i in idx], dtype=torch.float32)
        s = torch.tensor([self._alpha_sigma(self.sigmas[i])[1] for i in idx], dtype=torch.float32)
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        a = a.reshape(shape).to(original_samples.device, original_samples.dtype)
        s = s.reshape(shape).to(original_samples.device, original_samples.dtype)
        return a * original_samples + s * noise

    # ---- scalar part of one step (shared by `step` and the fused kernel) -------------------------
    def coefficients(self, step_index: int, have_prev: bool):
        """x' = c_x*x - c_d0*x0 - c_d1*(x0 - x0_prev) with x0 = (x - sigma_s*eps)/alpha_s.
        Returns dict(sigma_s, alpha_s, c_x, c_d0, c_d1, second_order)."""
        n = len(self._ts)
        final_low = (step_index == n - 1) and self.config.lower_order_final and n < 15
        a_s, s_s = self._alpha_sigma(self.sigmas[step_index])
        a_t, s_t = self._alpha_sigma(self.sigmas[step_index + 1])
        lam_s, lam_t = math.log(a_s) - math.log(s_s), math.log(a_t) - math.log(s_t)
        h = lam_t - lam_s
        coef = a_t * (math.exp(-h) - 1.0)
        first = self.config.solver_order == 1 or not have_prev or final_low
        c_d1 = 0.0
        if not first:
            a_p, s_p = self._alpha_sigma(self.sigmas[step_index - 1])
            r0 = (lam_s - (math.log(a_p) - math.log(s_p))) / h
            c_d1 = 0.5 * coef / r0
        return dict(sigma_s=s_s, alpha_s=a_s, c_x=s_t / s_s, c_d0=coef, c_d1=c_d1, second_order=not first)

    def index_for_timestep(self, timestep):
        return self._ts.index(int(timestep))

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """diffusers-compatible tensor update (torch ops)."""
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        i = self._step_index
        k = self.coefficients(i, have_prev=self.lower_order_nums >= 1)
        x0 = (sample.float() - k["sigma_s"] * model_output.float()) / k["alpha_s"]
        prev = k["c_x"] * sample.float() - k["c_d0"] * x0
        if k["second_order"]:
            prev = prev - k["c_d1"] * (x0 - self.model_outputs[-1])
        self.model_outputs = self.model_outputs[1:] + [x0]
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        prev = prev.to(sample.dtype)
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)


class EulerDiscreteScheduler:
    """diffusers==0.24.0 EulerDiscreteScheduler in the configuration Stable Video Diffusion ships (the scheduler the
    reference's SVD pipelines drive: /root/reference/models/pipeline.py:399-401 set_timesteps, :419 scale_model_input,
    :440 step; /root/reference/train_svd.py:732-733): v-prediction, Karras sigmas between sigma_min and sigma_max,
    continuous timesteps t = 0.25 ln(sigma), leading spacing (init_noise_sigma = sqrt(sigma_max^2 + 1)).

    The schedule is host math (float64, sigmas rounded to float32 like diffusers); the tensor update is either the torch
    expression of `step()` or the fused HIP kernel `aa_cfg_euler_step_tokens` fed by `coefficients()`."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002,
                 sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1, **_):
        if prediction_type != "v_prediction" or not use_karras_sigmas or timestep_type != "continuous":
            raise NotImplementedError("only v_prediction / Karras sigmas / continuous timesteps (the SVD configuration)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas,
                                      sigma_min=sigma_min, sigma_max=sigma_max, timestep_spacing=timestep_spacing,
                                      timestep_type=timestep_type, steps_offset=steps_offset)
        acp = np.cumprod(1.0 - _betas(num_train_timesteps, beta_start, beta_end, beta_schedule))
        self._train_sigmas = ((1 - acp) / acp) ** 0.5
        self.timesteps = None
        self.sigmas = None
        self._step_index = None

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(vars(config)) if isinstance(config, SimpleNamespace) else dict(config)
        cfg.update(overrides)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        s_min = c.sigma_min if c.sigma_min is not None else float(self._train_sigmas[0])
        s_max = c.sigma_max if c.sigma_max is not None else float(self._train_sigmas[-1])
        rho = 7.0
        ramp = np.linspace(0, 1, num_inference_steps)
        lo, hi = s_min ** (1 / rho), s_max ** (1 / rho)
        sig = ((hi + ramp * (lo - hi)) ** rho).astype(np.float32)
        self._sig = np.concatenate([sig.astype(np.float64), [0.0]])
        self.sigmas = torch.from_numpy(np.concatenate([sig, np.zeros(1, np.float32)]))
        ts = torch.from_numpy((0.25 * np.log(sig.astype(np.float64))).astype(np.float32))
        self.timesteps = ts.to(device) if device is not None else ts
        self._ts = [float(t) for t in ts]
        self.num_inference_steps = num_inference_steps
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = float(self._sig.max())
        return m if self.config.timestep_spacing in ("linspace", "trailing") else (m * m + 1.0) ** 0.5

    def index_for_timestep(self, timestep):
        t = float(timestep)
        return min(range(len(self._ts)), key=lambda i: abs(self._ts[i] - t))

    def input_scale(self, step_index: int) -> float:
        s = self._sig[step_index]
        return 1.0 / math.sqrt(s * s + 1.0)

    def scale_model_input(self, sample, timestep):
        i = self._step_index if self._step_index is not None else self.index_for_timestep(timestep)
        return sample * self.input_scale(i)

    def coefficients(self, step_index: int):
        """x' = c_x * x + c_v * v for  x0 = v * (-s / sqrt(s^2+1)) + x / (s^2+1),  x' = x + (x - x0) / s * (s_next - s)."""
        s, sn = self._sig[step_index], self._sig[step_index + 1]
        r = (sn - s) / s
        return dict(c_x=1.0 + r * (1.0 - 1.0 / (s * s + 1.0)), c_v=r * s / math.sqrt(s * s + 1.0))

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """diffusers-compatible tensor update (torch ops, fp32 arithmetic)."""
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        s, sn = self._sig[self._step_index], self._sig[self._step_index + 1]
        x = sample.float()
        x0 = model_output.float() * (-s / math.sqrt(s * s + 1.0)) + x / (s * s + 1.0)
        prev = (x + (x - x0) / s * (sn - s)).to(sample.dtype)
        self._step_index += 1
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0.to(sample.dtype)) if return_dict else (prev,)
