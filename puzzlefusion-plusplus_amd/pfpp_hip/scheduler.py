"""Piecewise-quadratic DDPM schedule and the host side of the ancestral step.

Mirrors PiecewiseScheduler (denoiser/model/modules/custom_diffusers.py:5-69), which replaces
the betas of a diffusers-0.21.4 DDPMScheduler after construction (so beta_schedule /
beta_start / beta_end are accepted and ignored, exactly as there), with the DDPMScheduler
behaviour the reference relies on: `leading` timestep spacing, epsilon prediction,
fixed_small variance, no clipping.  The schedule is a table of 1000 fp32 numbers computed
once on the host; the per-step tensor update runs in the pfpp_ddpm_step / pfpp_add_noise
kernels.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops


def _piecewise_alpha_bar(u: float) -> float:
    s = u * 1000
    if s <= 700:
        return 1 - 0.1 * (s / 700) ** 2          # 1 -> 0.9 over the first 700 steps
    return 0.9 * (1 - ((s - 700) / 300) ** 2)    # 0.9 -> 0 over the last 300


def piecewise_betas(num_steps: int = 1000, max_beta: float = 0.999) -> torch.Tensor:
    vals = [min(1 - _piecewise_alpha_bar((i + 1) / num_steps) / _piecewise_alpha_bar(i / num_steps), max_beta)
            for i in range(num_steps)]
    return torch.tensor(vals, dtype=torch.float32)


class StepOutput(SimpleNamespace):
    pass


class PiecewiseScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "linear",
                 prediction_type: str = "epsilon", beta_start: float = 1e-4, beta_end: float = 2e-2,
                 clip_sample: bool = False, timestep_spacing: str = "leading", **unused):
        if prediction_type != "epsilon":
            raise ValueError("only epsilon prediction is used by PuzzleFusion++ (config/denoiser/model.yaml:21)")
        if timestep_spacing != "leading":
            raise ValueError("only `leading` timestep spacing is supported (config/denoiser/model.yaml:19)")
        if clip_sample:
            raise ValueError("clip_sample=True is not used on this path (denoiser.py:33)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_schedule=beta_schedule,
                                      prediction_type=prediction_type, beta_start=beta_start, beta_end=beta_end,
                                      clip_sample=clip_sample, timestep_spacing=timestep_spacing,
                                      variance_type="fixed_small", steps_offset=0)
        self.betas = piecewise_betas(num_train_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self._dev_cache = {}

    # ------------------------------------------------------------------ timesteps
    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def previous_timestep(self, t: int) -> int:
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    # ------------------------------------------------------------------ coefficients
    def step_coefficients(self, t: int):
        """(c_eps, c_div, c_x0, c_x, c_noise) as python floats holding fp32 values; evaluated with
        0-dim fp32 tensors in the order DDPMScheduler.step evaluates them."""
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        coef = [b_t ** 0.5, a_t ** 0.5, (a_prev ** 0.5 * cur_beta) / b_t, cur_alpha ** 0.5 * b_prev / b_t]
        if t > 0:
            coef.append(torch.clamp(b_prev / b_t * cur_beta, min=1e-20) ** 0.5)
        else:
            coef.append(torch.tensor(0.0))
        return tuple(float(c) for c in coef)

    # ------------------------------------------------------------------ tensor ops (GPU)
    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(abar_t) x0 + sqrt(1 - abar_t) eps with one t per leading-dim entry (denoiser.py:92)"""
        dev = original_samples.device
        key = ("ac", str(dev))
        if key not in self._dev_cache:
            ac = self.alphas_cumprod.to(dev)
            self._dev_cache[key] = (ac ** 0.5, (1 - ac) ** 0.5)
        sa_tab, sb_tab = self._dev_cache[key]
        t = timesteps.to(dev).long().flatten()
        return ops.add_noise(original_samples.contiguous(), noise.contiguous(), sa_tab[t].contiguous(),
                             sb_tab[t].contiguous())

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             variance_noise: Optional[torch.Tensor] = None, ref_part: Optional[torch.Tensor] = None,
             reference: Optional[torch.Tensor] = None) -> StepOutput:
        """one ancestral step; `ref_part`/`reference` optionally fuse the re-pin of the reference
        fragments that always follows in the reference (denoiser.py:184-185)."""
        t = int(timestep)
        coef = self.step_coefficients(t)
        noise = None
        if t > 0:
            noise = variance_noise if variance_noise is not None else torch.randn(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            noise = noise.contiguous()
        ref_u8 = None
        if ref_part is not None:
            ref_u8 = ref_part.reshape(-1).to(torch.uint8).contiguous()
            reference = reference.contiguous()
        prev = ops.ddpm_step(sample.contiguous(), model_output.contiguous(), noise, ref_u8, reference, coef)
        return StepOutput(prev_sample=prev)
