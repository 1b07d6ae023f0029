"""CPU restatement of the two samplers on the TextFlux hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations: D = /root/reference/diffusers/src/diffusers.

* flow-matching Euler: D/schedulers/scheduling_flow_match_euler_discrete.py
* AMO / overshoot sampler (TextFlux's own addition):
  D/schedulers/scheduling_stochastic_rf_discrete_overshot.py

Both are restated as pure functions over explicit sigma tables so the trajectory
can be replayed with caller-supplied noise.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def calculate_shift(image_seq_len: int, base_seq_len: int = 256, max_seq_len: int = 4096,
                    base_shift: float = 0.5, max_shift: float = 1.16) -> float:
    """calculate_shift (D/pipelines/flux/pipeline_flux_fill.py:1248-1258).  NB the *function* default
    max_shift is 1.16 but the pipeline passes scheduler.config.max_shift (=1.15 for FLUX.1)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def pipeline_sigmas(num_inference_steps: int) -> np.ndarray:
    """np.linspace(1.0, 1/n, n) (D/pipelines/flux/pipeline_flux_fill.py:2049), float64."""
    return np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)


def _time_shift(mu: float, sigma: float, t: np.ndarray) -> np.ndarray:
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def euler_sigmas(sigmas: Sequence[float], mu: Optional[float], shift: float = 1.0,
                 dynamic: bool = True) -> torch.Tensor:
    """FlowMatchEulerDiscreteScheduler.set_timesteps with `sigmas=` given
    (scheduling_flow_match_euler_discrete.py:209-241): cast to f32 FIRST, then shift, then
    torch f32, append 0.  Returns n+1 sigmas (f32)."""
    s = np.array(sigmas).astype(np.float32)
    s = _time_shift(mu, 1.0, s) if dynamic else shift * s / (1 + (shift - 1) * s)
    s = torch.from_numpy(np.asarray(s)).to(torch.float32)
    return torch.cat([s, torch.zeros(1)])


def amo_sigmas(sigmas: Sequence[float], mu: Optional[float], shift: float = 1.0,
               dynamic: bool = True) -> torch.Tensor:
    """StochasticRFOvershotDiscreteScheduler.set_timesteps with `sigmas=` given
    (scheduling_stochastic_rf_discrete_overshot.py:202-224): NO f32 cast before the shift, so the
    table differs from the Euler one in the last ulp."""
    s = np.asarray(sigmas)
    s = _time_shift(mu, 1.0, s) if dynamic else shift * s / (1 + (shift - 1) * s)
    s = torch.from_numpy(s).to(torch.float32)
    return torch.cat([s, torch.zeros(1)])


def timesteps_from_sigmas(sig: torch.Tensor, num_train_timesteps: int = 1000) -> torch.Tensor:
    return sig[:-1] * num_train_timesteps


def euler_step(model_output: torch.Tensor, sample: torch.Tensor, sigma: torch.Tensor,
               sigma_next: torch.Tensor) -> torch.Tensor:
    """FlowMatchEulerDiscreteScheduler.step (scheduling_flow_match_euler_discrete.py:319-330)."""
    prev = sample.to(torch.float32) + (sigma_next - sigma) * model_output
    return prev.to(model_output.dtype)


def amo_coefficients(sigma: float, sigma_next: float, c: float = 2.0) -> Tuple[float, float, float]:
    """Scalar part of StochasticRFOvershotDiscreteScheduler.step with attn_map None and
    overshot_func = lambda t, dt: t + dt (scheduling_stochastic_rf_discrete_overshot.py:306-349;
    callers run_inference.py:84-88).  Returns (dt_over, a, b) with
        x_over = x + dt_over * (-v);  x' = a * x_over + b * eps.
    The reference evaluates these on 0-dim f32 tensors; this mirrors that in f32."""
    s = torch.tensor(sigma, dtype=torch.float32)
    sn = torch.tensor(sigma_next, dtype=torch.float32)
    t = 1 - s
    step = s - sn
    t_next = min(t + step, 1)
    t_over = min(t_next + step * c, 1)
    a = t_next / t_over
    b = ((1 - t_next) ** 2 - (a - t_next) ** 2) ** 0.5
    return float(t_over - t), float(a), float(b)


def amo_step(model_output: torch.Tensor, sample: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor,
             noise: torch.Tensor, c: float = 2.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Tensor part of the AMO step (same lines).  `noise` is the eps the reference would have drawn with
    randn_tensor(sample.shape, generator=None, dtype=float32) from the global RNG (:351-355)."""
    x = sample.to(torch.float32)
    t = 1 - sigma
    step = sigma - sigma_next
    t_next = min(t + step, 1)
    t_over = min(t_next + step * c, 1)
    x_over = x + (t_over - t) * (-model_output)
    a = t_next / t_over
    b = ((1 - t_next) ** 2 - (a - t_next) ** 2) ** 0.5
    prev = (x_over * a + noise * b).to(model_output.dtype)
    predicted_x1 = x - sigma * model_output
    return prev, predicted_x1
