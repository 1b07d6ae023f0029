"""The two samplers of the TextFlux hot path, same protocol as the reference classes
(`set_timesteps(num_inference_steps, device, sigmas=, mu=)`, `step(model_output, timestep, sample, return_dict=False)[0]`,
`.timesteps`, `.sigmas`, `.config`, `.order`, plus AMO's `set_c` / `set_overshot_func`).

* FlowMatchEulerDiscreteScheduler         <- D/schedulers/scheduling_flow_match_euler_discrete.py
* StochasticRFOvershotDiscreteScheduler   <- D/schedulers/scheduling_stochastic_rf_discrete_overshot.py  (AMO sampler)

Host side (this file): the sigma / timestep tables and the per-step scalar coefficients, evaluated in the same
precision and order as the reference.  Device side: the state update itself runs in libtextflux_hip.so
(`tfx_euler_step` / `tfx_amo_step`); the AMO sampler's per-step `min(tensor, 1)` host syncs (reference :313, :342)
disappear because the coefficient table is built once from the host copy of the sigmas.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)

    def to_dict(self):
        return dict(self.__dict__)


def _cfg_dict(config) -> dict:
    if isinstance(config, dict):
        return dict(config)
    if hasattr(config, "to_dict"):
        return config.to_dict()
    return dict(vars(config))


class _FlowMatchBase:
    order = 1
    _keys: Sequence[str] = ()

    def _init_tables(self):
        c = self.config
        ts = np.linspace(1, c.num_train_timesteps, c.num_train_timesteps, dtype=np.float32)[::-1].copy()
        sigmas = torch.from_numpy(ts).to(torch.float32) / c.num_train_timesteps
        if not c.use_dynamic_shifting:
            sigmas = c.shift * sigmas / (1 + (c.shift - 1) * sigmas)
        self.timesteps = sigmas * c.num_train_timesteps
        self.sigmas = sigmas.to("cpu")
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self._step_index = self._begin_index = None
        self._sigmas_host: Optional[torch.Tensor] = None
        self._coef: Optional[torch.Tensor] = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config, **kwargs):
        """Accepts another scheduler's config (dict or namespace) and ignores keys it lacks, as ConfigMixin does
        (used at run_inference.py:81-82 to swap the sampler)."""
        d = _cfg_dict(config)
        d.update(kwargs)
        return cls(**{k: d[k] for k in cls._keys if k in d})

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    def time_shift(self, mu: float, sigma: float, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        st = self.timesteps if schedule_timesteps is None else schedule_timesteps
        idx = (st == timestep).nonzero()
        pos = 1 if len(idx) > 1 else 0
        return idx[pos].item()

    def _init_step_index(self, timestep):
        if self.begin_index is None:
            if isinstance(timestep, torch.Tensor):
                timestep = timestep.to(self.timesteps.device)
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def _finish_set_timesteps(self, sigmas_np, device):
        sig = torch.from_numpy(np.asarray(sigmas_np)).to(dtype=torch.float32)
        self._sigmas_host = torch.cat([sig, torch.zeros(1)])          # host copy: coefficient tables, no syncs later
        sig_dev = sig.to(device=device)
        self.timesteps = (sig_dev * self.config.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sig_dev, torch.zeros(1, device=sig_dev.device)])
        self._step_index = self._begin_index = None
        self._coef = None

    @staticmethod
    def _check_timestep(timestep):
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to `step()` is "
                             "not supported. Make sure to pass one of the `scheduler.timesteps` as a timestep.")

    def __len__(self):
        return self.config.num_train_timesteps


class FlowMatchEulerDiscreteScheduler(_FlowMatchBase):
    _keys = ("num_train_timesteps", "shift", "use_dynamic_shifting", "base_shift", "max_shift", "base_image_seq_len",
             "max_image_seq_len", "invert_sigmas", "use_karras_sigmas", "use_exponential_sigmas", "use_beta_sigmas")

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting=False,
                 base_shift: Optional[float] = 0.5, max_shift: Optional[float] = 1.15,
                 base_image_seq_len: Optional[int] = 256, max_image_seq_len: Optional[int] = 4096,
                 invert_sigmas: bool = False, use_karras_sigmas: Optional[bool] = False,
                 use_exponential_sigmas: Optional[bool] = False, use_beta_sigmas: Optional[bool] = False):
        if invert_sigmas or use_karras_sigmas or use_exponential_sigmas or use_beta_sigmas:
            raise NotImplementedError("sigma re-parameterisations are not used by FLUX.1-Fill / TextFlux (SURVEY Appendix C)")
        self.config = _Config(num_train_timesteps=num_train_timesteps, shift=shift,
                              use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift, max_shift=max_shift,
                              base_image_seq_len=base_image_seq_len, max_image_seq_len=max_image_seq_len,
                              invert_sigmas=invert_sigmas, use_karras_sigmas=use_karras_sigmas,
                              use_exponential_sigmas=use_exponential_sigmas, use_beta_sigmas=use_beta_sigmas)
        self._init_tables()

    def set_timesteps(self, num_inference_steps: int = None, device: Union[str, torch.device] = None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None):
        """scheduling_flow_match_euler_discrete.py:184-241 -- note the cast to f32 BEFORE the shift."""
        c = self.config
        if c.use_dynamic_shifting and mu is None:
            raise ValueError(" you have a pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            timesteps = np.linspace(self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min), num_inference_steps)
            sigmas = timesteps / c.num_train_timesteps
        else:
            sigmas = np.array(sigmas).astype(np.float32)
            num_inference_steps = len(sigmas)
        self.num_inference_steps = num_inference_steps
        sigmas = self.time_shift(mu, 1.0, sigmas) if c.use_dynamic_shifting else c.shift * sigmas / (1 + (c.shift - 1) * sigmas)
        self._finish_set_timesteps(sigmas, device)

    def coef_table(self, device, state_dtype=torch.bfloat16) -> torch.Tensor:
        """dsigma_i = sigma_{i+1} - sigma_i (f32), rounded to the model-output dtype exactly where the reference's
        `(sigma_next - sigma) * model_output` rounds it (0-dim f32 tensor cast to the common dtype, :327)."""
        if self._coef is None or self._coef.device != torch.device(device):
            s = self._sigmas_host
            self._coef = (s[1:] - s[:-1]).to(state_dtype).float().to(device)
        return self._coef

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True):
        self._check_timestep(timestep)
        if self.step_index is None:
            self._init_step_index(timestep)
        prev = sample.to(model_output.dtype).clone().contiguous()
        ops.euler_step_(model_output.contiguous(), prev, self.coef_table(sample.device, model_output.dtype),
                        step=self._step_index)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return SimpleNamespace(prev_sample=prev)


class StochasticRFOvershotDiscreteScheduler(_FlowMatchBase):
    _keys = ("num_train_timesteps", "shift", "use_dynamic_shifting", "base_shift", "max_shift", "base_image_seq_len",
             "max_image_seq_len")

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting=False,
                 base_shift: Optional[float] = 0.5, max_shift: Optional[float] = 1.15,
                 base_image_seq_len: Optional[int] = 256, max_image_seq_len: Optional[int] = 4096):
        self.config = _Config(num_train_timesteps=num_train_timesteps, shift=shift,
                              use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift, max_shift=max_shift,
                              base_image_seq_len=base_image_seq_len, max_image_seq_len=max_image_seq_len)
        self._init_tables()
        self.attn_map = None
        self.c: Optional[float] = None
        self.overshot_func: Optional[Callable] = None

    def set_c(self, c: float):
        self.c = c
        self._coef = None

    def set_overshot_func(self, overshot_func):
        self.overshot_func = overshot_func
        self._coef = None

    def set_attn_map(self, attn_map):
        if attn_map is not None:
            raise NotImplementedError("attn_map is a dead experiment in the reference: no live caller sets it (SURVEY §0.3)")
        self.attn_map = None

    def set_timesteps(self, num_inference_steps: int = None, device: Union[str, torch.device] = None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None):
        """scheduling_stochastic_rf_discrete_overshot.py:182-224 -- NO f32 cast before the shift (differs from Euler
        in the last ulp)."""
        c = self.config
        if c.use_dynamic_shifting and mu is None:
            raise ValueError(" you have a pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            self.num_inference_steps = num_inference_steps
            timesteps = np.linspace(self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min), num_inference_steps)
            sigmas = timesteps / c.num_train_timesteps
        else:
            sigmas = np.asarray(sigmas)
            self.num_inference_steps = len(sigmas)
        sigmas = self.time_shift(mu, 1.0, sigmas) if c.use_dynamic_shifting else c.shift * sigmas / (1 + (c.shift - 1) * sigmas)
        self._finish_set_timesteps(sigmas, device)

    def coef_table(self, device, state_dtype=torch.bfloat16) -> torch.Tensor:
        """Per step {t_over - t, a, b} (:306-349), evaluated on 0-dim f32 host tensors in the reference's order;
        (t_over - t) is rounded to the model-output dtype where `(t_overshoot - t) * (-model_output)` rounds it."""
        if self.c is None or self.overshot_func is None:
            raise RuntimeError("call set_c(...) and set_overshot_func(...) before stepping the AMO sampler")
        if self._coef is None or self._coef.device != torch.device(device):
            s = self._sigmas_host
            rows = []
            for i in range(len(s) - 1):
                sigma, sigma_next = s[i], s[i + 1]
                t = 1 - sigma
                step_size = sigma - sigma_next
                t_next = min(t + step_size, 1)
                t_over = min(self.overshot_func(t_next, step_size * self.c), 1)
                a = t_next / t_over
                b = ((1 - t_next) ** 2 - (a - t_next) ** 2) ** 0.5
                dt = torch.as_tensor(t_over - t, dtype=torch.float32).to(state_dtype).float()
                rows.append([float(dt), float(a), float(b)])
            self._coef = torch.tensor(rows, dtype=torch.float32).to(device)
        return self._coef

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True,
             noise: Optional[torch.Tensor] = None):
        """`noise=` (extra, optional) injects the eps the reference would have drawn from the global RNG
        (randn_tensor(..., generator=generator), :351-355) so trajectories can be replayed across devices."""
        self._check_timestep(timestep)
        if self.step_index is None:
            self._init_step_index(timestep)
        if noise is None:
            noise = torch.randn(sample.shape, generator=generator, device=sample.device, dtype=torch.float32)
        prev = sample.to(model_output.dtype).clone().contiguous()
        ops.amo_step_(model_output.contiguous(), prev, self.coef_table(sample.device, model_output.dtype),
                      noise.to(sample.device, torch.float32).contiguous(), step=self._step_index)
        sigma = self._sigmas_host[self._step_index].to(sample.device)
        predicted_x1 = sample.to(torch.float32) - sigma * model_output
        self._step_index += 1
        if not return_dict:
            return (prev, predicted_x1)
        return SimpleNamespace(prev_sample=prev, predicted_x1=predicted_x1)
