"""Noise-schedule and timestep-embedding constant tables (host side, numpy).

These are weight-independent constants, built once on the host exactly the way the
reference builds its scheduler state (python float64 loop -> float32 arrays) and then
uploaded to HBM; the kernels only index them.

Reference anchors:
  * FlaxDDPMScheduler(num_train_timesteps=100, beta_schedule='squaredcos_cap_v2',
    clip_sample=True, prediction_type='epsilon')      agent/ldp_agent.py:637-650
    (diffusers==0.27.2, restated in SURVEY.md Appendix A.2)
  * SinusoidalPosEmb                                    networks/diffusion_nets_v2.py:21-31
  * FourierFeatures(learnable=False)                    networks/diffusion.py:7-22
  * DDIM (eta=0) is defined by this repository; the reference has no DDIM.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SAMPLER_DDPM = 0
SAMPLER_DDIM = 1


def betas_squaredcos_cap_v2(n: int, max_beta: float = 0.999) -> np.ndarray:
    def alpha_bar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2.0) ** 2
    b = [min(1.0 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)]
    return np.asarray(b, dtype=np.float32)


@dataclass(frozen=True)
class NoiseSchedule:
    betas: np.ndarray            # (n,) f32
    alphas: np.ndarray           # (n,) f32
    alphas_cumprod: np.ndarray   # (n,) f32

    @property
    def n(self) -> int:
        return int(self.betas.shape[0])


def make_schedule(n_train: int = 100) -> NoiseSchedule:
    betas = betas_squaredcos_cap_v2(n_train)
    alphas = (np.float32(1.0) - betas).astype(np.float32)
    ac = np.cumprod(alphas, dtype=np.float32)
    return NoiseSchedule(betas, alphas, ac)


# Per-step coefficient rows consumed by the fused sampler epilogue.  One row per *executed*
# step i (i = 0 is the first, noisiest step).  Row layout (8 floats):
#   0: t (timestep index fed to the eps-model, as float)      1: 1/sqrt(abar_t)
#   2: sqrt(1-abar_t)                                          3: c_x0  (weight of clipped x0)
#   4: c_x   (weight of x_t; 0 for DDIM)                       5: c_eps (weight of eps; 0 for DDPM)
#   6: sigma (noise scale; 0 at t==0 and for DDIM)             7: unused
COEF_STRIDE = 8


def step_timesteps(n_train: int, n_steps: int, sampler: int) -> np.ndarray:
    """Timesteps visited, first executed step first."""
    if sampler == SAMPLER_DDPM:
        if n_steps != n_train:
            raise ValueError("the reference DDPM sampler visits every training timestep "
                             f"(n_steps must equal {n_train}, got {n_steps})")
        return np.arange(n_train - 1, -1, -1, dtype=np.int64)
    if n_train % n_steps != 0:
        raise ValueError(f"DDIM needs n_steps | n_train ({n_steps} vs {n_train})")
    stride = n_train // n_steps
    return (np.arange(n_steps - 1, -1, -1, dtype=np.int64) * stride)


def step_coefficients(sched: NoiseSchedule, n_steps: int, sampler: int) -> np.ndarray:
    ts = step_timesteps(sched.n, n_steps, sampler)
    ac = sched.alphas_cumprod.astype(np.float64)
    out = np.zeros((len(ts), COEF_STRIDE), dtype=np.float64)
    stride = sched.n // n_steps
    for i, t in enumerate(ts):
        a_t = ac[t]
        out[i, 0] = float(t)
        out[i, 1] = 1.0 / math.sqrt(a_t)
        out[i, 2] = math.sqrt(1.0 - a_t)
        if sampler == SAMPLER_DDPM:
            a_prev = ac[t - 1] if t > 0 else 1.0
            beta = float(sched.betas[t])
            alpha = float(sched.alphas[t])
            out[i, 3] = math.sqrt(a_prev) * beta / (1.0 - a_t)
            out[i, 4] = math.sqrt(alpha) * (1.0 - a_prev) / (1.0 - a_t)
            var = max((1.0 - a_prev) / (1.0 - a_t) * beta, 1e-20)
            out[i, 6] = math.sqrt(var) if t > 0 else 0.0
        else:
            tp = t - stride
            a_prev = ac[tp] if tp >= 0 else 1.0
            out[i, 3] = math.sqrt(a_prev)
            out[i, 5] = math.sqrt(1.0 - a_prev)
    return out.astype(np.float32)


def _freqs(dim: int) -> np.ndarray:
    """exp(-j * ln(1e4)/(half-1)), evaluated in float32 like the reference's traced graph."""
    half = dim // 2
    step = np.float32(np.log(np.float32(10000.0))) / np.float32(half - 1)
    return np.exp(np.arange(half, dtype=np.float32) * -step).astype(np.float32)


def sinusoidal_table(n: int, dim: int, cos_first: bool) -> np.ndarray:
    """(n, dim) rows for timesteps 0..n-1.  cos_first=False: planner [sin | cos];
    cos_first=True: IDM FourierFeatures [cos | sin]."""
    f = _freqs(dim)
    arg = (np.arange(n, dtype=np.float32)[:, None] * f[None, :]).astype(np.float32).astype(np.float64)
    s, c = np.sin(arg), np.cos(arg)
    tab = np.concatenate([c, s], -1) if cos_first else np.concatenate([s, c], -1)
    return tab.astype(np.float32)


# ---- learning-rate schedule of the training step ---------------------------------------------------------------------------------
def warmup_cosine_decay_schedule(init_value: float, peak_value: float, warmup_steps: int, decay_steps: int, end_value: float,
                                 exponent: float = 1.0):
    """optax 0.2.2 `warmup_cosine_decay_schedule` (the reference's `lr_schedule`, agent/ldp_agent.py:583-589, 621-627) as a host function
    count -> learning rate: join_schedules([linear_schedule(init, peak, warmup_steps), cosine_decay_schedule(peak, decay_steps -
    warmup_steps, alpha = end / peak, exponent)], [warmup_steps]).  optax evaluates it in float32 inside the traced step; here float64 on the
    host, rounded to float32 when it is handed to the Adam kernel."""
    alpha = 0.0 if peak_value == 0.0 else end_value / peak_value
    cos_steps = decay_steps - warmup_steps
    if cos_steps <= 0:
        raise ValueError("The cosine_decay_schedule requires positive decay_steps!")          # (optax's own message)

    def schedule(count) -> float:
        count = int(count)
        if count < warmup_steps:                                       # linear_schedule = polynomial_schedule(power=1)
            frac = 1.0 - min(max(count, 0), warmup_steps) / warmup_steps
            return (init_value - peak_value) * frac + peak_value
        c = min(count - warmup_steps, cos_steps)
        decayed = (1.0 - alpha) * (0.5 * (1.0 + math.cos(math.pi * c / cos_steps))) ** exponent + alpha
        return peak_value * decayed
    return schedule
