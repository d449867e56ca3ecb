"""ORACLE (test infrastructure, NOT a product path) -- the TRAINING step of the reference agent.

PARITY UNPINNED like the rest of oracle/ (no executable reference here: jax / flax / optax are not installable).

What is restated, and from where:
  * the two losses and their sum               agent/ldp_agent.py:113-180 (`plan_loss`, `idm_loss`, `loss`)
  * `jax.grad(self.loss, has_aux=True)`        agent/ldp_agent.py:252 -- DEFINED here as torch autograd in float64 over the forward
                                                restatement of oracle/torch32.py (which tests/test_oracle_kats.py ties to oracle/np64.py)
  * `linear_algebra.global_norm(grads)`        agent/ldp_agent.py:253 (optax: sqrt of the sum of squares over every leaf)
  * `TrainState.apply_gradients`               flax 0.8.4 train_state.py: updates, opt_state = tx.update(grads, opt_state, params);
                                                params = optax.apply_updates(params, updates); step += 1
  * `optax.adam(lr_schedule)`                  optax 0.2.2 (env.yml:28) alias.py: chain(scale_by_adam(b1=0.9, b2=0.999, eps=1e-8, eps_root=0),
                                                scale_by_learning_rate(schedule)); transform.py scale_by_adam:
                                                    mu = (1 - b1) * g + b1 * mu;  nu = (1 - b2) * g**2 + b2 * nu;  count += 1
                                                    mu_hat = mu / (1 - b1**count);  nu_hat = nu / (1 - b2**count)
                                                    u = mu_hat / (sqrt(nu_hat + eps_root) + eps)
                                                scale_by_schedule: step_size = -schedule(count_before_increment); updates = step_size * u
  * `optax.warmup_cosine_decay_schedule`       optax 0.2.2 schedules/_schedule.py (agent/ldp_agent.py:583-589, 621-627):
                                                join_schedules([linear_schedule(init, peak, warmup_steps),
                                                                cosine_decay_schedule(peak, decay_steps - warmup_steps, alpha = end / peak)], [warmup_steps])
  * the metrics dict of `update_step`          agent/ldp_agent.py:239-272 (g_norm, planner_lr / idm_lr from `self.lr_schedule` -- the LAST schedule
                                                `create` built, i.e. the IDM's when use_idm (:621-627,669) --, planner_step / idm_step = the state's
                                                step BEFORE the update, noise_diff = 0 only on the no-planner branch)
`update` / `update_mixed` gating (agent/ldp_agent.py:223-232, 274-283) is restated in `update_gates`.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import torch32
from .np64 import ddpm_tables

F64 = np.float64


class GradParams(torch32.TorchParams):
    """TorchParams whose leaves are float64 autograd leaves in the Flax layout (gradients come back in that layout)."""

    def __init__(self, params: Dict[str, np.ndarray]):
        super().__init__(params, dtype=torch.float64)
        self.leaves = OrderedDict((k, torch.tensor(np.asarray(v, dtype=F64), dtype=torch.float64, requires_grad=True))
                                  for k, v in params.items())

    def t(self, key):
        return self.leaves[key]

    def grads(self) -> "OrderedDict[str, np.ndarray]":
        return OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy()) for k, v in self.leaves.items())


def _add_noise(x0: torch.Tensor, noise: torch.Tensor, t: np.ndarray, n_train: int) -> torch.Tensor:
    """FlaxDDPMScheduler.add_noise on the float32 abar table (agent/ldp_agent.py:119,136)."""
    acp = torch.tensor(np.asarray(ddpm_tables(n_train)[2], F64))[torch.as_tensor(np.asarray(t).reshape(-1).astype(np.int64))]
    shape = (-1,) + (1,) * (x0.dim() - 1)
    return acp.sqrt().reshape(shape) * x0 + (1 - acp).sqrt().reshape(shape) * noise


def plan_loss(P: torch32.TorchParams, obs_emb: torch.Tensor, t, noise, obs_horizon: int, n_train: int, **unet_kw) -> torch.Tensor:
    """agent/ldp_agent.py:113-127 with the timesteps / noise given explicitly (the reference draws them from its key)."""
    nxt = obs_emb[:, obs_horizon:]
    noise = torch.as_tensor(np.asarray(noise, F64))
    noisy = _add_noise(nxt, noise, t, n_train)
    cond = obs_emb[:, :obs_horizon].reshape(obs_emb.shape[0], -1)
    pred = torch32.unet_forward(P, noisy, torch.as_tensor(np.asarray(t).reshape(-1)), cond, **unet_kw)
    return ((pred - noise) ** 2).mean()


def idm_loss(P: torch32.TorchParams, obs_emb: torch.Tensor, actions: torch.Tensor, t, noise, obs_horizon: int, n_train: int) -> torch.Tensor:
    """agent/ldp_agent.py:129-140."""
    s = torch.cat([obs_emb[:, obs_horizon - 1:-1], obs_emb[:, obs_horizon:]], dim=-1)
    s = s.reshape(-1, s.shape[-1])                                  # 'B H D -> (B H) D'
    a = actions[:, :-1].reshape(-1, actions.shape[-1])
    noise = torch.as_tensor(np.asarray(noise, F64))
    noisy = _add_noise(a, noise, np.asarray(t).reshape(-1), n_train)
    pred = torch32.idm_forward(P, s, noisy, torch.as_tensor(np.asarray(t).reshape(-1)))
    return ((pred - noise) ** 2).mean()


def hier_plan_loss(P, obs_emb, t, noise, obs_horizon: int, idm_horizon: int, n_train: int, **unet_kw) -> torch.Tensor:
    """agent/ldp_hier_agent.py:111-123: the planner denoises every `idm_horizon`-th future state."""
    nxt = obs_emb[:, obs_horizon::idm_horizon]
    noise = torch.as_tensor(np.asarray(noise, F64))
    noisy = _add_noise(nxt, noise, t, n_train)
    cond = obs_emb[:, :obs_horizon].reshape(obs_emb.shape[0], -1)
    pred = torch32.unet_forward(P, noisy, torch.as_tensor(np.asarray(t).reshape(-1)), cond, **unet_kw)
    return ((pred - noise) ** 2).mean()


def hier_idm_loss(P, obs_emb, actions, t, noise, obs_horizon: int, idm_horizon: int, n_train: int, **unet_kw) -> torch.Tensor:
    """agent/ldp_hier_agent.py:125-137: the IDM is a ConditionalUnet1D over chunks of `idm_horizon` actions, conditioned on (state, state + idm_horizon)."""
    oh, ih = obs_horizon, idm_horizon
    s = torch.cat([obs_emb[:, oh - 1:-1:ih], obs_emb[:, oh - 1 + ih::ih]], dim=-1)
    s = s.reshape(-1, s.shape[-1])                                  # 'B H D -> (B H) D'
    a = actions[:, oh - 1:-1]
    a = a.reshape(a.shape[0], -1, ih, a.shape[-1]).reshape(-1, ih, a.shape[-1])     # 'B K H D -> (B K) H D'
    noise = torch.as_tensor(np.asarray(noise, F64))
    noisy = _add_noise(a, noise, np.asarray(t).reshape(-1), n_train)
    pred = torch32.unet_forward(P, noisy, torch.as_tensor(np.asarray(t).reshape(-1)), s, **unet_kw)
    return ((pred - noise) ** 2).mean()


def loss_and_grads(planner_params, idm_params, obs_emb, actions, *, t_plan=None, noise_plan=None, t_idm=None, noise_idm=None,
                   idm_obs_emb=None, idm_actions=None, obs_horizon=1, n_train_planner=100, n_train_idm=100, alpha_planner=1.0,
                   alpha_idm=1.0, idm_horizon=None, idm_unet_kw=None, **unet_kw):
    """`jax.grad(self.loss, has_aux=True)(combined_params, ...)` (agent/ldp_agent.py:141-160, 252): float64 autograd.
    planner_params / idm_params = None leaves that module out (use_planner / use_idm False).  idm_obs_emb / idm_actions: the mixed batch of
    `loss_mixed` (:182-203); default = the same batch.  idm_horizon: the hierarchical agent's losses (agent/ldp_hier_agent.py:111-160: strided
    planner targets, a U-Net IDM over action chunks).  -> dict(plan_loss, idm_loss, loss, grads_planner, grads_idm, g_norm)."""
    emb = torch.as_tensor(np.asarray(obs_emb, F64))
    act = torch.as_tensor(np.asarray(actions, F64))
    emb_i = emb if idm_obs_emb is None else torch.as_tensor(np.asarray(idm_obs_emb, F64))
    act_i = act if idm_actions is None else torch.as_tensor(np.asarray(idm_actions, F64))
    out = dict(plan_loss=0.0, idm_loss=0.0, grads_planner=None, grads_idm=None)
    total = None
    PP = PI = None
    if planner_params is not None:
        PP = GradParams(planner_params)
        lp = alpha_planner * (plan_loss(PP, emb, t_plan, noise_plan, obs_horizon, n_train_planner, **unet_kw) if idm_horizon is None else
                              hier_plan_loss(PP, emb, t_plan, noise_plan, obs_horizon, idm_horizon, n_train_planner, **unet_kw))
        out["plan_loss"] = float(lp.detach())
        total = lp
    if idm_params is not None:
        PI = GradParams(idm_params)
        li = alpha_idm * (idm_loss(PI, emb_i, act_i, t_idm, noise_idm, obs_horizon, n_train_idm) if idm_horizon is None else
                          hier_idm_loss(PI, emb_i, act_i, t_idm, noise_idm, obs_horizon, idm_horizon, n_train_idm, **(idm_unet_kw or {})))
        out["idm_loss"] = float(li.detach())
        total = li if total is None else total + li
    out["loss"] = out["plan_loss"] + out["idm_loss"]
    sq = 0.0
    if total is not None:
        total.backward()
    if PP is not None:
        out["grads_planner"] = PP.grads()
        sq += sum(float((g ** 2).sum()) for g in out["grads_planner"].values())
    if PI is not None:
        out["grads_idm"] = PI.grads()
        sq += sum(float((g ** 2).sum()) for g in out["grads_idm"].values())
    out["g_norm"] = math.sqrt(sq)
    return out


# ------------------------------------------------------------------------------------------------ optax restatements
def warmup_cosine_decay_schedule(init_value, peak_value, warmup_steps, decay_steps, end_value, exponent=1.0):
    """optax 0.2.2 warmup_cosine_decay_schedule -> f(count) in float64 (the traced original evaluates it in float32: agreement to ~1e-7 relative)."""
    alpha = 0.0 if peak_value == 0.0 else end_value / peak_value
    cos_steps = decay_steps - warmup_steps

    def linear(count):                                    # polynomial_schedule(power=1, transition_begin=0)
        if warmup_steps <= 0:
            return peak_value
        c = min(max(count, 0), warmup_steps)
        frac = 1.0 - c / warmup_steps
        return (init_value - peak_value) * frac + peak_value

    def cosine(count):                                    # cosine_decay_schedule(init_value=peak, decay_steps=cos_steps, alpha)
        if cos_steps <= 0:
            raise ValueError("The cosine_decay_schedule requires positive decay_steps!")
        c = min(count, cos_steps)
        decayed = (1 - alpha) * (0.5 * (1 + math.cos(math.pi * c / cos_steps))) ** exponent + alpha
        return peak_value * decayed

    def schedule(count):                                  # join_schedules(..., boundaries=[warmup_steps])
        count = int(count)
        return linear(count) if count < warmup_steps else cosine(count - warmup_steps)
    return schedule


def adam_init(params) -> dict:
    return dict(mu=OrderedDict((k, np.zeros_like(np.asarray(v, F64))) for k, v in params.items()),
                nu=OrderedDict((k, np.zeros_like(np.asarray(v, F64))) for k, v in params.items()), count=0)


def adam_apply(params, grads, state, lr_schedule, b1=0.9, b2=0.999, eps=1e-8):
    """One `TrainState.apply_gradients` with tx = optax.adam(lr_schedule).  -> (new params, new state), float64."""
    count_inc = state["count"] + 1
    step_size = -lr_schedule(state["count"])              # scale_by_schedule reads the count BEFORE the increment
    bc1, bc2 = 1.0 - b1 ** count_inc, 1.0 - b2 ** count_inc
    mu, nu, new = OrderedDict(), OrderedDict(), OrderedDict()
    for k, p in params.items():
        g = np.asarray(grads[k], F64)
        m = (1 - b1) * g + b1 * state["mu"][k]
        v = (1 - b2) * (g * g) + b2 * state["nu"][k]
        u = (m / bc1) / (np.sqrt(v / bc2) + eps)
        mu[k], nu[k] = m, v
        new[k] = np.asarray(p, F64) + step_size * u
    return new, dict(mu=mu, nu=nu, count=count_inc)


def update_gates(cfg: dict, use_planner: bool, use_idm: bool, step: int):
    """The python-side gating of `update` / `update_mixed` (agent/ldp_agent.py:223-232) -> (use_planner, use_idm) of this step."""
    up = bool(use_planner) and step % cfg["update_planner_every"] == 0
    ui = bool(use_idm) and step % cfg["update_idm_every"] == 0
    ui = ui and step >= cfg["update_idm_after"]
    upd = cfg["update_planner_until"] < 0 or step < cfg["update_planner_until"]
    upd = upd and step >= cfg["update_planner_after"]
    return up and upd, ui


class TrainOracle:
    """The state the reference's agent carries through `update` (two TrainStates) plus one `update_step` (agent/ldp_agent.py:239-272)."""

    def __init__(self, planner_params, idm_params, *, obs_horizon=1, n_train_planner=100, n_train_idm=100, alpha_planner=1.0, alpha_idm=1.0,
                 lr=1e-4, end_lr=1e-6, idm_lr=1e-4, idm_end_lr=1e-6, warmup_steps=1000, decay_steps=500000, idm_horizon=None, idm_unet_kw=None, **unet_kw):
        self.pp = None if planner_params is None else OrderedDict((k, np.asarray(v, F64)) for k, v in planner_params.items())
        self.ip = None if idm_params is None else OrderedDict((k, np.asarray(v, F64)) for k, v in idm_params.items())
        self.p_state = None if self.pp is None else adam_init(self.pp)
        self.i_state = None if self.ip is None else adam_init(self.ip)
        self.p_sched = warmup_cosine_decay_schedule(end_lr, lr, warmup_steps, decay_steps, end_lr)
        self.i_sched = warmup_cosine_decay_schedule(idm_end_lr, idm_lr, warmup_steps, decay_steps, idm_end_lr)
        # `self.lr_schedule` of the reference object is the variable `create` assigned LAST (agent/ldp_agent.py:669): the IDM's if use_idm
        self.reported_sched = self.i_sched if self.ip is not None else self.p_sched
        self.kw = dict(obs_horizon=obs_horizon, n_train_planner=n_train_planner, n_train_idm=n_train_idm, alpha_planner=alpha_planner,
                       alpha_idm=alpha_idm, idm_horizon=idm_horizon, idm_unet_kw=idm_unet_kw, **unet_kw)

    def update_step(self, obs_emb, actions, *, use_planner=True, use_idm=True, **noise):
        """-> metrics (the update_step additions: g_norm, *_lr, *_step [, noise_diff]) + plan_loss / idm_loss / loss; the state advances."""
        r = loss_and_grads(self.pp if use_planner else None, self.ip if use_idm else None, obs_emb, actions, **noise, **self.kw)
        m = dict(plan_loss=r["plan_loss"], idm_loss=r["idm_loss"], loss=r["loss"], g_norm=r["g_norm"])
        if use_planner:
            m["planner_lr"] = self.reported_sched(self.p_state["count"])
            m["planner_step"] = self.p_state["count"]
            self.pp, self.p_state = adam_apply(self.pp, r["grads_planner"], self.p_state, self.p_sched)
        else:
            m.update(planner_lr=0, planner_step=0, noise_diff=0)
        if use_idm:
            m["idm_lr"] = self.reported_sched(self.i_state["count"])
            m["idm_step"] = self.i_state["count"]
            self.ip, self.i_state = adam_apply(self.ip, r["grads_idm"], self.i_state, self.i_sched)
        else:
            m.update(idm_lr=0, idm_step=0)
        self.last = r
        return m
