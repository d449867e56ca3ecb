"""Closed-loop evaluation harnesses -- counterparts of utils/rm_env_utils.py:18-221 (`EvalProc`,
`run_robomimic_eval`: `run_eval` below) and of utils/aloha_env_utils.py:28-163 (`process_aloha_obs`,
`run_aloha_eval_single`: `run_aloha_eval` below) for the MI355X agent.

robomimic protocol:

Same protocol, any environment: `n_proc` CPU worker processes each own one environment and run
`n_rollout / n_proc` episodes with seeds `seed + i * rollouts_per_proc + j` (rm_env_utils.py:107).
A worker sends `(process_id, [obs_t-H+1 .. obs_t])`, the parent stacks whatever workers are
waiting into ONE batch -- so the batch size changes call to call -- calls
`policy.sample_viz(dict(obs=...), rng)` once on the GPU, and answers each worker with
`(action[i],)` or `(action[i], plan_viz[i])`; the worker executes the `action_horizon` actions,
and ends an episode with the sentinel `[dict(reset=True)]` (rm_env_utils.py:53-84,150-199).
Worker exceptions arrive as a 3-tuple on the terminal queue and are re-raised in the parent
(rm_env_utils.py:93-94,120-126).

The simulator is injected: `env_factory(**env_kwargs)` must return an object with
`reset() -> obs dict`, `step(action) -> (obs dict, reward, done, info)` and
`is_success() -> {"task": bool, ...}`.  robosuite / dm_control are not available in this
environment; tests drive the harness with a deterministic fake environment.
"""
from __future__ import annotations

import os
import time
import traceback
from collections import deque
from typing import Callable, Dict, List, Optional

import numpy as np
import torch.multiprocessing as mp


def _worker(process_id, seeds, env_factory, env_params, send_q, recv_q, term_q):
    try:
        env = env_factory(**env_params.get("env_kwargs", {}))
        results = {}
        oh = env_params["obs_horizon"]
        viz_key = env_params.get("rgb_viz")
        for seed in seeds:
            np.random.seed(seed)
            ob = env.reset()
            success = {k: False for k in env.is_success()}
            total_reward, env_steps, frames = 0.0, 0, []
            obs_deque = deque([ob] * oh, maxlen=oh)
            while True:
                send_q.put((process_id, list(obs_deque)))
                out = recv_q.get()
                action = out[0]
                plan_viz = out[1] if len(out) == 2 else None
                r, done = 0.0, False
                for idx, ac in enumerate(action):
                    ob, r_step, done, _ = env.step(ac)
                    env_steps += 1
                    obs_deque.append(ob)
                    if viz_key is not None and viz_key in ob:
                        img = ob[viz_key]
                        if plan_viz is not None:
                            img = np.concatenate([np.transpose(img, (2, 0, 1)), plan_viz[idx]], axis=-1)
                        frames.append(img)
                    r += r_step
                    if done:
                        break
                total_reward += r
                cur = env.is_success()
                for k in success:
                    success[k] = success[k] or bool(cur[k])
                if done or success["task"]:
                    send_q.put((process_id, [dict(reset=True)]))
                    break
            results[seed] = dict(success=float(success["task"]), reward=float(total_reward), horizon=env_steps,
                                 debug_obs=frames)
        term_q.put((process_id, results))
        close = getattr(env, "close", None)
        if close:
            close()
    except Exception:                                                  # noqa: BLE001
        term_q.put((process_id, "error", traceback.format_exc()))


def _obs_keys_for_agent(env_kwargs) -> List[str]:
    rgb = [k[len("latent_"):] if k.startswith("latent_") else k for k in env_kwargs.get("rgb_obs", [])]
    return rgb + list(env_kwargs.get("lowdim_obs", []))


def run_eval(env_params: dict, policy, n_rollout: int, n_proc: int, seed: int, eval_rng: int,
             env_factory: Callable, visualize_plan: Optional[bool] = None, keep_latent_keys: bool = False,
             poll_s: float = 0.001, verbose: bool = False):
    """-> (rollout_logs, videos) like run_robomimic_eval.  `eval_rng` is an int; every policy call
    gets a fresh seed derived from it (the reference splits a JAX key per call)."""
    assert n_rollout % n_proc == 0
    per = n_rollout // n_proc
    ctx = mp.get_context("spawn")
    term_q = ctx.Queue()
    send_qs, recv_qs, procs = {}, {}, {}
    for i in range(n_proc):
        seeds = list(range(seed + i * per, seed + (i + 1) * per))
        send_qs[i], recv_qs[i] = ctx.Queue(), ctx.Queue()
        procs[i] = ctx.Process(target=_worker, args=(i, seeds, env_factory, env_params, send_qs[i], recv_qs[i], term_q),
                               daemon=True)
        procs[i].start()
    if visualize_plan is None:
        visualize_plan = policy.config.get("name") in ("ldp_agent", "ldp_hier_agent")
    agent_keys = None if keep_latent_keys else _obs_keys_for_agent(env_params.get("env_kwargs", {}))
    t0 = time.time()
    results: Dict[int, dict] = {}
    n_calls, batch_sizes = 0, []
    try:
        while procs:
            while not term_q.empty():
                out = term_q.get()
                if len(out) == 3:
                    raise RuntimeError(f"eval process {out[0]} failed:\n{out[2]}")
                idx, proc_results = out
                results.update(proc_results)
                procs[idx].join()
                procs.pop(idx); send_qs.pop(idx); recv_qs.pop(idx)
            dead = [i for i, p in procs.items() if not p.is_alive()]
            if dead and term_q.empty():
                time.sleep(0.05)                                      # results may still be in flight
                if term_q.empty():
                    raise RuntimeError(f"eval processes {dead} died (exit codes "
                                       f"{[procs[i].exitcode for i in dead]})")
                continue
            idxs, stacks = [], {}
            for i, q in send_qs.items():
                if q.empty():
                    continue
                pid, obs_deque = q.get()
                if "reset" in obs_deque[0] and obs_deque[0]["reset"] is True:
                    continue
                for k in obs_deque[0]:
                    stacks.setdefault(k, []).append(np.stack([o[k] for o in obs_deque]))
                idxs.append(pid)
            if not idxs:
                time.sleep(poll_s)
                continue
            obs = {k: np.asarray(v, dtype=np.float32) for k, v in stacks.items()}
            if agent_keys:
                if "optimal" in agent_keys and "optimal" not in obs:
                    ref = obs[env_params["env_kwargs"]["lowdim_obs"][0]]
                    obs["optimal"] = np.ones((ref.shape[0], 1, 1), dtype=ref.dtype)
                obs = {k: obs[k] for k in agent_keys}
            n_calls += 1
            batch_sizes.append(len(idxs))
            call_seed = (int(eval_rng) * 1000003 + n_calls) & 0x7FFFFFFFFFFFFFFF
            # the reference's call-site idiom, verbatim (utils/rm_env_utils.py:183-196)
            if visualize_plan:
                batch_action, plan_dict = policy.sample_viz(dict(obs=obs), call_seed)
                plan_viz = plan_dict["plan_viz"]
                pv = (np.clip((np.array(plan_viz) + 1) / 2, 0, 1) * 255).astype(np.uint8)
                action = np.array(batch_action)
            else:
                batch_action, _ = policy.sample(dict(obs=obs), call_seed)
                action = np.array(batch_action)
                pv = None
            for j, pid in enumerate(idxs):
                recv_qs[pid].put((action[j],) if pv is None else (action[j], pv[j]))
    finally:
        for p in procs.values():
            p.terminate()
    logs: Dict[str, list] = {}
    videos = []
    for res in results.values():
        for k, v in res.items():
            if k.startswith("debug"):
                videos.append(v)
            else:
                logs.setdefault(k, []).append(v)
    rollout_logs = {k: float(np.mean(v)) for k, v in logs.items()}
    rollout_logs["total_time"] = time.time() - t0
    rollout_logs["policy_calls"] = n_calls
    rollout_logs["mean_batch"] = float(np.mean(batch_sizes)) if batch_sizes else 0.0
    try:
        import psutil
        rollout_logs["RAM_MB"] = int(psutil.Process(os.getpid()).memory_info().rss / 1e6)
    except Exception:                                                  # noqa: BLE001
        pass
    if verbose:
        print(rollout_logs)
    return rollout_logs, videos


# ------------------------------------------------------------------------------------------------
# ALOHA: the reference's single-process protocol (utils/aloha_env_utils.py:28-163)
# ------------------------------------------------------------------------------------------------
def process_aloha_obs(ob_dict: dict, env_params: dict) -> dict:
    """utils/aloha_env_utils.py:28-46: lowdim keys pass through ('optimal' is a constant one), every camera named by `rgb_obs` /
    `rgb_viz` is taken from ob_dict['images'][<camera>] ('latent_wrist64_image' -> 'wrist64'), scaled to [0, 255] when it arrives in
    [0, 1], moved to HWC when it arrives channel-first, and stored under the key without its 'latent_' prefix."""
    new = {}
    for key in env_params["lowdim_obs"]:
        if key == "optimal":
            new["optimal"] = np.ones((1,), dtype=np.asarray(ob_dict[env_params["lowdim_obs"][0]]).dtype)
        else:
            new[key] = ob_dict[key]
    cams = list(env_params["rgb_obs"]) + ([env_params["rgb_viz"]] if env_params.get("rgb_viz") else [])
    for key in cams:
        img = np.asarray(ob_dict["images"][key.replace("_image", "").replace("latent_", "")])
        if img.min() > -0.01 and img.max() < 1.1:
            img = img * 255
        assert img.min() >= 0 and 50 < img.max() < 257, f"min: {img.min()} max: {img.max()}"
        if img.shape[-1] != 3:
            img = np.moveaxis(img, -3, -1)                     # ... C H W -> ... H W C
        assert img.shape[-1] == 3
        new[key.replace("latent_", "")] = img
    return new


def run_aloha_eval(env_params: dict, policy, n_rollout: int, seed: int, eval_rng: int, env_factory: Callable,
                   episode_len: int = 400, reset_hook: Optional[Callable] = None, verbose: bool = False):
    """Counterpart of `run_aloha_eval_single` (utils/aloha_env_utils.py:51-163): ONE process, one environment, one plan per policy
    call (B = 1), `action_horizon` environment steps per call.  The simulator is injected: `env_factory(**env_kwargs)` returns a
    dm_control-style object -- `reset() -> ts`, `step(action) -> ts` with `ts.observation` (lowdim keys + 'images': {camera: HWC})
    and `ts.reward`, and `task.max_reward`.  `reset_hook(i, env_params)` stands for the reference's object-pose sampling
    (`BOX_POSE[0] = sample_box_pose()`, :64-69), called after `np.random.seed(seed + 100 + i)` like there.
    The policy call site is the reference's, verbatim (:91-96): `sample_viz(dict(obs=obs_dict), rng)` when the policy is an
    'ldp_agent', its `plan_viz` turned into uint8 HWC frames by `((plan_viz + 1) / 2 * 255).astype(np.uint8).transpose(0, 1, 3, 4, 2)`
    and pasted next to the `rgb_viz` camera frame of every executed step; otherwise `sample`, the scene cameras ('top' | 'angle' | 'vis') as the debug
    frames and the call's scalar metrics folded into the rollout's log (min over keys containing 'min', max over the rest: :97-105, 118-120, 151-152).
    -> (rollout_logs, videos)."""
    env = env_factory(**env_params.get("env_kwargs", {}))
    max_reward = env.task.max_reward
    oh = env_params["obs_horizon"]
    t0 = time.time()
    results = {}
    n_calls = 0
    ob_stats: Dict[str, float] = {}                                # (one dict over ALL rollouts, like the reference's ob_dict_stats: :60)
    for i in range(n_rollout):
        np.random.seed(seed + 100 + i)                         # + 100: the training data was collected from seeds [0, 49] (:63)
        if reset_hook is not None:
            reset_hook(i, env_params)
        ts = env.reset()
        ob = process_aloha_obs(ts.observation, env_params)
        frames, total_reward, env_steps = [], 0.0, 0
        obs_deque = deque([ob] * oh, maxlen=oh)
        done = False
        while True:
            obs_dict = {k: np.array([np.stack([x[k] for x in obs_deque])], dtype=np.float32) for k in obs_deque[0]}     # (1, H, ...)
            n_calls += 1
            call_seed = (int(eval_rng) * 1000003 + n_calls) & 0x7FFFFFFFFFFFFFFF
            visualize_plan = policy.config["name"] == "ldp_agent"
            if visualize_plan:
                action, plan_dict = policy.sample_viz(dict(obs=obs_dict), call_seed)
                plan_viz = ((plan_dict["plan_viz"] + 1) / 2 * 255).astype(np.uint8).transpose(0, 1, 3, 4, 2)
            else:                                                   # :97-105: scalar metrics of the call, min over '*min*' keys, max over the others
                action, metrics = policy.sample(dict(obs=obs_dict), call_seed)
                for k, v in (metrics or {}).items():
                    if np.ndim(v) != 0:
                        continue                                        # (the reference's other agents return scalars only; this package's 'plan' array is not a statistic)
                    v = float(v)
                    ob_stats[k] = (min(ob_stats[k], v) if "min" in k else max(ob_stats[k], v)) if k in ob_stats else v
            action = np.array(action)
            for idx, ac in enumerate(action[0]):
                try:
                    ts = env.step(ac)
                except Exception:                                   # noqa: BLE001  (the reference: "broke", :107-110)
                    done = True
                ob = process_aloha_obs(ts.observation, env_params)
                obs_deque.append(ob)
                if visualize_plan and env_params.get("rgb_viz"):
                    frames.append(np.concatenate([ob[env_params["rgb_viz"]], plan_viz[0, idx]], axis=1))
                elif not visualize_plan:                            # :118-120: the three scene cameras side by side
                    images = ts.observation.get("images", {}) if isinstance(ts.observation, dict) else {}
                    if all(k in images for k in ("top", "angle", "vis")):
                        frames.append(np.concatenate([images[k] for k in ("top", "angle", "vis")], axis=1))
                total_reward += ts.reward
                env_steps += 1
                done = ts.reward == max_reward
                if env_steps > episode_len:
                    done = True
                    break
                if done:
                    break
            if done or ts.reward == max_reward:
                break
        results[i] = dict(success=float(ts.reward == max_reward), reward=float(total_reward), horizon=env_steps,
                          avg_reward=float(total_reward / max(env_steps, 1)), debug_obs=frames)
        if verbose:
            print(f"{i}: {results[i]['success']} ({results[i]['avg_reward']}/{max_reward})")
    logs: Dict[str, list] = {}
    videos = []
    for res in results.values():
        for k, v in res.items():
            if k.startswith("debug"):
                videos.append(v)
            else:
                logs.setdefault(k, []).append(v)
    rollout_logs = {k: float(np.mean(v)) for k, v in logs.items()}
    rollout_logs["total_time"] = time.time() - t0
    rollout_logs["policy_calls"] = n_calls
    try:
        import psutil
        rollout_logs["RAM_MB"] = int(psutil.Process(os.getpid()).memory_info().rss / 1e6)
        rollout_logs["RAM_GB"] = float(rollout_logs["RAM_MB"] / 1000)
    except Exception:                                                  # noqa: BLE001
        pass
    for k, v in ob_stats.items():                                  # :157-158
        rollout_logs[k] = np.array(v)
    return rollout_logs, videos


def eval_loss_metrics(policy, batch: dict, rng) -> dict:
    """The sampling metrics of eval_bc.py:128-151 on one held-out batch {'obs': (B,H,...), 'actions':
    (B,H,A)}, by the reference's own definitions (note: the targets are the RAW batch actions):
        pred_action      = agent.sample_action(batch, rng)        # IDM on the ground-truth plan, (B, H-1, A)
        action_mse       = mean((actions[:, :H'] - pred_action[:, :H'])**2),  H' = pred_action.shape[1]
        action_mse_{0,1,2}      = the same at time index 0 / 1 / 2 (1 and 2 skipped when out of range, like
                                  the reference's try/except)
        pred_action_full = agent.sample(batch, rng)[0]            # planner + IDM, (B, action_horizon, A)
        full_action_mse, full_action_mse_{0,1,2}: as above with pred_action_full
        plan_mse         = stats['plan_mse']
    preceded, as at eval_bc.py:127, by `metrics = agent.get_metrics(batch, rng)` (the two training losses, forward only, and the
    statistics scalars of agent/ldp_agent.py:141-180): the sampling metrics are added to THAT dict.  A policy without get_metrics
    (or one that raises NotImplementedError: LDPHierAgent, like the reference's early return at :108-109) contributes none.
    One rng is used for both samplers, as the reference passes `sample_rng` to both (:134,143)."""
    cfg = getattr(policy, "config", None)
    if cfg is not None and cfg.get("name") == "ldp_hier_agent":          # eval_bc.py:107-109: `return dict(), eval_rng`
        return {}
    use_planner = bool(getattr(policy, "use_planner", True))
    actions = np.asarray(batch["actions"], dtype=np.float32)
    metrics = {}
    gm = getattr(policy, "get_metrics", None)
    if gm is not None:
        try:
            metrics.update({k: float(v) for k, v in gm(batch, rng).items()})
        except NotImplementedError:
            pass

    def mse(a, b):
        return float(np.mean(np.square(a - b)))

    def per_index(prefix, pred):
        H = pred.shape[1]
        pred = np.array(pred)
        metrics[prefix] = mse(actions[:, :H, :], pred[:, :H, :])
        metrics[f"{prefix}_0"] = mse(actions[:, 0, :], pred[:, 0, :])
        for i in (1, 2):
            if i < H and i < actions.shape[1]:
                metrics[f"{prefix}_{i}"] = mse(actions[:, i, :], pred[:, i, :])

    pred_action = policy.sample_action(batch, rng)
    per_index("action_mse", pred_action)
    if use_planner:
        pred_action_full, stats = policy.sample(batch, rng)
        per_index("full_action_mse", pred_action_full)
        metrics["plan_mse"] = float(stats["plan_mse"])
    return metrics
