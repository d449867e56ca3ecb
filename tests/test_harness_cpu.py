"""Eval-harness protocol on CPU with a fake environment and a fake policy."""
import numpy as np
import pytest

from latent_diffusion_planning_amd.harness import run_eval
from tests.fake_env import FakePolicy, make_env

ENV = dict(obs_horizon=2, rgb_viz=None,
           env_kwargs=dict(lowdim_obs=["robot0_eef_pos"], rgb_obs=["latent_agentview_image"], horizon=24,
                           keep="x"))


def _raise_env(**kw):
    raise RuntimeError("boom in worker")


def test_rollouts_batches_and_logs():
    pol = FakePolicy()
    env = dict(ENV, env_kwargs=dict(ENV["env_kwargs"]))
    env["env_kwargs"].pop("keep")
    logs, videos = run_eval(env, pol, n_rollout=6, n_proc=3, seed=10, eval_rng=1, env_factory=make_env,
                            keep_latent_keys=True)
    assert logs["success"] == 1.0                      # constant 0.5 actions always reach the goal
    assert 4 <= logs["horizon"] <= 24 and logs["reward"] >= 1.0
    assert logs["policy_calls"] == len(pol.batches) and max(pol.batches) <= 3 and min(pol.batches) >= 1
    assert len(videos) == 6 and "total_time" in logs


def test_worker_failure_is_raised_in_the_parent():
    with pytest.raises(RuntimeError, match="boom in worker"):
        run_eval(dict(ENV, env_kwargs={}), FakePolicy(), 2, 2, 0, 0, env_factory=_raise_env, keep_latent_keys=True)


def test_rollouts_must_divide():
    with pytest.raises(AssertionError):
        run_eval(ENV, FakePolicy(), 5, 2, 0, 0, env_factory=make_env)


def test_aloha_single_process_protocol():
    """utils/aloha_env_utils.py:51-163 on a fake dm_control-style env: seeds seed + 100 + i, obs stacked to (1, H, ...), camera frames
    scaled from [0, 1] and moved to HWC, action_horizon steps per call, success = reward reaching task.max_reward."""
    from latent_diffusion_planning_amd.harness import process_aloha_obs, run_aloha_eval
    from tests.fake_env import make_aloha_env
    env_params = dict(obs_horizon=1, lowdim_obs=["qpos", "optimal"], rgb_obs=["latent_wrist64_image"], rgb_viz="top_image",
                      env_kwargs=dict(task_name="sim_transfer_cube", horizon=12))
    env = make_aloha_env()
    np.random.seed(0)
    ob = process_aloha_obs(env.reset().observation, env_params)
    assert set(ob) == {"qpos", "optimal", "wrist64_image", "top_image"}
    assert ob["wrist64_image"].shape == (64, 64, 3) and 50 < ob["wrist64_image"].max() <= 255 and ob["optimal"].shape == (1,)

    seen = []

    class Pol(FakePolicy):
        def sample(self, batch, rng):
            seen.append({k: v.shape for k, v in batch["obs"].items()})
            return FakePolicy.sample(self, batch, rng)
    hooks = []
    logs, videos = run_aloha_eval(env_params, Pol(), n_rollout=3, seed=5, eval_rng=2, env_factory=make_aloha_env,
                                  reset_hook=lambda i, ep: hooks.append(i))
    assert hooks == [0, 1, 2] and logs["success"] == 1.0 and logs["policy_calls"] == len(seen) and len(videos) == 3
    assert seen[0]["wrist64_image"] == (1, 1, 64, 64, 3) and seen[0]["qpos"] == (1, 1, 14)
    assert 1 <= logs["horizon"] <= 13 and logs["reward"] >= 4


def test_aloha_other_agents_get_scene_frames_and_folded_metrics():
    """The reference's non-'ldp_agent' branch (utils/aloha_env_utils.py:97-105, 118-120, 151-158): `sample`, the three scene cameras
    side by side as the debug frames, and the calls' scalar metrics folded over ALL rollouts -- min for keys containing 'min', max otherwise."""
    from latent_diffusion_planning_amd.harness import run_aloha_eval
    from tests.fake_env import FakeAlohaEnv

    class SceneEnv(FakeAlohaEnv):
        def _ts(self):
            ts = FakeAlohaEnv._ts(self)
            for cam, val in (("top", 60), ("angle", 120), ("vis", 180)):
                ts.observation["images"][cam] = np.full((8, 6, 3), val, dtype=np.uint8)
            return ts
    calls = []

    class Pol(FakePolicy):
        config = dict(FakePolicy.config, name="dp_agent")

        def sample(self, batch, rng):
            a, _ = FakePolicy.sample(self, batch, rng)
            calls.append(len(calls))
            return a, {"plan_min": 5.0 - len(calls), "plan_max": float(len(calls)), "plan": np.zeros((1, 5, 3))}
    env_params = dict(obs_horizon=1, lowdim_obs=["qpos"], rgb_obs=["latent_wrist64_image"], rgb_viz="top_image",
                      env_kwargs=dict(task_name="sim_transfer_cube", horizon=12))
    logs, videos = run_aloha_eval(env_params, Pol(), n_rollout=2, seed=5, eval_rng=2, env_factory=lambda **kw: SceneEnv(**kw))
    n = len(calls)
    assert n >= 2 and float(logs["plan_min"]) == 5.0 - n and float(logs["plan_max"]) == float(n) and "plan" not in logs
    assert len(videos) == 2 and len(videos[0]) >= 1 and videos[0][0].shape == (8, 18, 3)
    assert (videos[0][0][:, :6] == 60).all() and (videos[0][0][:, 6:12] == 120).all() and (videos[0][0][:, 12:] == 180).all()
