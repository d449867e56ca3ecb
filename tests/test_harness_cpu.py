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
