"""Deterministic stand-in for robosuite / dm_control (not installable here): the state is a point
that moves by the mean of the first three action dims; the task succeeds once the accumulated
|motion| exceeds a per-seed threshold or after `horizon` steps."""
import numpy as np


class FakeEnv:
    def __init__(self, lowdim_obs=(), rgb_obs=(), horizon=24, obs_dims=None, with_image=False, **_):
        self.keys = list(lowdim_obs)
        self.rgb = [k[len("latent_"):] if k.startswith("latent_") else k for k in rgb_obs]
        self.horizon = horizon
        self.obs_dims = obs_dims or {}
        self.with_image = with_image
        self.t = 0

    def _obs(self):
        ob = {}
        for k in self.keys:
            d = self.obs_dims.get(k, 3)
            ob[k] = (np.arange(d, dtype=np.float32) * 0.01 + self.pos).astype(np.float32)
        for k in self.rgb:
            if self.with_image:
                ob[k] = np.full((64, 64, 3), (self.t * 10) % 256, dtype=np.float32)
            else:
                ob["latent_" + k] = np.full((16,), 0.1 * self.pos, dtype=np.float32)
        return ob

    def reset(self):
        self.rng = np.random.RandomState(np.random.randint(1 << 30))
        self.goal = 0.05 + 0.05 * self.rng.rand()
        self.pos, self.moved, self.t = 0.0, 0.0, 0
        return self._obs()

    def step(self, action):
        d = float(np.mean(np.asarray(action)[:3]))
        self.pos += d * 0.01
        self.moved += abs(d) * 0.01
        self.t += 1
        return self._obs(), 1.0 if self.is_success()["task"] else 0.0, self.t >= self.horizon, {}

    def is_success(self):
        return {"task": self.moved >= self.goal}


def make_env(**kw):
    return FakeEnv(**kw)


class FakePolicy:
    """CPU policy with the agent's calling convention; records the batch sizes it saw."""
    config = {"name": "dp_agent", "action_horizon": 4}

    def __init__(self):
        self.batches = []

    def sample(self, batch, rng):
        import numpy as np
        x = next(iter(batch["obs"].values()))
        self.batches.append(x.shape[0])
        a = np.full((x.shape[0], 4, 7), 0.5, dtype=np.float32)      # (a NumPy array: torch's Tensor.__array__ has no copy= keyword under NumPy 2)
        return a, {}

    sample_viz = sample


class _TS:
    def __init__(self, observation, reward):
        self.observation, self.reward = observation, reward


class FakeAlohaEnv:
    """dm_control-style stand-in for envs/alohasim_env.make_sim_env: TimeStep objects, `task.max_reward`, images in
    observation['images'][camera] (here channel-first and in [0, 1], so that process_aloha_obs has both conversions to do)."""
    class task:
        max_reward = 4

    def __init__(self, task_name="sim_transfer_cube", horizon=12, **_):
        self.horizon, self.t, self.moved = horizon, 0, 0.0

    def _ts(self):
        img = np.full((3, 64, 64), ((self.t * 10) % 200 + 55) / 255.0, dtype=np.float32)
        obs = {"qpos": (np.arange(14, dtype=np.float32) * 0.01 + 0.1 * self.moved), "images": {"wrist64": img, "top": img}}
        return _TS(obs, self.task.max_reward if self.moved >= self.goal else 0)

    def reset(self):
        self.goal = 0.02 + 0.02 * np.random.rand()
        self.t, self.moved = 0, 0.0
        return self._ts()

    def step(self, action):
        self.t += 1
        self.moved += abs(float(np.mean(np.asarray(action)[:3]))) * 0.01
        return self._ts()


def make_aloha_env(**kw):
    return FakeAlohaEnv(**kw)
