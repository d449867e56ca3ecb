"""The Python boundary on CPU: DeviceArray against the reference's call-site idioms
(utils/rm_env_utils.py:183-196, utils/aloha_env_utils.py:93-98, eval_bc.py:128-151), the
eval_loss metric definitions on a fake policy, and bench.py's self-launcher (dry run, gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd.arrays import CallRecord, DeviceArray, as_tensor
from latent_diffusion_planning_amd.harness import eval_loss_metrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_array_survives_the_reference_call_sites():
    g = np.random.default_rng(0)
    act = g.standard_normal((5, 4, 7)).astype(np.float32)
    viz = g.uniform(-1, 1, (5, 5, 3, 8, 8)).astype(np.float32)
    batch_action, plan_dict = DeviceArray(torch.tensor(act)), {"plan_viz": DeviceArray(torch.tensor(viz))}
    # ---- utils/rm_env_utils.py:184-192 idiom
    plan_viz = plan_dict["plan_viz"]
    plan_viz = (np.clip((np.array(plan_viz) + 1) / 2, 0, 1) * 255).astype(np.uint8)
    batch_action = np.array(batch_action)
    for idx_enum in range(5):
        assert batch_action[idx_enum].shape == (4, 7) and plan_viz[idx_enum].shape == (5, 3, 8, 8)
    np.testing.assert_array_equal(batch_action, act)
    # ---- utils/aloha_env_utils.py:96 idiom (arithmetic and ndarray methods on the result itself)
    pv = ((plan_dict["plan_viz"] + 1) / 2 * 255).astype(np.uint8).transpose(0, 1, 3, 4, 2)
    assert pv.shape == (5, 5, 8, 8, 3) and pv.dtype == np.uint8
    # ---- eval_bc.py:135-151 idiom
    pred = DeviceArray(torch.tensor(act))
    actions = g.standard_normal((5, 9, 7)).astype(np.float32)
    H = pred.shape[1]
    m = np.mean(np.square(actions[:, :H, :] - pred[:, :H, :]))
    assert np.isclose(m, np.mean((actions[:, :4] - act) ** 2))
    assert np.isclose(float(DeviceArray(torch.tensor(3.5))), 3.5)
    assert len(pred) == 5 and pred.ndim == 3 and pred.dtype == np.float32 and pred.size == 140
    assert as_tensor(pred) is pred.tensor and torch.equal(pred.cpu(), torch.tensor(act))
    assert np.array_equal(np.stack([pred, pred])[1], act)


def test_device_array_is_lazy_and_runs_the_completion_hook_once():
    calls = []
    rec = CallRecord(lambda r: calls.append(len(r.arrays)))
    a = DeviceArray(torch.ones(2, 3), record=rec)
    made = []

    def thunk():
        made.append(1)
        return a.tensor * 2
    lazy = DeviceArray(thunk=thunk, shape=(2, 3), record=rec)
    assert lazy.shape == (2, 3) and not made and not calls         # nothing computed, nothing synchronised
    assert np.array_equal(np.array(lazy), np.full((2, 3), 2.0)) and made == [1] and calls == [2]
    np.array(a)
    assert calls == [2] and made == [1]                            # hook ran once for the whole call


def test_fault_recovery_swaps_every_array_of_the_call():
    def recover(rec):
        arrs = rec.arrays
        arrs[0]._swap(torch.full((2,), 7.0))
        arrs[1]._swap(None)                                         # lazy: recomputed from the swapped input
    rec = CallRecord(recover)
    a = DeviceArray(torch.zeros(2), record=rec)
    lazy = DeviceArray(thunk=lambda: a.tensor + 1, shape=(2,), record=rec)
    lazy.tensor                                                     # decoded early from the bad data
    assert np.array_equal(np.array(a), [7.0, 7.0])
    assert np.array_equal(np.array(lazy), [8.0, 8.0])


class _MetricPolicy:
    """Returns fixed predictions so the metric definitions can be checked by hand."""
    config = dict(obs_horizon=1, action_horizon=4)
    use_planner = True

    def __init__(self, sa, full):
        self.sa, self.full = sa, full

    def sample_action(self, batch, rng):
        return DeviceArray(torch.tensor(self.sa))

    def sample(self, batch, rng):
        return DeviceArray(torch.tensor(self.full)), {"plan_mse": DeviceArray(torch.tensor(0.25))}


def test_eval_loss_metrics_follow_eval_bc_definitions():
    g = np.random.default_rng(1)
    B, H, A = 3, 9, 7
    actions = g.standard_normal((B, H, A)).astype(np.float32)
    sa = g.standard_normal((B, H - 1, A)).astype(np.float32)       # sample_action: IDM on the true plan
    full = g.standard_normal((B, 4, A)).astype(np.float32)         # sample: planner + IDM, action_horizon rows
    m = eval_loss_metrics(_MetricPolicy(sa, full), {"obs": {}, "actions": actions}, 0)
    assert set(m) == {"action_mse", "action_mse_0", "action_mse_1", "action_mse_2", "full_action_mse",
                      "full_action_mse_0", "full_action_mse_1", "full_action_mse_2", "plan_mse"}
    # eval_bc.py:135-138: raw batch actions against sample_action
    assert np.isclose(m["action_mse"], np.mean((actions[:, :H - 1] - sa) ** 2))
    for i in range(3):
        assert np.isclose(m[f"action_mse_{i}"], np.mean((actions[:, i] - sa[:, i]) ** 2))
        assert np.isclose(m[f"full_action_mse_{i}"], np.mean((actions[:, i] - full[:, i]) ** 2))
    # eval_bc.py:143-151: against sample
    assert np.isclose(m["full_action_mse"], np.mean((actions[:, :4] - full) ** 2))
    assert m["plan_mse"] == 0.25
    short = eval_loss_metrics(_MetricPolicy(sa[:, :2], full[:, :2]), {"obs": {}, "actions": actions}, 0)
    assert "action_mse_1" in short and "action_mse_2" not in short   # the reference's try/except


@pytest.mark.parametrize("n,config", [(1, 1), (2, 1), (2, 2), (2, 3), (2, 4)])
def test_bench_self_launches_its_ranks(n, config):
    """`python bench.py --gpus N [--config C]` must start N ranks by itself; --dry-run swaps the GPU work
    for a gloo all-gather (of what configuration C gathers: plans, or plans and actions) so the launcher and the collective
    plumbing of every BASELINE.json configuration run on CPU."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--config", str(config)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["config"]["ranks_seen_by_backend"] == n and line["config"]["gather_ok"]
    assert line["data"].startswith("INVALID")
    want = {1: (256, "ddim", 100), 2: (1024, "ddpm", 100), 3: (512, "ddpm", 100), 4: (1024, "ddim", 50)}[config]
    c = line["config"]
    assert (c["baseline_config"], c["plans_per_gpu"], c["sampler"], c["denoise_steps"]) == (config,) + want


def test_bench_without_gpus_fails_on_the_hardware_not_on_the_launcher():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "needs 2 MI355X" in r.stderr
