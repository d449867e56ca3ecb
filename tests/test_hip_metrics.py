"""LDPAgent.get_metrics (forward only) against the float64 oracle, and the primitives under it.  -m gpu.
Reference: agent/ldp_agent.py:113-180 (plan_loss / idm_loss / loss), :328-349 (get_metrics_step); caller eval_bc.py:127."""
import numpy as np
import pytest
import torch

from oracle import np64
from tests import cfgs
from tests.cases import load_case, unflat_obs
from tests.util import assert_close, idm_params, make_agent, planner_params, rng

pytestmark = pytest.mark.gpu


def test_add_noise_and_reduce_stats_primitives():
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    g = rng(3)
    x0, nz = g.uniform(-1, 1, (37, 8, 25)), g.standard_normal((37, 8, 25))
    t = g.integers(0, 100, 37)
    t[:3] = (0, 99, 50)
    got = e.add_noise(torch.tensor(x0, dtype=torch.float32), torch.tensor(nz, dtype=torch.float32), torch.tensor(t), 100).cpu().numpy()
    ref = np64.ddpm_add_noise(np.float32(x0).astype(np.float64), np.float32(nz).astype(np.float64), t)
    assert_close(got, ref, 1e-6, "FlaxDDPMScheduler.add_noise")
    x = g.standard_normal(100003) * 3 + 0.7
    st = e.reduce_stats(torch.tensor(x, dtype=torch.float32)).cpu().numpy()
    x32 = np.float32(x).astype(np.float64)
    assert st[0] == np.float32(x32.min()) and st[1] == np.float32(x32.max())
    assert_close(st[2:], [x32.mean(), x32.std()], 1e-6, "mean / population std")
    e.close()


@pytest.mark.parametrize("cfg", ["rm", "aloha"])
def test_get_metrics_matches_golden(cfg):
    """Explicit (t, noise) for both losses; every key of the reference's metrics dict, values against the float64 oracle: the two
    losses at 1e-4 relative (they are means over 600 / 168 squared errors of O(1) eps), the statistics at 2e-6 (relative where |x| > 1: the synthetic aloha actions normalise to +-131)."""
    D, A = (25, 7) if cfg == "rm" else (30, 14)
    ag, data = make_agent(cfg, planner_params(D=D), idm_params(D=D, A=A))
    inp, exp = load_case(f"agent_get_metrics_{cfg}")
    noise = dict(t_plan=inp["t_plan"].astype(np.int64), noise_plan=inp["noise_plan"], t_idm=inp["t_idm"].astype(np.int64),
                 noise_idm=inp["noise_idm"])
    m = ag.get_metrics(unflat_obs(inp), 0, noise=noise)
    assert set(m) == set(exp), f"metric keys differ: {set(m) ^ set(exp)}"
    for k in ("plan_loss", "idm_loss", "loss"):
        assert abs(float(m[k]) - float(exp[k])) <= 1e-4 * max(1.0, abs(float(exp[k]))), (k, float(m[k]), float(exp[k]))
    for k in exp:
        if k not in ("plan_loss", "idm_loss", "loss"):
            assert abs(float(m[k]) - float(exp[k])) <= 2e-6 * max(1.0, abs(float(exp[k]))), (k, float(m[k]), float(exp[k]))
    # the reference's aggregation idiom (eval_bc.py:152)
    agg = {k: float(np.mean([mm[k] for mm in (m, m)])) for k in m}
    assert abs(agg["loss"] - float(exp["loss"])) <= 1e-4 * float(exp["loss"])
    # seeded mode: deterministic, finite, different seeds differ; loss = plan_loss + idm_loss
    batch = unflat_obs(inp)
    a, b, c = ag.get_metrics(batch, 5), ag.get_metrics(batch, 5), ag.get_metrics(batch, 6)
    assert float(a["loss"]) == float(b["loss"]) and float(a["loss"]) != float(c["loss"])
    assert abs(float(a["loss"]) - (float(a["plan_loss"]) + float(a["idm_loss"]))) < 1e-5 and np.isfinite(float(a["loss"]))
    assert 0.5 < float(a["plan_loss"]) < 3.0 and 0.5 < float(a["idm_loss"]) < 30.0     # random-init nets: eps-MSE of order 1 (aloha's actions are +-131)
    with pytest.raises(KeyError):
        ag.get_metrics({"obs": batch["obs"]}, 0)
    ag._engine.close()


def test_eval_loss_metrics_merges_get_metrics_like_the_reference():
    """eval_bc.py:107-159: metrics = agent.get_metrics(batch, rng); then the sampling metrics are ADDED to that dict."""
    from latent_diffusion_planning_amd.harness import eval_loss_metrics
    ag, data = make_agent("rm", planner_params(), idm_params())
    batch = cfgs.synth_latent_batch(data, 4, 9, 31, with_actions=True)
    m = eval_loss_metrics(ag, batch, 3)
    for k in ("plan_loss", "idm_loss", "loss", "emb_std", "action_max", "action_mse", "action_mse_0", "full_action_mse", "plan_mse"):
        assert k in m and np.isfinite(m[k]), k
    ag._engine.close()
