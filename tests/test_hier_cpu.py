"""LDPHierAgent (SURVEY.md 8f tail; agent/ldp_hier_agent.py): oracle cross-checks and host logic that need no GPU."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import np64, torch32
from tests import cfgs
from tests.cases import HIER_IDM_DOWN, hier_idm_params
from tests.util import rng


def test_two_level_unet_the_two_restatements_agree():
    """The hierarchical agent's IDM: ConditionalUnet1D(down_dims [256, 512]) over 4 action steps, input_dim = action_dim,
    cond = 2 obs_dim.  oracle/np64.py (explicit loops) against oracle/torch32.py (torch.nn.functional), float64."""
    ip = hier_idm_params()
    g = rng(11)
    a, tr = g.standard_normal((3, 4, 7)), g.uniform(-1, 1, (3, 50))
    k = np.array([0, 37, 99])
    ref = np64.unet_forward(ip, a, k, tr, down_dims=HIER_IDM_DOWN)
    P = torch32.TorchParams(ip, dtype=torch.float64)
    got = torch32.unet_forward(P, torch.tensor(a), torch.tensor(k), torch.tensor(tr), down_dims=HIER_IDM_DOWN).numpy()
    assert ref.shape == (3, 4, 7)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)      # the sinusoid argument is float32 in both, as in the reference
    # only ONE of the two skips is consumed (networks/diffusion_nets_v2.py:141-156): the parameter count says so
    spec = W.PlannerSpec(7, 50, down_dims=HIER_IDM_DOWN)
    assert [b[:2] for b in spec.blocks()] == [(7, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512), (1024, 256), (256, 256)]


def test_the_reference_configuration_is_refused_with_the_reason():
    """train_bc.yaml: horizon 16 -> pred_horizon 15, idm_horizon 4 -> 3 planner states, which the three-level U-Net
    cannot process; the checks run before any device is touched."""
    from latent_diffusion_planning_amd.hier_agent import LDPHierAgent
    data = cfgs.RM_LIFT
    with pytest.raises(ValueError, match="3 planner states.*multiple of 4"):
        LDPHierAgent.create(0, None, data["shape_meta"], **cfgs.hier_kwargs(data, pred_horizon=15))
    with pytest.raises(AssertionError):                                   # agent/ldp_hier_agent.py:618
        LDPHierAgent.create(0, None, data["shape_meta"], **cfgs.hier_kwargs(data, action_horizon=6))
    with pytest.raises(ValueError, match="action_horizon 12 states are taken from a plan of 8"):
        LDPHierAgent.create(0, None, data["shape_meta"], **cfgs.hier_kwargs(data, action_horizon=12))


def test_hier_oracle_shapes_and_plan_assembly():
    """HierAgentOracle with stand-in loops: the plan is [last observed state, first action_horizon predicted states], the
    transitions are (state, next state) rows, the action is the row-major chunk concatenation."""
    data = cfgs.RM_LIFT
    D, A, B, ih, ah, Tp = 25, 7, 2, 4, 4, 8
    conf = dict(planner_n_diffusion_steps=100, idm_n_diffusion_steps=100, lowdim_obs=data["lowdim_obs"], rgb_obs=data["rgb_obs"],
                obs_horizon=1, pred_horizon=32, action_horizon=ah, idm_horizon=ih, obs_dim=D, action_dim=A, vae_feature_dim=16)
    seen = {}

    def pfn(p, cond, x0, z, n, s, smp):
        seen["cond"] = cond
        return np.arange(B * Tp * D, dtype=np.float64).reshape(B, Tp, D) * 1e-3

    def ifn(p, trans, a0, z, n, s, smp):
        seen["trans"] = trans
        return np.tile(np.linspace(-0.5, 0.5, ih * A).reshape(1, ih, A), (trans.shape[0], 1, 1)) + np.arange(trans.shape[0])[:, None, None] * 0.01

    orc = np64.HierAgentOracle(conf, None, None, None, data["obs_normalization"], planner_sample_fn=pfn, idm_sample_fn=ifn)
    batch = cfgs.synth_latent_batch(data, B, 1, 3)
    act, m = orc.sample_viz(batch, np.zeros((B, Tp, D)), None, np.zeros((B * ah, ih, A)), None, decode=False)
    assert act.shape == (B, ah * ih, A) and m["plan"].shape == (B, ah + 1, D) and seen["trans"].shape == (B * ah, 2 * D)
    assert np.allclose(seen["trans"][1, :D], m["plan"][0, 1]) and np.allclose(seen["trans"][1, D:], m["plan"][0, 2])
    assert np.allclose(act[1, 4 * 1 + 2], np.clip(np.linspace(-0.5, 0.5, ih * A).reshape(ih, A)[2] + 0.05, -1, 1))      # row (b=1, h=1), step 2
