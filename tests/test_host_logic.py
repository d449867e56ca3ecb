"""Host-side logic of the drop-in boundary that needs no GPU."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import _lib, flops, schedule, weights as W
from latent_diffusion_planning_amd.agent import LDPAgent, ParamState, _norm_entry, _seed_of
from latent_diffusion_planning_amd.dist import shard_batch, shard_bounds
from tests import cfgs


def test_create_fails_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.LDPHipUnavailable):
        LDPAgent.create(0, None, cfgs.RM_LIFT["shape_meta"], **cfgs.agent_kwargs(cfgs.RM_LIFT))


def test_seed_forms():
    assert _seed_of(7) == 7 and _seed_of(np.int64(9)) == 9
    assert _seed_of(np.array([1, 2], dtype=np.uint32)) == (1 << 32) | 2
    g = torch.Generator().manual_seed(123)
    s1, s2 = _seed_of(g), _seed_of(g)                   # a stateful generator advances: fresh noise every call
    assert s1 != s2 and _seed_of(torch.Generator().manual_seed(123)) == s1
    with pytest.raises(TypeError):
        _seed_of(np.zeros(3))


def test_param_state_replace_accepts_nested_flax_trees():
    st = ParamState({"Dense_0/kernel": np.zeros((2, 3), np.float32)})
    nested = {"Dense_0": {"kernel": np.ones((2, 3)), "bias": np.ones(3)}}
    st2 = st.replace(params=nested, ema_params=nested)
    assert set(st2.params) == {"Dense_0/kernel", "Dense_0/bias"} and st2.params["Dense_0/kernel"].dtype == np.float32
    assert st.params["Dense_0/kernel"].sum() == 0            # original untouched
    assert W.unflatten(st2.params)["Dense_0"]["bias"].shape == (3,)


def test_norm_entries():
    assert _norm_entry(dict(min=0, max=255)) == dict(min=0.0, max=255.0)
    e = _norm_entry(dict(min=[0.0, 1.0], max=[2.0, 3.0]))
    assert e["min"].dtype == np.float32
    with pytest.raises(NotImplementedError):
        _norm_entry(dict(mean=0, std=1))
    with pytest.raises(NotImplementedError):
        _norm_entry(dict(foo=1))


def _host_agent(data):
    kw = cfgs.agent_kwargs(data)
    cfg = dict(lowdim_obs=kw["lowdim_obs"], rgb_obs=kw["rgb_obs"], obs_horizon=1, pred_horizon=8,
               action_horizon=4, obs_dim=25, action_dim=7, vae_feature_dim=16)
    a = LDPAgent(ParamState({}), ParamState({}), None, None, True, True, 1, 1, cfg, None, None, None, None,
                 torch.device("cpu"))
    return a


def test_get_obs_cond_layout_and_replace():
    a = _host_agent(cfgs.RM_LIFT)
    b = cfgs.synth_latent_batch(cfgs.RM_LIFT, 3, 2, 0)
    oc = a.get_obs_cond(b["obs"])
    assert oc.shape == (3, 2, 25)
    np.testing.assert_array_equal(oc[..., :16].numpy(), b["obs"]["latent_agentview_image"])
    np.testing.assert_array_equal(oc[..., 16:19].numpy(), b["obs"]["robot0_eef_pos"])
    np.testing.assert_array_equal(oc[..., 19:23].numpy(), b["obs"]["robot0_eef_quat"])
    np.testing.assert_array_equal(oc[..., 23:].numpy(), b["obs"]["robot0_gripper_qpos"])
    st = ParamState({"x": np.zeros(1, np.float32)})
    a2 = a.replace(planner_state=st)
    assert a2.planner_state is st and a.planner_state is not st and a2.config is a.config
    assert st.replace(params=st.params).version != st.version and st.replace(step=3).version == st.version
    assert a2.get_params()["planner_params"] is st.params
    with pytest.raises(AttributeError):
        a.replace(nope=1)
    # opt_state travels with .replace (checkpoint restore); new parameters start a new optimiser state
    o = dict(mu={"x": np.ones(1)}, nu={"x": np.ones(1)}, count=7)
    st3 = st.replace(opt_state=o, step=7)
    assert st3.opt_state["count"] == 7 and st3.step == 7 and st3.version != st.version and st3.replace(params=st.params).opt_state is None
    with pytest.raises(AttributeError):
        st.replace(nope=1)


def test_update_gates_follow_the_reference():
    """agent/ldp_agent.py:223-232: update_planner_every / update_idm_every / update_idm_after / update_planner_until / update_planner_after."""
    a = _host_agent(cfgs.RM_LIFT)
    a.config.update(update_planner_every=2, update_idm_every=1, update_idm_after=3, update_planner_until=6, update_planner_after=2)
    got = [a._gates(s) for s in range(8)]
    assert got == [(False, False), (False, False), (True, False), (False, True), (True, True), (False, True), (False, True), (False, True)]


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 5, 8, 256, 8192):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, 5, 1, 0, with_actions=True)
    loc, lo, n = shard_batch(batch, 2, 1)
    assert (lo, n) == (3, 5) and loc["actions"].shape[0] == 2 and loc["obs"]["robot0_eef_pos"].shape[0] == 2


def test_flop_model_matches_survey():
    p = W.PlannerSpec(25, 25)
    # SURVEY.md 8d counts 2 taps per output of the transposed convs (0.16349 / 0.45467 GFLOP);
    # counting only taps that hit data removes 2*(512^2 + 256^2) MACs (edge outputs).
    edge = 2.0 * 2 * (512 * 512 + 256 * 256)
    assert abs(flops.planner_forward_flops(p, 8) + edge - 0.16349e9) < 2e4
    assert abs(flops.planner_forward_flops(p, 16) + edge - 0.45467e9) < 2e4
    assert abs(flops.planner_forward_flops(p, 8, hoisted=True) + edge - 0.15438e9) < 2e4
    assert abs(flops.idm_forward_flops(W.IDMSpec(25, 7)) - 3.572e6) < 1e3
    assert abs(flops.idm_forward_flops(W.IDMSpec(30, 14)) - 3.584e6) < 1e3


def test_weight_tree_checks_and_npz_roundtrip(tmp_path):
    spec = W.PlannerSpec(5, 5, down_dims=(16, 32))
    p = W.init_planner_params(spec, 3)
    W.check_params(p, W.planner_shapes(spec))
    bad = dict(p)
    bad.pop("Conv_0/bias")
    with pytest.raises(KeyError):
        W.check_params(bad, W.planner_shapes(spec))
    bad = dict(p)
    bad["Conv_0/bias"] = np.zeros(7, np.float32)
    with pytest.raises(ValueError):
        W.check_params(bad, W.planner_shapes(spec))
    f = str(tmp_path / "w.npz")
    W.save_npz(f, planner_params=p)
    q = W.load_npz(f)["planner_params"]
    assert list(q) == list(p) and all(np.array_equal(q[k], p[k]) for k in p)
    exact = W.init_planner_params(spec, 3, perturb=False)
    assert np.all(exact["Conv_0/bias"] == 0) and np.all(exact["Conv1dBlock_0/GroupNorm_0/scale"] == 1)
