"""LDPHierAgent on the GPU: the two-level U-Net against the oracle, the agent call against its goldens (1e-4), the
reference's return structure.  -m gpu."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import torch32
from tests import cfgs
from tests.cases import HIER_IDM_DOWN, hier_idm_params, load_case, unflat_obs
from tests.util import assert_close, planner_params, rng

pytestmark = pytest.mark.gpu


def _f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


@pytest.mark.parametrize("B", [1, 3, 17, 300, 1030])
def test_two_level_unet_forward_matches_oracle(B):
    """ConditionalUnet1D(down_dims [256, 512]) over 4 positions (the hierarchical agent's IDM): launch plans for the
    (4, 256) / (2, 512) levels, the 4 -> 2 stride-2 conv and the 2 -> 4 transposed conv at 256 channels."""
    from latent_diffusion_planning_amd.engine import HipEngine
    ip = hier_idm_params()
    e = HipEngine(obs_dim=7, action_dim=7, global_cond_dim=50, pred_horizon=4, action_horizon=4, down_dims=HIER_IDM_DOWN)
    e.load_params(planner=ip)
    g = rng(70 + B)
    a, tr = g.standard_normal((B, 4, 7)), g.uniform(-1, 1, (B, 50))
    k = g.integers(0, 100, size=B)
    got = e.unet_forward(_f32(a), torch.tensor(k), _f32(tr)).cpu().numpy()
    rows = np.unique(np.concatenate([np.arange(min(B, 3)), np.arange(max(B - 2, 0), B)]))
    P = torch32.TorchParams(ip, dtype=torch.float64)
    ref = torch32.unet_forward(P, torch.tensor(a[rows]), torch.tensor(k[rows]), torch.tensor(tr[rows]), down_dims=HIER_IDM_DOWN).numpy()
    assert_close(got[rows], ref, 2e-5, f"two-level U-Net forward B={B}")
    e.check_fault()
    e.close()


@pytest.fixture(scope="module")
def hier():
    from latent_diffusion_planning_amd.hier_agent import LDPHierAgent
    data = cfgs.RM_LIFT
    ag = LDPHierAgent.create(0, None, data["shape_meta"], vae_params=W.init_vae_params(seed=2), **cfgs.hier_kwargs(data))
    ag = ag.replace(planner_state=ag.planner_state.replace(params=planner_params()),
                    idm_state=ag.idm_state.replace(params=hier_idm_params()))
    yield ag, data
    ag._engine.close()
    ag._idm_engine.close()


@pytest.mark.parametrize("name,sampler,n_steps", [("agent_hier_sample_viz_rm_b2", "ddpm", None),
                                                  ("agent_hier_sample_viz_rm_ddim50_b3", "ddim", 50)])
def test_hier_sample_viz_matches_golden(hier, name, sampler, n_steps):
    ag, data = hier
    inp, exp = load_case(name)
    noise = {k: _f32(inp[k]) for k in ("x_init", "x_noise", "a_init", "a_noise") if k in inp}
    act, met = ag.sample(unflat_obs(inp), 0, noise=noise, sampler=sampler, n_steps=n_steps)
    assert_close(np.array(met["plan"]), exp["plan"], 1e-4, f"{name}: plan")
    assert_close(np.array(act), exp["action"], 1e-4, f"{name}: action (rm actions are clipped, not scaled)")
    assert act.shape == (exp["action"].shape[0], 16, 7)


def test_hier_agent_return_structure_and_philox_mode(hier):
    """agent/ldp_hier_agent.py:385-403, 437-461: action (B, action_horizon * idm_horizon, A); plan_viz = the decoded plan
    states after the start state, each repeated idm_horizon times; plan_mse for training batches; rows do not depend on
    what else is in the batch (Philox keyed by the global row)."""
    ag, data = hier
    B = 5
    batch = cfgs.synth_latent_batch(data, B, 1, 21)
    act, met = ag.sample_viz(batch, 7)
    assert act.shape == (B, 16, 7) and met["plan"].shape == (B, 5, 25) and met["plan_viz"].shape == (B, 16, 3, 64, 64)
    a = np.array(act)
    assert np.isfinite(a).all() and np.abs(a).max() <= 1.0
    viz = np.array(met["plan_viz"])
    assert np.array_equal(viz[:, 0], viz[:, 3]) and not np.array_equal(viz[:, 3], viz[:, 4])
    dec = np.array(ag.vae_decode(met["plan"]))
    assert np.array_equal(viz[:, 4], dec[:, 2])
    # the same rows as their own sub-batch at the right row offset
    sub = {"obs": {k: v[3:] for k, v in batch["obs"].items()}}
    act2, met2 = ag.sample(sub, 7, row_offset=3)
    assert np.array_equal(np.array(act2), a[3:]) and np.array_equal(np.array(met2["plan"]), np.array(met["plan"])[3:])
    # a training batch: obs_horizon + planner-states frames and actions -> plan_mse
    tb = cfgs.synth_latent_batch(data, 3, 9, 22, with_actions=True)
    _, m3 = ag.sample(tb, 1)
    assert "plan_mse" in m3 and np.isfinite(float(m3["plan_mse"]))
    with pytest.raises(ValueError, match="must have shape"):         # update() is built (below); this batch has 9 frames, the planner trains on 1 + 32
        ag.update(tb, 0, 0)
    # the flat agent's get_metrics must not run the MLP IDM this agent never loads: the hierarchical readings are overridden (test below);
    # the reference's own evaluation returns an empty dict for this agent (eval_bc.py:107-109), and so does the harness
    from latent_diffusion_planning_amd.harness import eval_loss_metrics
    assert eval_loss_metrics(ag, tb, 3) == {}
    assert set(ag.get_params()) == {"planner_params", "idm_params"} and ag.config["idm_horizon"] == 4


def test_a_fault_on_the_idm_handle_is_recovered(hier):
    """LDPHierAgent drives two engine handles; the fault protocol used to poll only the planner's, so a fault on the
    IDM handle went unnoticed and then wedged that handle.  Both kinds of fault injected on the IDM handle: the call is recomputed with the
    right warning, later calls run clean, neither handle stays refused."""
    import warnings
    ag, data = hier
    batch = cfgs.synth_latent_batch(data, 5, 1, 33)
    clean = np.array(ag.sample(batch, 4)[0])
    act, met = ag.sample(batch, 4)
    ag._idm_engine.set_option("inject_fault", 1)
    with pytest.warns(RuntimeWarning, match="recomputed in safe mode"):
        got = np.array(act)
    assert ag._idm_engine.get_option("safe_mode") == 1 and ag._engine.get_option("safe_mode") == 0
    assert_close(got, clean, 1e-4, "actions recomputed after a fault on the IDM handle")
    act, met = ag.sample(batch, 4)
    ag._idm_engine.set_option("inject_fault", 2)
    with pytest.warns(RuntimeWarning, match="three bf16 planes"):
        got2 = np.array(act)
    assert ag._idm_engine.get_option("range_fallback") == 1
    assert_close(got2, clean, 1e-4, "actions recomputed after a range fault on the IDM handle")
    # an unread faulted call followed by a new one: acknowledged on both handles, no wedge
    first = ag.sample(batch, 5)
    ag._idm_engine.set_option("inject_fault", 1)
    second = np.array(ag.sample(batch, 6)[0])
    with pytest.warns(RuntimeWarning):
        np.array(first[0])
    assert np.isfinite(second).all()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        np.array(ag.sample(batch, 7)[0])


def test_hier_sample_action_matches_golden(hier):
    """agent/ldp_hier_agent.py:345-383: the IDM U-Net on the batch's own consecutive frames -- (B, H, obs) in, (B, (H - 1) * idm_horizon, A) out."""
    ag, data = hier
    inp, exp = load_case("agent_hier_sample_action_rm_ddim50_b2")
    act = ag.sample_action(unflat_obs(inp), 0, noise=dict(a_init=_f32(inp["a_init"])), sampler="ddim", n_steps=50)
    assert act.shape == exp["action"].shape == (2, 16, 7)
    assert_close(np.array(act), exp["action"], 1e-4, "hier sample_action (rm actions are clipped, not scaled)")
    seeded = np.array(ag.sample_action(unflat_obs(inp), 3))
    assert seeded.shape == (2, 16, 7) and np.isfinite(seeded).all() and np.array_equal(seeded, np.array(ag.sample_action(unflat_obs(inp), 3)))
    with pytest.raises(NotImplementedError):
        ag.sample_action_from_plan(unflat_obs(inp), None, 0)


# ---- the hierarchical training step (agent/ldp_hier_agent.py:111-137, 223-322) ---------------------------------------------------------
REF_KEYS = {"plan_loss", "idm_loss", "loss", "emb_min", "emb_max", "emb_mean", "emb_std", "action_min", "action_max", "g_norm", "planner_lr",
            "planner_step", "idm_lr", "idm_step"}


def _hier_step_inputs(inp, s, mixed=False):
    pre = f"s{s}_"
    batch = unflat_obs({k[len(pre):]: v for k, v in inp.items() if k.startswith(pre) and not k.startswith(pre + "mixed_")})
    nz = {k: inp[pre + k] for k in ("noise_plan", "noise_idm")}
    nz["t_plan"], nz["t_idm"] = inp[pre + "t_plan"].astype(np.int64), inp[pre + "t_idm"].astype(np.int64)
    mb = unflat_obs({k[len(pre) + 6:]: v for k, v in inp.items() if k.startswith(pre + "mixed_")}) if mixed else None
    return batch, mb, nz


def _fresh_hier():
    from latent_diffusion_planning_amd.hier_agent import LDPHierAgent
    data = cfgs.RM_LIFT
    ag = LDPHierAgent.create(0, None, data["shape_meta"], **cfgs.hier_kwargs(data))
    return ag.replace(planner_state=ag.planner_state.replace(params=planner_params()),
                      idm_state=ag.idm_state.replace(params=hier_idm_params())), data


@pytest.mark.parametrize("name,mixed", [("agent_hier_update_rm", False), ("agent_hier_update_mixed_rm", True)])
def test_hier_agent_update_matches_golden(name, mixed):
    """`agent, metrics = agent.update(batch, rng, step)` on the hierarchical agent: both ConditionalUnet1Ds train through csrc/train.hip's U-Net tape
    (the planner on every 4th future state, the IDM on chunks of 4 actions), each in its own engine handle.  Losses, the global norm over BOTH
    gradient trees, learning rates of every step; per-leaf digests of the step-0 gradients and of the parameters after 1 and after n steps."""
    from tests.util import tree_digest
    inp, exp = load_case(name)
    ag, data = _fresh_hier()
    n = len(exp["g_norm"])
    for s in range(n):
        batch, mb, nz = _hier_step_inputs(inp, s, mixed)
        prev = ag
        ag, m = ag.update_mixed(batch, mb, 100 + s, s, noise=nz) if mixed else ag.update(batch, 100 + s, s, noise=nz)
        assert set(m) == REF_KEYS | {f"{k}_{e}" for k in batch["obs"] for e in ("min", "max")}, sorted(set(m) ^ REF_KEYS)
        assert m["planner_step"] == s and m["idm_step"] == s and ag.planner_state.step == s + 1 and ag.idm_state.step == s + 1
        for k in ("plan_loss", "idm_loss", "g_norm"):
            assert abs(float(m[k]) - exp[k][s]) <= 2e-5 * max(1.0, abs(exp[k][s])) * (1 if k != "g_norm" else 5), (s, k, float(m[k]), exp[k][s])
        for k in ("planner_lr", "idm_lr"):
            assert abs(float(m[k]) - exp[k][s]) <= 1e-6 * max(exp[k][s], 1e-12), (s, k)
        if s == 0:
            for k in ("emb_min", "emb_max", "emb_mean", "emb_std", "action_min", "action_max"):
                assert abs(float(m[k]) - float(exp[k])) <= 1e-5, k
            for eng, key, spec, seed in ((ag._engine, "grads_planner", ag._planner_spec, 11), (ag._idm_engine, "grads_idm", ag._idm_unet_spec, 12)):
                got = tree_digest(eng.train_read("planner", eng.TRAIN_GRADS, W.planner_shapes(spec)), seed)
                scale = np.maximum(exp[key][:, 1:2], 1e-30)
                err = (np.abs(got - exp[key]) / scale)[:, 3:].max()
                print(f"{name}: {key}, step 0: worst digest entry off by {err:.2e} of its leaf's max")
                assert err <= 1e-4, key
            for tree, key, seed in ((ag.planner_state.params, "planner_after_1", 13), (ag.idm_state.params, "idm_after_1", 14)):
                assert np.abs(tree_digest(tree, seed)[:, 3:] - exp[key][:, 3:]).max() <= 1e-5, key
    for tree, key, seed in ((ag.planner_state.params, "planner_after_n", 15), (ag.idm_state.params, "idm_after_n", 16)):
        worst = np.abs(tree_digest(tree, seed)[:, 3:] - exp[key][:, 3:]).max()
        print(f"{name}: {key}: max |param - float64 optimiser| over the digests = {worst:.2e}")
        assert worst <= 1e-5, key
    with pytest.raises(RuntimeError, match="superseded"):
        prev.idm_state.opt_state                                      # (never read while it was the newest: its buffers went to the next step)
    # the trained agent samples with the trained weights of BOTH handles (published on demand), bit-equal to an agent built from them
    sb = cfgs.synth_latent_batch(data, 3, 1, 5)
    act = np.array(ag.sample(sb, 9)[0])
    ag2, _ = _fresh_hier()
    ag2 = ag2.replace(planner_state=ag2.planner_state.replace(params=ag.planner_state.params), idm_state=ag2.idm_state.replace(params=ag.idm_state.params))
    assert np.array_equal(np.array(ag2.sample(sb, 9)[0]), act)
    for a in (ag, ag2):
        a._engine.close(); a._idm_engine.close()


def test_hier_update_gates_and_philox_mode():
    """The gating is LDPAgent's (agent/ldp_hier_agent.py:223-232 is the same code); a step without explicit noise draws timesteps (host PCG64) and
    Philox noise from `rng`: same rng -> same step, and a skipped network keeps its state object."""
    ag, data = _fresh_hier()
    ag.config.update(update_idm_every=2)
    batch = cfgs.synth_latent_batch(data, 3, 33, 91, with_actions=True)
    a1, m1 = ag.update(batch, 5, 1)                                   # step 1: the IDM is skipped
    assert a1.idm_state is ag.idm_state and a1.planner_state.step == 1 and m1["idm_lr"] == 0 and m1["idm_step"] == 0 and float(m1["idm_loss"]) == 0.0
    a2, m2 = a1.update(batch, 6, 2)
    assert a2.idm_state.step == 1 and a2.planner_state.step == 2 and float(m2["idm_loss"]) > 0 and float(m2["g_norm"]) > 0
    ag3, _ = _fresh_hier()
    ag3.config.update(update_idm_every=2)
    a3, m3 = ag3.update(batch, 5, 1)
    assert float(m3["plan_loss"]) == float(m1["plan_loss"]) and float(m3["g_norm"]) == float(m1["g_norm"])
    for a in (ag, ag3):
        a._engine.close(); a._idm_engine.close()


def test_hier_get_metrics_is_the_forward_half_of_the_training_step():
    """agent/ldp_hier_agent.py:324-343: the two losses at explicit (t, noise), forward only == the step-0 losses of the update golden (same inputs,
    same initial parameters); nothing trains."""
    inp, exp = load_case("agent_hier_update_rm")
    ag, data = _fresh_hier()
    batch, _, nz = _hier_step_inputs(inp, 0)
    m = ag.get_metrics(batch, 3, noise=nz)
    for k in ("plan_loss", "idm_loss"):
        assert abs(float(m[k]) - exp[k][0]) <= 2e-5 * max(1.0, exp[k][0]), (k, float(m[k]), exp[k][0])
    assert abs(float(m["loss"]) - (exp["plan_loss"][0] + exp["idm_loss"][0])) <= 4e-5 * (exp["plan_loss"][0] + exp["idm_loss"][0])
    for k in ("emb_min", "emb_max", "emb_mean", "emb_std", "action_min", "action_max"):
        assert abs(float(m[k]) - float(exp[k])) <= 1e-5, k
    assert ag.planner_state.step == 0 and ag._engine.train_token["planner"] is None and ag._idm_engine.train_token["planner"] is None
    m2 = ag.get_metrics(batch, 3)                                      # drawn timesteps + Philox noise: finite, repeatable
    assert np.isfinite(float(m2["loss"])) and float(ag.get_metrics(batch, 3)["loss"]) == float(m2["loss"])
    ag._engine.close(); ag._idm_engine.close()
