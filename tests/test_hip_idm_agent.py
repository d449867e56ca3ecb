"""Parity of the IDM path and of the LDPAgent surface (through the C ABI) against the oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import np64, torch32
from tests import cfgs
from tests.util import assert_close, idm_params, planner_params, rng

pytestmark = pytest.mark.gpu


def _t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def _f32(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


# torch-float64 restatement with the np64 signatures (same math, fast enough for 100-step loops)
def _planner_fn(params, obs_cond, x_init, step_noise, n_train, n_steps, sampler):
    P = torch32.TorchParams(params, dtype=torch.float64)
    return torch32.planner_sample(P, _t64(obs_cond), _t64(x_init), None if step_noise is None else _t64(step_noise),
                                  n_train=n_train, n_steps=n_steps, sampler=sampler).numpy()


def _idm_fn(params, trans, a_init, step_noise, n_train, n_steps, sampler):
    P = torch32.TorchParams(params, dtype=torch.float64)
    return torch32.idm_sample(P, _t64(trans), _t64(a_init), None if step_noise is None else _t64(step_noise),
                              n_train=n_train, n_steps=n_steps, sampler=sampler).numpy()


@pytest.fixture(scope="module", params=["rm", "aloha"])
def setup(request):
    from latent_diffusion_planning_amd.engine import HipEngine
    D, A = (25, 7) if request.param == "rm" else (30, 14)
    pp, ip = planner_params(D=D), idm_params(D=D, A=A)
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
    e.load_params(planner=pp, idm=ip)
    yield dict(eng=e, D=D, A=A, pp=pp, ip=ip, name=request.param)
    e.close()


@pytest.mark.parametrize("R", [4, 12, 257])
def test_idm_forward_matches_oracle(setup, R):
    e, D, A = setup["eng"], setup["D"], setup["A"]
    g = rng(300 + R)
    s, a = g.uniform(-1, 1, (R, 2 * D)), g.standard_normal((R, A))
    P = torch32.TorchParams(setup["ip"], dtype=torch.float64)
    for k in (0, 63, 99):
        ref = torch32.idm_forward(P, _t64(s), _t64(a), k).numpy()
        got = e.idm_forward(_f32(s), _f32(a), k).cpu().numpy()
        assert_close(got, ref, 2e-5, f"idm forward R={R} k={k}")
    ref = np64.idm_forward(setup["ip"], s, a, 17)                  # the NumPy definition itself
    assert_close(e.idm_forward(_f32(s), _f32(a), 17).cpu().numpy(), ref, 2e-5, "idm forward vs np64")
    ks = g.integers(0, 100, size=R)
    ref = torch32.idm_forward(P, _t64(s), _t64(a), ks).numpy()
    got = e.idm_forward(_f32(s), _f32(a), torch.tensor(ks)).cpu().numpy()
    assert_close(got, ref, 2e-5, f"idm forward R={R} per-row k")


@pytest.mark.parametrize("sampler,n_steps", [("ddpm", 100), ("ddim", 50)])
def test_idm_sample_matches_oracle(setup, sampler, n_steps):
    e, D, A = setup["eng"], setup["D"], setup["A"]
    R = 12
    g = rng(400 + n_steps)
    tr, a0 = g.uniform(-1, 1, (R, 2 * D)), g.standard_normal((R, A))
    nz = g.standard_normal((n_steps, R, A))
    ref = _idm_fn(setup["ip"], tr, a0, nz, 100, n_steps, sampler)
    for use_graph in (False, True):
        got = e.idm_sample(_f32(tr), a_init=_f32(a0), step_noise=_f32(nz) if sampler == "ddpm" else None,
                           sampler=sampler, n_steps=n_steps, use_graph=use_graph).cpu().numpy()
        assert_close(got, ref, 1e-4, f"idm {sampler}/{n_steps} graph={use_graph}")


def test_idm_philox_sharding_invariance(setup):
    e, D = setup["eng"], setup["D"]
    tr = _f32(rng(9).uniform(-1, 1, (32, 2 * D)))
    full = e.idm_sample(tr, seed=4)
    lo = e.idm_sample(tr[:12], seed=4, row_offset=0)
    hi = e.idm_sample(tr[12:], seed=4, row_offset=12)
    assert torch.equal(full[:12], lo) and torch.equal(full[12:], hi)
    assert not torch.equal(full, e.idm_sample(tr, seed=5))


def test_normalize_kernels_match_reference_expressions(setup):
    e = setup["eng"]
    g = rng(1)
    lo, hi = np.array([-0.162, -0.05, 0.728], np.float32), np.array([0.068, 0.058, 1.141], np.float32)
    v = g.uniform(-0.3, 1.3, (5, 2, 3)).astype(np.float32)
    assert_close(e.normalize_bounds(_f32(v), lo, hi, True).cpu().numpy(), np64.normalize_bounds(v, lo, hi), 1e-6, "normalize")
    n = g.uniform(-1.5, 1.5, (5, 2, 3)).astype(np.float32)
    assert_close(e.normalize_bounds(_f32(n), lo, hi, False).cpu().numpy(), np64.unnormalize_bounds(n, lo, hi), 1e-6, "unnormalize")
    img = g.integers(0, 256, (2, 1, 8, 8, 3)).astype(np.float32)
    assert_close(e.normalize_bounds(_f32(img), [0], [255], True).cpu().numpy(), img / 255 * 2 - 1, 1e-6, "image")
    assert_close(e.normalize_bounds(_f32(n), [-1], [1], 2).cpu().numpy(), np.clip(n, -1, 1), 0, "clip")


# ---------------------------------------------------------------------------------------------
# the agent surface
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def agents(setup):
    from latent_diffusion_planning_amd.agent import LDPAgent, ParamState
    data = cfgs.RM_LIFT if setup["name"] == "rm" else cfgs.ALOHA_CUBE
    ag = LDPAgent.create(0, None, data["shape_meta"], **cfgs.agent_kwargs(data))
    # load the synthetic "checkpoint" the way load_snapshot does (train_bc.py:210-240)
    ag = ag.replace(planner_state=ag.planner_state.replace(params=setup["pp"], ema_params=setup["pp"]),
                    idm_state=ag.idm_state.replace(params=setup["ip"], ema_params=setup["ip"]))
    cfg = dict(ag.config)
    orc = np64.AgentOracle(cfg, setup["pp"], setup["ip"], None, data["obs_normalization"],
                           planner_sample_fn=_planner_fn, idm_sample_fn=_idm_fn)
    return ag, orc, data


@pytest.mark.parametrize("B", [1, 5])
def test_sample_viz_matches_agent_oracle(agents, B):
    ag, orc, data = agents
    D, A = ag.config["obs_dim"], ag.config["action_dim"]
    batch = cfgs.synth_latent_batch(data, B, 1, 50 + B)
    g = rng(60 + B)
    noise = dict(x_init=g.standard_normal((B, 8, D)), x_noise=g.standard_normal((100, B, 8, D)),
                 a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)))
    ref_a, ref_m = orc.sample_viz(batch, noise["x_init"], noise["x_noise"], noise["a_init"], noise["a_noise"],
                                  decode=False)
    act, met = ag.sample_viz(batch, 0, noise={k: _f32(v) for k, v in noise.items()})
    assert act.shape == (B, 4, A) and met["plan"].shape == (B, 5, D)
    assert set(met) >= {"plan", "plan_viz"} and "plan_mse" not in met
    assert_close(met["plan"].cpu().numpy(), ref_m["plan"], 1e-4, "plan")
    scale = 1.0
    if "min" in data["obs_normalization"]["actions"]:
        scale = float(np.max(np.asarray(data["obs_normalization"]["actions"]["max"]) -
                             np.asarray(data["obs_normalization"]["actions"]["min"])))
    assert_close(act.cpu().numpy(), ref_a, 2e-4 * max(scale, 1.0), "action")
    a0 = np.array(act.cpu())[0]                          # harness indexing (rm_env_utils.py:188-192)
    assert a0.shape == (4, A)


def test_training_batch_gets_plan_mse_and_sample_action(agents):
    ag, orc, data = agents
    D, A = ag.config["obs_dim"], ag.config["action_dim"]
    B, H = 3, 9
    batch = cfgs.synth_latent_batch(data, B, H, 77, with_actions=True)
    g = rng(78)
    noise = dict(x_init=g.standard_normal((B, 8, D)), x_noise=g.standard_normal((100, B, 8, D)),
                 a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)))
    ref_a, ref_m = orc.sample_viz(batch, noise["x_init"], noise["x_noise"], noise["a_init"], noise["a_noise"], decode=False)
    act, met = ag.sample(batch, 0, noise={k: _f32(v) for k, v in noise.items()})
    assert "plan_mse" in met
    assert abs(float(met["plan_mse"]) - float(ref_m["plan_mse"])) < 1e-4
    # sample_action: ground-truth plan, IDM rows = B*(H-1)
    n2 = dict(a_init=g.standard_normal((B * (H - 1), A)), a_noise=g.standard_normal((100, B * (H - 1), A)))
    ref = orc.sample_action(batch, n2["a_init"], n2["a_noise"])
    got = ag.sample_action(batch, 0, noise={k: _f32(v) for k, v in n2.items()})
    assert got.shape == (B, H - 1, A)
    assert_close(got.cpu().numpy(), ref, 3e-4, "sample_action")
    # sample_action_from_plan: externally supplied next states
    obs_only = {"obs": {k: v[:, :4] for k, v in batch["obs"].items()}}
    nxt = g.uniform(-1, 1, (B, 4, D))
    n3 = dict(a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)))
    ref = orc.sample_action_from_plan(obs_only, nxt, n3["a_init"], n3["a_noise"])
    got = ag.sample_action_from_plan(obs_only, _f32(nxt), 0, noise={k: _f32(v) for k, v in n3.items()})
    assert_close(got.cpu().numpy(), ref, 3e-4, "sample_action_from_plan")


def test_agent_batch_contract_and_variable_batch(agents):
    ag, _, data = agents
    batch = cfgs.synth_latent_batch(data, 2, 1, 5)
    bad = dict(batch)
    bad["extra"] = 1
    with pytest.raises(AssertionError):                         # agent/ldp_agent.py:439
        ag.sample(bad, 0)
    outs = []
    for B in (4, 5, 3, 5):                                      # live-env count changes call to call
        b = cfgs.synth_latent_batch(data, B, 1, 5)
        a, m = ag.sample(b, 11)
        assert a.shape[0] == B and torch.isfinite(a).all() and torch.isfinite(m["plan"]).all()
        outs.append(a)
    assert torch.equal(outs[1], outs[3])                        # same inputs + seed -> same actions
    assert ag.get_action(cfgs.synth_latent_batch(data, 5, 1, 5), 11).equal(outs[1])
    with pytest.raises(NotImplementedError):
        ag.update(batch)
