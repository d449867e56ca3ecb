"""The training step on the GPU (csrc/train.hip) against the float64 autograd oracle (oracle/train.py).  -m gpu.

Tolerances: every gradient leaf within 1e-4 of the leaf's largest |gradient|; g_norm relative 1e-5; parameters
after 10 Adam steps within 1e-5 of the float64 optimiser's; the metrics keys exactly the reference's."""
import os

import numpy as np
import pytest
import torch

from latent_diffusion_planning_amd import weights as W
from oracle import train as OT
from tests import cfgs
from tests.cases import DIMS, load_case, unflat_obs
from tests.util import idm_params, make_agent, planner_params, rng, tree_digest

pytestmark = pytest.mark.gpu

D, A, T = 25, 7, 8


@pytest.fixture(scope="module")
def eng():
    from latent_diffusion_planning_amd.engine import HipEngine
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params(D=D), idm=idm_params(D=D, A=A))
    yield e
    e.close()


def _batch(B, seed, H=T + 1):
    g = rng(seed)
    obs_emb = g.uniform(-1, 1, (B, H, D)).astype(np.float32)
    actions = g.uniform(-1, 1, (B, H, A)).astype(np.float32)
    nz = dict(t_plan=g.integers(0, 100, B), noise_plan=g.standard_normal((B, H - 1, D)).astype(np.float32),
              t_idm=g.integers(0, 100, B * (H - 1)), noise_idm=g.standard_normal((B * (H - 1), A)).astype(np.float32))
    return obs_emb, actions, nz


def _leaf_report(got, ref, what):
    worst = (0.0, None)
    for k, r in ref.items():
        g = got[k]
        assert g.shape == r.shape, (k, g.shape, r.shape)
        scale = float(np.abs(r).max())
        err = float(np.abs(g.astype(np.float64) - r).max())
        rel = err / max(scale, 1e-30)
        if rel > worst[0]:
            worst = (rel, k)
    print(f"{what}: worst leaf error / leaf max = {worst[0]:.2e} at {worst[1]}")
    bad = {k: float(np.abs(got[k].astype(np.float64) - r).max() / max(float(np.abs(r).max()), 1e-30)) for k, r in ref.items()
           if np.abs(got[k].astype(np.float64) - r).max() > 1e-4 * max(float(np.abs(r).max()), 1e-30) + 1e-12}
    assert not bad, f"{what}: {len(bad)} of {len(ref)} leaves off: " + ", ".join(f"{k}: {v:.2e}" for k, v in list(bad.items())[:12])


def _relu_margins(ip, obs_emb, actions, nz):
    """Per (sample of the batch): the smallest |pre-activation| any ReLU of the IDM sees for its rows, in float64.  A ReLU whose input is within
    fp32 round-off of zero gates differently in fp32 than in float64 and moves that row's whole contribution to the gradient (about 1 / rows
    of a leaf): not an error of the kernels, but not comparable at 1e-4 either -- such samples are left out of the comparison batches."""
    from oracle import torch32
    import torch.nn.functional as F
    P = torch32.TorchParams(ip, dtype=torch.float64)
    emb, act = torch.tensor(obs_emb, dtype=torch.float64), torch.tensor(actions, dtype=torch.float64)
    s = torch.cat([emb[:, :-1], emb[:, 1:]], dim=-1).reshape(-1, 2 * D)
    a = OT._add_noise(act[:, :-1].reshape(-1, A), torch.tensor(nz["noise_idm"], dtype=torch.float64), nz["t_idm"], 100)
    k = torch.tensor(nz["t_idm"])
    arg = k[:, None].float() * torch32._freqs(256, "cpu")[None, :]
    c = F.mish(F.linear(torch.cat([torch.cos(arg), torch.sin(arg)], -1).double(), P.t("MLP_0/Dense_0/kernel").t(), P.t("MLP_0/Dense_0/bias")))
    c = F.linear(c, P.t("MLP_0/Dense_1/kernel").t(), P.t("MLP_0/Dense_1/bias"))
    h = F.linear(torch.cat([a, s, c], -1), P.t("MLPResNet_0/Dense_0/kernel").t(), P.t("MLPResNet_0/Dense_0/bias"))
    m = torch.full((h.shape[0],), float("inf"), dtype=torch.float64)
    for i in range(3):
        p = f"MLPResNet_0/MLPResNetBlock_{i}"
        y = F.layer_norm(h, (256,), P.t(f"{p}/LayerNorm_0/scale"), P.t(f"{p}/LayerNorm_0/bias"), eps=1e-6)
        u = F.linear(y, P.t(f"{p}/Dense_0/kernel").t(), P.t(f"{p}/Dense_0/bias"))
        m = torch.minimum(m, u.abs().min(dim=1).values)
        h = h + F.linear(F.relu(u), P.t(f"{p}/Dense_1/kernel").t(), P.t(f"{p}/Dense_1/bias"))
    m = torch.minimum(m, h.abs().min(dim=1).values)
    return m.reshape(obs_emb.shape[0], -1).min(dim=1).values.numpy()


@pytest.mark.parametrize("B", [3, 40])
def test_idm_loss_and_gradients_match_the_float64_oracle(eng, B):
    obs_emb, actions, nz = _batch(B + 12, 900 + B)
    ip = idm_params(D=D, A=A)
    keep = np.argsort(-_relu_margins(ip, obs_emb, actions, nz))[:B]                 # the B samples whose ReLU inputs stay clear of zero
    keep.sort()
    rows = (keep[:, None] * T + np.arange(T)[None, :]).reshape(-1)
    obs_emb, actions = obs_emb[keep], actions[keep]
    nz = dict(nz, t_idm=nz["t_idm"][rows], noise_idm=nz["noise_idm"][rows])
    print(f"smallest |ReLU input| among the {B} samples kept: {_relu_margins(ip, obs_emb, actions, nz).min():.1e}")
    ref = OT.loss_and_grads(None, ip, obs_emb, actions, t_idm=nz["t_idm"], noise_idm=nz["noise_idm"], alpha_idm=0.7)
    eng.train_init(["idm"])
    s = np.concatenate([obs_emb[:, :-1], obs_emb[:, 1:]], axis=-1).reshape(-1, 2 * D)
    a0 = actions[:, :-1].reshape(-1, A)
    loss = eng.train_idm_grad(torch.tensor(s), torch.tensor(a0), torch.tensor(nz["noise_idm"]), nz["t_idm"], alpha=0.7)
    assert abs(float(loss) - ref["idm_loss"]) <= 1e-5 * max(1.0, abs(ref["idm_loss"])), (float(loss), ref["idm_loss"])
    got = eng.train_read("idm", eng.TRAIN_GRADS, W.idm_shapes(W.IDMSpec(D, A)))
    _leaf_report(got, ref["grads_idm"], f"IDM gradients, {B * T} rows")
    gn = float(eng.train_grad_norm(["idm"]))
    assert abs(gn - ref["g_norm"]) <= 1e-5 * ref["g_norm"], (gn, ref["g_norm"])


@pytest.mark.parametrize("B", [3, 33])
def test_planner_loss_and_gradients_match_the_float64_oracle(eng, B):
    obs_emb, actions, nz = _batch(B, 950 + B)
    pp = planner_params(D=D)
    ref = OT.loss_and_grads(pp, None, obs_emb, actions, t_plan=nz["t_plan"], noise_plan=nz["noise_plan"], alpha_planner=1.3)
    eng.train_init(["planner"])
    loss = eng.train_planner_grad(torch.tensor(obs_emb[:, 1:].copy()), torch.tensor(nz["noise_plan"]), nz["t_plan"], torch.tensor(obs_emb[:, 0].copy()), alpha=1.3)
    assert abs(float(loss) - ref["plan_loss"]) <= 1e-5 * max(1.0, abs(ref["plan_loss"])), (float(loss), ref["plan_loss"])
    got = eng.train_read("planner", eng.TRAIN_GRADS, W.planner_shapes(W.PlannerSpec(D, D)))
    _leaf_report(got, ref["grads_planner"], f"planner gradients, {B} plans")
    gn = float(eng.train_grad_norm(["planner"]))
    assert abs(gn - ref["g_norm"]) <= 1e-5 * ref["g_norm"], (gn, ref["g_norm"])


def test_ten_adam_steps_of_the_idm_follow_the_float64_optimiser(eng):
    """optax.adam(warmup_cosine_decay_schedule) restated in oracle/train.py, the reference's hyper-parameters (train_bc.yaml:14-16: lr 1e-4,
    end 1e-6, warmup 1000): parameters after 1 and after 10 steps, Adam moments, step count."""
    ip = idm_params(D=D, A=A)
    shapes = W.idm_shapes(W.IDMSpec(D, A))
    orc = OT.TrainOracle(None, ip, lr=1e-4, end_lr=1e-6, idm_lr=1e-4, idm_end_lr=1e-6, warmup_steps=1000, decay_steps=500000)
    eng.load_params(idm=ip)
    eng.train_init(["idm"])
    for step in range(10):
        obs_emb, actions, nz = _batch(8, 1200 + step)
        m = orc.update_step(obs_emb, actions, use_planner=False, use_idm=True, t_idm=nz["t_idm"], noise_idm=nz["noise_idm"])
        s = np.concatenate([obs_emb[:, :-1], obs_emb[:, 1:]], axis=-1).reshape(-1, 2 * D)
        loss = eng.train_idm_grad(torch.tensor(s), torch.tensor(actions[:, :-1].reshape(-1, A).copy()), torch.tensor(nz["noise_idm"]), nz["t_idm"])
        gn = float(eng.train_grad_norm(["idm"]))
        assert eng.train_step_count("idm") == step
        eng.train_apply("idm", orc.i_sched(step))
        assert abs(float(loss) - m["idm_loss"]) <= 2e-5 * max(1.0, m["idm_loss"]) and abs(gn - m["g_norm"]) <= 1e-4 * m["g_norm"], (step, float(loss), m["idm_loss"], gn, m["g_norm"])
        if step in (0, 9):
            got = eng.train_read("idm", eng.TRAIN_PARAMS, shapes)
            worst = max(float(np.abs(got[k].astype(np.float64) - orc.ip[k]).max()) for k in shapes)
            moved = max(float(np.abs(np.asarray(ip[k], np.float64) - orc.ip[k]).max()) for k in shapes)
            n = sum(int(np.prod(v)) for v in shapes.values())
            mean_err = sum(float(np.abs(got[k].astype(np.float64) - orc.ip[k]).sum()) for k in shapes) / n
            mean_move = sum(float(np.abs(np.asarray(ip[k], np.float64) - orc.ip[k]).sum()) for k in shapes) / n
            print(f"after {step + 1} Adam steps: max |param - float64 optimiser| = {worst:.2e} (the parameters moved by up to {moved:.2e}); "
                  f"mean error {mean_err:.2e} of a mean move of {mean_move:.2e}")
            # Adam's first steps are sign-like (|update| = lr whatever |g|): an element whose gradient is at the fp32 round-off level of its
            # leaf may step the other way, so the MAX error is bounded by the distance moved, not by the gradient accuracy; the mean is
            assert worst <= 1e-5 and mean_err <= 0.02 * mean_move
    assert eng.train_step_count("idm") == 10
    mu = eng.train_read("idm", eng.TRAIN_MU, shapes)
    for k in shapes:            # (mean, not max: a ReLU whose pre-activation sits at the fp32 round-off level may gate differently than in float64)
        ref = orc.i_state["mu"][k]
        assert np.abs(mu[k].astype(np.float64) - ref).mean() <= 1e-4 * max(np.abs(ref).max(), 1e-12) + 1e-12, k


def test_publish_hands_the_trained_parameters_to_the_sampling_path(eng):
    ip = idm_params(D=D, A=A)
    eng.load_params(idm=ip)
    g = rng(31)
    s, a = torch.tensor(g.uniform(-1, 1, (12, 2 * D)), dtype=torch.float32), torch.tensor(g.standard_normal((12, A)), dtype=torch.float32)
    before = eng.idm_forward(s, a, 17).clone()
    eng.train_init(["idm"])
    obs_emb, actions, nz = _batch(8, 77)
    ss = np.concatenate([obs_emb[:, :-1], obs_emb[:, 1:]], axis=-1).reshape(-1, 2 * D)
    eng.train_idm_grad(torch.tensor(ss), torch.tensor(actions[:, :-1].reshape(-1, A).copy()), torch.tensor(nz["noise_idm"]), nz["t_idm"])
    eng.train_apply("idm", 1e-3)
    assert torch.equal(eng.idm_forward(s, a, 17), before)                 # the sampling path still holds the old weights ...
    eng.train_publish(["idm"])
    after = eng.idm_forward(s, a, 17)
    assert not torch.equal(after, before)                                 # ... until they are published
    from oracle import torch32
    new = eng.train_read("idm", eng.TRAIN_PARAMS, W.idm_shapes(W.IDMSpec(D, A)))
    ref = torch32.idm_forward(torch32.TorchParams(new, dtype=torch.float64), s.double(), a.double(), 17).numpy()
    assert np.abs(after.cpu().numpy() - ref).max() <= 2e-5


# ---- the agent surface: LDPAgent.update / update_mixed against the committed goldens (tests/golden/agent_update_*.npz) ---------------------
REF_KEYS = {"plan_loss", "idm_loss", "loss", "emb_min", "emb_max", "emb_mean", "emb_std", "action_min", "action_max", "g_norm", "planner_lr",
            "planner_step", "idm_lr", "idm_step"}


def _step_inputs(inp, s, mixed=False):
    pre = f"s{s}_"
    batch = unflat_obs({k[len(pre):]: v for k, v in inp.items() if k.startswith(pre) and not k.startswith(pre + "mixed_")})
    nz = {k: inp[pre + k] for k in ("noise_plan", "noise_idm")}
    nz["t_plan"], nz["t_idm"] = inp[pre + "t_plan"].astype(np.int64), inp[pre + "t_idm"].astype(np.int64)
    mb = unflat_obs({k[len(pre) + 6:]: v for k, v in inp.items() if k.startswith(pre + "mixed_")}) if mixed else None
    return batch, mb, nz


def _digest_close(got_tree, exp, seed, what, rel):
    got = tree_digest(got_tree, seed)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    scale = np.maximum(exp[:, 1:2], 1e-30)                      # the leaf's max |x| (column 1 of a digest)
    err = np.abs(got - exp) / scale
    err[:, 0] /= np.sqrt(np.maximum(1.0, 1.0))                  # (norms and projections are on the leaf's own scale too)
    i = np.unravel_index(np.argmax(err), err.shape)
    print(f"{what}: worst digest entry off by {err.max():.2e} of its leaf's max (leaf {list(got_tree)[i[0]]}, entry {i[1]})")
    assert err[:, 3:].max() <= rel, what


@pytest.mark.parametrize("name", ["rm", "aloha"])
def test_agent_update_matches_golden(name):
    """Ten `agent, metrics = agent.update(batch, rng, step)` calls (train_bc.py:107) on the committed inputs: metrics keys exactly the
    reference's (agent/ldp_agent.py:156-178, 253-271), losses / g_norm / learning rates of every step, per-leaf digests of the step-0
    gradients and of the parameters after 1 and after 10 steps."""
    D, A, data = DIMS[name]
    inp, exp = load_case(f"agent_update_{name}")
    pp, ip = planner_params(D=D), idm_params(D=D, A=A)
    ag, _ = make_agent(name, pp, ip)
    n = len(exp["g_norm"])
    first = ag
    for s in range(n):
        batch, _, nz = _step_inputs(inp, s)
        prev = ag
        ag, m = ag.update(batch, 100 + s, s, noise=nz)
        assert set(m) == REF_KEYS | {f"{k}_{e}" for k in batch["obs"] for e in ("min", "max")}, sorted(set(m) ^ REF_KEYS)
        assert m["planner_step"] == s and m["idm_step"] == s and ag.planner_state.step == s + 1 and ag.idm_state.step == s + 1
        for k in ("plan_loss", "idm_loss", "g_norm"):
            assert abs(float(m[k]) - exp[k][s]) <= 2e-5 * max(1.0, abs(exp[k][s])) * (1 if k != "g_norm" else 5), (s, k, float(m[k]), exp[k][s])
        assert abs(float(m["loss"]) - (exp["plan_loss"][s] + exp["idm_loss"][s])) <= 4e-5 * max(1.0, exp["plan_loss"][s] + exp["idm_loss"][s])
        for k in ("planner_lr", "idm_lr"):
            assert abs(float(m[k]) - exp[k][s]) <= 1e-6 * exp[k][s], (s, k)
        if s == 0:
            for k in ("emb_min", "emb_max", "emb_mean", "emb_std", "action_min", "action_max"):
                assert abs(float(m[k]) - float(exp[k])) <= 1e-5, k
            eng = ag._engine
            _digest_close(eng.train_read("planner", eng.TRAIN_GRADS, W.planner_shapes(ag._planner_spec)), exp["grads_planner"], 11, "planner gradients, step 0", 1e-4)
            _digest_close(eng.train_read("idm", eng.TRAIN_GRADS, W.idm_shapes(ag._idm_spec)), exp["grads_idm"], 12, "IDM gradients, step 0", 1e-4)
            for tree, key, seed in ((ag.planner_state.params, "planner_after_1", 13), (ag.idm_state.params, "idm_after_1", 14)):
                got = tree_digest(tree, seed)
                assert np.abs(got[:, 3:] - exp[key][:, 3:]).max() <= 1e-5, key
    for tree, key, seed in ((ag.planner_state.params, "planner_after_n", 15), (ag.idm_state.params, "idm_after_n", 16)):
        got = tree_digest(tree, seed)
        worst = np.abs(got[:, 3:] - exp[key][:, 3:]).max()
        print(f"{name}: {key}: max |param - float64 optimiser| over the digests = {worst:.2e} (parameters moved by up to {float(exp['planner_moved']):.2e})")
        assert worst <= 1e-5, key
    # a superseded device state cannot be read any more; the newest can, and its optimiser state travels with it
    with pytest.raises(RuntimeError, match="superseded"):
        prev.planner_state.params
    o = ag.idm_state.opt_state
    assert o["count"] == n and set(o["mu"]) == set(ip) and float(np.abs(o["nu"]["MLPResNet_0/Dense_0/kernel"]).max()) > 0
    # the trained agent samples with the trained weights (published to the sampling path on demand), bit-equal to an agent built from them
    sb = cfgs.synth_latent_batch(data, 3, 1, 5)
    act = np.array(ag.sample(sb, 9)[0])
    ag2, _ = make_agent(name, ag.planner_state.params, ag.idm_state.params)
    assert np.array_equal(np.array(ag2.sample(sb, 9)[0]), act)
    # and a restored TrainState continues exactly where the first left off: params + opt_state + step through .replace, as load_snapshot would
    restored = ag2.replace(planner_state=ag2.planner_state.replace(opt_state=ag.planner_state.opt_state, step=ag.planner_state.step),
                           idm_state=ag2.idm_state.replace(opt_state=ag.idm_state.opt_state, step=ag.idm_state.step))
    batch, _, nz = _step_inputs(inp, 0)
    a1, m1 = ag.update(batch, 1, n, noise=nz)
    a2, m2 = restored.update(batch, 1, n, noise=nz)
    assert float(m1["g_norm"]) == float(m2["g_norm"]) and m2["planner_step"] == n and float(m1["planner_lr"]) == float(m2["planner_lr"])
    p1, p2 = a1.idm_state.params, a2.idm_state.params
    assert all(np.array_equal(p1[k], p2[k]) for k in p1)
    ag._engine.close()
    ag2._engine.close()


def test_agent_update_mixed_and_gating():
    """update_mixed (agent/ldp_agent.py:274-323): planner on `batch`, IDM on `mixed_batch`; and the schedule gates (update_planner_every = 2:
    odd steps train the IDM only and report planner_lr = planner_step = noise_diff = 0, :259-263)."""
    D, A, data = DIMS["rm"]
    inp, exp = load_case("agent_update_mixed_rm")
    ag, _ = make_agent("rm", planner_params(D=D), idm_params(D=D, A=A))
    for s in range(len(exp["g_norm"])):
        batch, mb, nz = _step_inputs(inp, s, mixed=True)
        ag, m = ag.update_mixed(batch, mb, 3, s, noise=nz)
        for k in ("plan_loss", "idm_loss", "g_norm"):
            assert abs(float(m[k]) - exp[k][s]) <= 1e-4 * max(1.0, abs(exp[k][s])), (s, k, float(m[k]), exp[k][s])
    got = tree_digest(ag.idm_state.params, 16)
    assert np.abs(got[:, 3:] - exp["idm_after_n"][:, 3:]).max() <= 1e-5
    ag.config["update_planner_every"] = 2
    batch, mb, nz = _step_inputs(inp, 0, mixed=True)
    before = ag.planner_state
    ag2, m = ag.update(batch, 0, 5, noise=nz)
    assert ag2.planner_state is before and ag2.idm_state.step == ag.idm_state.step + 1
    assert m["planner_lr"] == 0 and m["planner_step"] == 0 and m["noise_diff"] == 0 and float(m["plan_loss"]) == 0.0 and float(m["idm_lr"]) > 0
    ag._engine.close()


def test_in_launch_split_k_finish_equals_the_reduce_launch_bitwise(eng):
    """train_fuse_reduce: the last work-group of a split-K tile adds the partials up in split order, reading its own back from memory -- the bits of
    every gradient leaf equal those of the separate reduce launch, call after call (a lost ticket or a stale partial would show here)."""
    obs_emb, actions, nz = _batch(64, 77)
    x0 = torch.tensor(obs_emb[:, 1:]); cond = torch.tensor(obs_emb[:, 0])
    s2 = np.concatenate([obs_emb[:, :-1], obs_emb[:, 1:]], axis=-1).reshape(-1, 2 * D)
    a0 = actions[:, :-1].reshape(-1, A).copy()
    eng.train_init(["planner", "idm"])
    shapes_p, shapes_i = W.planner_shapes(W.PlannerSpec(D, D)), W.idm_shapes(W.IDMSpec(D, A))

    def grads(fuse):
        eng.set_option("train_fuse_reduce", fuse)
        lp = float(eng.train_planner_grad(x0, torch.tensor(nz["noise_plan"]), nz["t_plan"], cond))
        li = float(eng.train_idm_grad(torch.tensor(s2), torch.tensor(a0), torch.tensor(nz["noise_idm"]), nz["t_idm"]))
        gn = float(eng.train_grad_norm(["planner", "idm"]))
        return lp, li, gn, eng.train_read("planner", eng.TRAIN_GRADS, shapes_p), eng.train_read("idm", eng.TRAIN_GRADS, shapes_i)
    eng.set_option("train_group_proj", 0)      # (the grouped projection launches have no C-layout workspace: without the in-launch finish they never split K)
    ref = grads(0)
    try:
        for rep in range(6):
            got = grads(1)
            assert got[:3] == ref[:3], (rep, got[:3], ref[:3])
            for k in ref[3]:
                assert np.array_equal(got[3][k], ref[3][k]), (rep, k)
            for k in ref[4]:
                assert np.array_equal(got[4][k], ref[4][k]), (rep, k)
    finally:
        eng.set_option("train_fuse_reduce", 1)
        eng.set_option("train_group_proj", 1)
    # the grouped launches of the projection blocks (two convolutions over one input / two data gradients into it as ONE launch each) against the
    # separate launches: the same sums in another order
    eng.set_option("train_group_proj", 0)
    sep = grads(1)
    eng.set_option("train_group_proj", 1)
    grp = grads(1)
    worst = 0.0
    for a, b in ((sep[3], grp[3]), (sep[4], grp[4])):
        for k in a:
            worst = max(worst, float(np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-30)))
    print(f"grouped vs separate projection launches: worst leaf difference {worst:.2e} of the leaf's max")
    assert worst <= 5e-6 and abs(sep[0] - grp[0]) <= 1e-6 * abs(sep[0])


def test_training_gradients_are_bit_reproducible_at_the_reference_batch(eng):
    """256 samples (train_bc.yaml:9): every launch splits K and finishes in-launch (tickets + sc1 partial blocks), the weight gradients run on a side
    stream, the two tapes on two streams.  40 repetitions of the same step: a stale partial, a lost ticket or a missing cross-stream dependency would
    show as a differing bit in some gradient leaf; so would any atomics in a reduction."""
    obs_emb, actions, nz = _batch(256, 78)
    x0 = torch.tensor(obs_emb[:, 1:]).cuda(); cond = torch.tensor(obs_emb[:, 0]).cuda()
    s2 = torch.tensor(np.concatenate([obs_emb[:, :-1], obs_emb[:, 1:]], axis=-1).reshape(-1, 2 * D)).cuda()
    a0 = torch.tensor(actions[:, :-1].reshape(-1, A).copy()).cuda()
    npl, nid = torch.tensor(nz["noise_plan"]).cuda(), torch.tensor(nz["noise_idm"]).cuda()
    eng.train_init(["planner", "idm"])
    side = eng.aux_streams()["idm"]
    main = torch.cuda.current_stream()

    def grads():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            li = eng.train_idm_grad(s2, a0, nid, nz["t_idm"])
        lp = eng.train_planner_grad(x0, npl, nz["t_plan"], cond)
        main.wait_stream(side)
        gn = eng.train_grad_norm(["planner", "idm"])
        gp = eng.train_arena("planner", eng.TRAIN_GRADS).clone()
        gi = eng.train_arena("idm", eng.TRAIN_GRADS).clone()
        return torch.stack([lp, li, gn]).clone(), gp, gi
    ref = grads()
    assert torch.isfinite(ref[0]).all() and float(ref[1].abs().max()) > 0 and float(ref[2].abs().max()) > 0
    for rep in range(40):
        got = grads()
        for r, g, what in zip(ref, got, ("losses / norm", "planner gradient arena", "IDM gradient arena")):
            assert torch.equal(r, g), (rep, what, int((r != g).sum()))


def test_training_loop_reduces_the_loss_and_snapshots_round_trip(tmp_path):
    """The reference's loop in miniature (train_bc.py:98-125): `agent, metrics = agent.update(batch, rng, step)` on one fixed batch for 60 steps with
    a short warm-up -- both losses must fall well below their start (the whole chain: gradients, schedule, Adam, state hand-over); then
    save_snapshot (train_bc.py:203-208) -> load_snapshot (:210-240) into a fresh agent, which samples bit-equal to the trained one."""
    from latent_diffusion_planning_amd import checkpoint
    from latent_diffusion_planning_amd.agent import LDPAgent
    data = cfgs.BY_NAME["rm"]
    kw = cfgs.agent_kwargs(data)
    kw.update(lr=3e-4, idm_lr=3e-4, warmup_steps=5, decay_steps=1000)
    ag = LDPAgent.create(0, None, data["shape_meta"], **kw)
    ag = ag.replace(planner_state=ag.planner_state.replace(params=planner_params()), idm_state=ag.idm_state.replace(params=idm_params()))
    batch = cfgs.synth_latent_batch(data, 16, 9, 123, with_actions=True)
    g = rng(124)
    nz = dict(t_plan=g.integers(0, 100, 16), noise_plan=g.standard_normal((16, 8, 25)).astype(np.float32),
              t_idm=g.integers(0, 100, 128), noise_idm=g.standard_normal((128, 7)).astype(np.float32))
    hist = []
    for step in range(60):
        ag, m = ag.update(batch, 7, step, noise=nz)                   # fixed (t, noise): the objective is one deterministic function of the parameters
        if step % 10 == 0 or step == 59:
            hist.append((float(m["plan_loss"]), float(m["idm_loss"]), float(m["g_norm"])))
    print("plan_loss / idm_loss / g_norm every 10 steps:", [tuple(round(v, 4) for v in h) for h in hist])
    assert all(np.isfinite(h).all() for h in hist)
    assert hist[-1][0] < 0.5 * hist[0][0] and hist[-1][1] < 0.5 * hist[0][1], hist
    assert ag.planner_state.step == 60 and ag.idm_state.step == 60
    path = checkpoint.save_snapshot(ag, str(tmp_path / "60.ckpt"), batch=batch, cfg={"n_grad_steps": 60})
    assert os.path.exists(path)
    fresh = LDPAgent.create(1, None, data["shape_meta"], **kw)
    fresh = checkpoint.load_snapshot(fresh, str(tmp_path / "60.ckpt"))
    sb = cfgs.synth_latent_batch(data, 3, 1, 5)
    assert np.array_equal(np.array(fresh.sample(sb, 9)[0]), np.array(ag.sample(sb, 9)[0]))
    raw = checkpoint.restore(str(tmp_path / "60.ckpt"))
    assert set(raw) == {"data", "cfg", "planner_params", "idm_params"} and raw["cfg"]["n_grad_steps"] == 60
    ag._engine.close(); fresh._engine.close()


@pytest.mark.parametrize("T,oh", [(8, 2), (16, 1)])
def test_planner_gradients_at_other_horizons(T, oh):
    """obs_horizon 2 (global_cond_dim = 2 D: the FiLM layers and the conditioning branch are 306 wide, agent/ldp_agent.py:573-575) and pred_horizon 16
    (BASELINE configs[2]'s reading of rm_square: levels of 16 / 8 / 4 positions): loss and per-leaf gradients of the planner against the float64 oracle."""
    from latent_diffusion_planning_amd.engine import HipEngine
    G = D * oh
    pp = planner_params(D=D, G=G)
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=G, pred_horizon=T, action_horizon=4)
    e.load_params(planner=pp)
    e.train_init(["planner"])
    B = 5
    g = rng(300 + T + oh)
    emb = g.uniform(-1, 1, (B, oh + T, D)).astype(np.float32)
    t, nz = g.integers(0, 100, B), g.standard_normal((B, T, D)).astype(np.float32)
    ref = OT.loss_and_grads(pp, None, emb, np.zeros((B, oh + T, A)), t_plan=t, noise_plan=nz, obs_horizon=oh)
    loss = float(e.train_planner_grad(torch.tensor(emb[:, oh:]), torch.tensor(nz), t, torch.tensor(emb[:, :oh].reshape(B, -1))))
    assert abs(loss - ref["plan_loss"]) <= 1e-5 * max(1.0, ref["plan_loss"])
    got = e.train_read("planner", e.TRAIN_GRADS, W.planner_shapes(W.PlannerSpec(D, G)))
    _leaf_report(got, ref["grads_planner"], f"planner gradients, T = {T}, obs_horizon = {oh}")
    gn = float(e.train_grad_norm(["planner"]))
    assert abs(gn - ref["g_norm"]) <= 1e-4 * ref["g_norm"]
    e.close()
