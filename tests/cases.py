"""Golden cases of the 100-step loops: seeded inputs + the oracle evaluation that produced
tests/golden/*.npz (run `python tests/golden/make_golden.py`).  The GPU parity tests replay the
stored inputs through the HIP path and compare with the stored outputs, so the expensive
float64 loops are not recomputed on the GPU box.  Weights are regenerated from their seeds
(weights.init_*_params), they are too large to store."""
import os

import numpy as np
import torch

from tests import cfgs
from tests.util import idm_params, idm_params_heavy, planner_params, planner_params_heavy, rng

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


# The oracle is imported only where a golden is COMPUTED (tests/golden/make_golden.py): loading a case
# (load_case: what the -m gpu tests and tools/parity_margin.py do) never executes anything under oracle/.
def planner_fn(params, obs_cond, x_init, step_noise, n_train, n_steps, sampler, dtype=torch.float64):
    """float64 torch restatement of the planner loop (any pred_horizon: T comes with x_init).  dtype=torch.float32: the same code in
    the reference's own precision (what `ref32_err` of the trained-like cases is measured with)."""
    from oracle import torch32
    P = torch32.TorchParams(params, dtype=dtype)
    t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)      # noqa: E731
    return torch32.planner_sample(P, t(obs_cond), t(x_init), t(step_noise),
                                  n_train=n_train, n_steps=n_steps, sampler=sampler).double().numpy()


def idm_fn(params, trans, a_init, step_noise, n_train, n_steps, sampler, dtype=torch.float64):
    from oracle import torch32
    P = torch32.TorchParams(params, dtype=dtype)
    t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)      # noqa: E731
    return torch32.idm_sample(P, t(trans), t(a_init), t(step_noise),
                              n_train=n_train, n_steps=n_steps, sampler=sampler).double().numpy()


DIMS = {"rm": (25, 7, cfgs.RM_LIFT), "aloha": (30, 14, cfgs.ALOHA_CUBE),
        "rm_square": (25, 7, cfgs.RM_SQUARE), "rm_can": (25, 7, cfgs.RM_CAN)}


# ---- case builders: each returns (inputs: dict of arrays, compute: () -> dict of arrays) ---------
def planner_loop(sampler, n_steps, B=3, T=8, D=25):
    g = rng(200 + n_steps + (7 if sampler == "ddim" else 0))
    inp = dict(cond=g.uniform(-1, 1, (B, D)), x0=g.standard_normal((B, T, D)),
               nz=g.standard_normal((n_steps, B, T, D)))

    def compute():
        return dict(plan=planner_fn(planner_params(D=D), inp["cond"], inp["x0"], inp["nz"], 100, n_steps, sampler))
    return inp, compute


def planner_loop_heavy(sampler, n_steps, B=3, T=8, D=25, wide=False):
    """The planner loop on a TRAINED-LIKE weight set (tests/util.py trained_like: norm scales log-normal over 1e-2 .. 1e2, biases O(10),
    one output channel of every kernel x 100, output head re-calibrated to O(1) eps; wide: also the un-normalised convs of the residual
    stream, which then runs at 2e7).  `ref32_err` = the error of the SAME restatement run
    in float32 (the reference's precision) against the float64 result: the floor no fp32 implementation can be asked to beat."""
    g = rng(1200 + n_steps + (7 if sampler == "ddim" else 0) + T)
    inp = dict(cond=g.uniform(-1, 1, (B, D)), x0=g.standard_normal((B, T, D)),
               nz=g.standard_normal((n_steps, B, T, D)))

    def compute():
        pp = planner_params_heavy(D=D, wide=wide)
        nz = inp["nz"] if sampler == "ddpm" else None
        plan = planner_fn(pp, inp["cond"], inp["x0"], nz, 100, n_steps, sampler)
        p32 = planner_fn(pp, inp["cond"], inp["x0"], nz, 100, n_steps, sampler, dtype=torch.float32)
        return dict(plan=plan, ref32_err=np.abs(p32 - plan).max())
    return inp, compute


def idm_loop_heavy(cfg, sampler, n_steps, R=12):
    D, A, _ = DIMS[cfg]
    g = rng(1400 + n_steps + D)
    inp = dict(tr=g.uniform(-1, 1, (R, 2 * D)), a0=g.standard_normal((R, A)), nz=g.standard_normal((n_steps, R, A)))

    def compute():
        ip = idm_params_heavy(D=D, A=A)
        nz = inp["nz"] if sampler == "ddpm" else None
        act = idm_fn(ip, inp["tr"], inp["a0"], nz, 100, n_steps, sampler)
        a32 = idm_fn(ip, inp["tr"], inp["a0"], nz, 100, n_steps, sampler, dtype=torch.float32)
        return dict(act=act, ref32_err=np.abs(a32 - act).max())
    return inp, compute


def bench_rows(B=256, T=8, D=25):
    """bench.py's workload (B=256, 100-step DDIM): expected plans of rows 0, 1, 254, 255."""
    g = rng(256)
    inp = dict(cond=g.uniform(-1, 1, (B, D)), x0=g.standard_normal((B, T, D)))
    rows = np.array([0, 1, B - 2, B - 1])

    def compute():
        return dict(rows=rows.astype(np.float64),
                    plan=planner_fn(planner_params(D=D), inp["cond"][rows], inp["x0"][rows], None, 100, 100, "ddim"))
    return inp, compute


def idm_loop(cfg, sampler, n_steps, R=12):
    D, A, _ = DIMS[cfg]
    g = rng(400 + n_steps + D)
    inp = dict(tr=g.uniform(-1, 1, (R, 2 * D)), a0=g.standard_normal((R, A)), nz=g.standard_normal((n_steps, R, A)))

    def compute():
        return dict(act=idm_fn(idm_params(D=D, A=A), inp["tr"], inp["a0"], inp["nz"], 100, n_steps, sampler))
    return inp, compute


def _agent_oracle(cfg, T=8, vae=None):
    from oracle import np64
    D, A, data = DIMS[cfg]
    conf = dict(planner_n_diffusion_steps=100, idm_n_diffusion_steps=100, lowdim_obs=data["lowdim_obs"],
                rgb_obs=data["rgb_obs"], obs_horizon=1, pred_horizon=T, action_horizon=4, obs_dim=D,
                action_dim=A, vae_feature_dim=16)
    return np64.AgentOracle(conf, planner_params(D=D), idm_params(D=D, A=A), vae, data["obs_normalization"],
                            planner_sample_fn=planner_fn, idm_sample_fn=idm_fn)


VAE_SEED = 2


def vae_params():
    from latent_diffusion_planning_amd import weights as W
    from tests.util import _cache
    if "vae" not in _cache:
        _cache["vae"] = W.init_vae_params(seed=VAE_SEED)
    return _cache["vae"]


def _flat_obs(batch):
    out = {f"obs__{k}": v for k, v in batch["obs"].items()}
    if "actions" in batch:
        out["actions"] = batch["actions"]
    return out


def unflat_obs(inp):
    batch = {"obs": {k[5:]: inp[k] for k in inp if k.startswith("obs__")}}
    if "actions" in inp:
        batch["actions"] = inp["actions"]
    return batch


def agent_sample_viz(cfg, B):
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, 1, 50 + B)
    g = rng(60 + B + D)
    inp = dict(x_init=g.standard_normal((B, 8, D)), x_noise=g.standard_normal((100, B, 8, D)),
               a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)), **_flat_obs(batch))

    def compute():
        a, m = _agent_oracle(cfg).sample_viz(batch, inp["x_init"], inp["x_noise"], inp["a_init"], inp["a_noise"],
                                            decode=False)
        return dict(action=a, plan=m["plan"])
    return inp, compute


def agent_sample_viz_ddim(cfg="rm", B=3, n_steps=50):
    """BASELINE configs[4]'s per-GPU path through the agent surface: 50-step DDIM planner + 50-step DDIM IDM
    (deterministic: only the initial states are random)."""
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, 1, 350 + B)
    g = rng(360 + B + D)
    inp = dict(x_init=g.standard_normal((B, 8, D)), a_init=g.standard_normal((B * 4, A)), **_flat_obs(batch))

    def compute():
        a, m = _agent_oracle(cfg).sample_viz(batch, inp["x_init"], None, inp["a_init"], None, decode=False,
                                            sampler="ddim", n_steps=n_steps)
        return dict(action=a, plan=m["plan"])
    return inp, compute


def agent_sample_viz_t16(cfg="rm", B=2, T=16):
    """BASELINE configs[2] (rm_square read as pred_horizon 16): joint planner + IDM."""
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, 1, 150 + B)
    g = rng(160 + B + D)
    inp = dict(x_init=g.standard_normal((B, T, D)), x_noise=g.standard_normal((100, B, T, D)),
               a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)), **_flat_obs(batch))

    def compute():
        a, m = _agent_oracle(cfg, T=T).sample_viz(batch, inp["x_init"], inp["x_noise"], inp["a_init"],
                                                  inp["a_noise"], decode=False)
        return dict(action=a, plan=m["plan"])
    return inp, compute


def agent_raw_image(cfg="aloha", B=2):
    """Raw [0,255] camera frames in (aloha: wrist64_image, latent bounds +-5.5): normalise -> StableVAE
    encode -> latent normalise -> DDPM-100 planner -> DDPM-100 IDM.  BASELINE configs[3]'s per-GPU path."""
    D, A, data = DIMS[cfg]
    low = cfgs.synth_latent_batch(data, B, 1, 250 + B)["obs"]
    g = rng(260 + B + D)
    obs = {k: v for k, v in low.items() if not k.startswith("latent_")}
    for k in data["rgb_obs"]:
        obs[k[len("latent_"):]] = g.integers(0, 256, (B, 1, 64, 64, 3)).astype(np.float64)
    batch = {"obs": obs}
    inp = dict(x_init=g.standard_normal((B, 8, D)), x_noise=g.standard_normal((100, B, 8, D)),
               a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)), **_flat_obs(batch))

    def compute():
        from oracle import np64
        orc = _agent_oracle(cfg, vae=np64.to64(vae_params()))
        a, m = orc.sample_viz(batch, inp["x_init"], inp["x_noise"], inp["a_init"], inp["a_noise"], decode=False)
        enc = orc.vae_encode(orc.postprocess(batch)["obs"])
        return dict(action=a, plan=m["plan"], latent=enc[data["rgb_obs"][0]])
    return inp, compute


def agent_training_batch(cfg, B=3, H=9):
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, H, 77, with_actions=True)
    g = rng(78 + D)
    inp = dict(x_init=g.standard_normal((B, 8, D)), x_noise=g.standard_normal((100, B, 8, D)),
               a_init=g.standard_normal((B * 4, A)), a_noise=g.standard_normal((100, B * 4, A)),
               a2_init=g.standard_normal((B * (H - 1), A)), a2_noise=g.standard_normal((100, B * (H - 1), A)),
               nxt=g.uniform(-1, 1, (B, 4, D)), a3_init=g.standard_normal((B * 4, A)),
               a3_noise=g.standard_normal((100, B * 4, A)), **_flat_obs(batch))

    def compute():
        orc = _agent_oracle(cfg)
        a, m = orc.sample_viz(batch, inp["x_init"], inp["x_noise"], inp["a_init"], inp["a_noise"], decode=False)
        sa = orc.sample_action(batch, inp["a2_init"], inp["a2_noise"])
        obs_only = {"obs": {k: v[:, :4] for k, v in batch["obs"].items()}}
        sp = orc.sample_action_from_plan(obs_only, inp["nxt"], inp["a3_init"], inp["a3_noise"])
        return dict(action=a, plan=m["plan"], plan_mse=np.asarray(m["plan_mse"]), sample_action=sa,
                    sample_action_from_plan=sp)
    return inp, compute


def agent_get_metrics(cfg, B=3, H=9):
    """LDPAgent.get_metrics (agent/ldp_agent.py:328-349 -> loss :141-180), forward only, on a training batch: explicit per-sample
    timesteps and noise for both losses (the reference draws them from a JAX key).  Computed with oracle/np64.py's own
    unet_forward / idm_forward (the float64 NumPy definition)."""
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, H, 177, with_actions=True)
    g = rng(178 + D)
    R = B * (H - 1)
    inp = dict(t_plan=g.integers(0, 100, B).astype(np.float64), noise_plan=g.standard_normal((B, H - 1, D)),
               t_idm=g.integers(0, 100, R).astype(np.float64), noise_idm=g.standard_normal((R, A)), **_flat_obs(batch))

    def compute():
        orc = _agent_oracle(cfg)
        m = orc.get_metrics(batch, inp["t_plan"].astype(np.int64), inp["noise_plan"], inp["t_idm"].astype(np.int64), inp["noise_idm"])
        return {k: np.asarray(v, np.float64) for k, v in m.items()}
    return inp, compute


def agent_update(cfg, B=4, H=9, steps=10, mixed=False):
    """LDPAgent.update / update_mixed (agent/ldp_agent.py:223-323): `steps` training steps on seeded batches with explicit timesteps and noise.
    Kept: the step-0 metrics, g_norm and the two losses of every step, per-leaf digests (tests/util.py leaf_digest) of the step-0 gradients and
    of the parameters after step 1 and after the last step.  float64 autograd + float64 optax.adam restatement (oracle/train.py)."""
    from tests.util import tree_digest
    D, A, data = DIMS[cfg]
    inp = {}
    batches, mixes, noises = [], [], []
    for s in range(steps):
        b = cfgs.synth_latent_batch(data, B, H, 7100 + 17 * s + D, with_actions=True)
        g = rng(7200 + 13 * s + D)
        R = B * (H - 1)
        nz = dict(t_plan=g.integers(0, 100, B).astype(np.float64), noise_plan=g.standard_normal((B, H - 1, D)),
                  t_idm=g.integers(0, 100, R).astype(np.float64), noise_idm=g.standard_normal((R, A)))
        batches.append(b)
        noises.append(nz)
        for k, v in _flat_obs(b).items():
            inp[f"s{s}_{k}"] = v
        for k, v in nz.items():
            inp[f"s{s}_{k}"] = v
        if mixed:
            mb = cfgs.synth_latent_batch(data, B, H, 7300 + 19 * s + D, with_actions=True)
            mixes.append(mb)
            for k, v in _flat_obs(mb).items():
                inp[f"s{s}_mixed_{k}"] = v

    def compute():
        from oracle import train as OT
        orc = _agent_oracle(cfg)
        kw = cfgs.AGENT_KW
        tr = OT.TrainOracle(planner_params(D=D), idm_params(D=D, A=A), lr=kw["lr"], end_lr=kw["end_lr"], idm_lr=kw["idm_lr"],
                            idm_end_lr=kw["idm_end_lr"], warmup_steps=kw["warmup_steps"], decay_steps=kw["decay_steps"])
        out = {}
        series = {k: [] for k in ("plan_loss", "idm_loss", "g_norm", "planner_lr", "idm_lr")}
        for s in range(steps):
            nb = orc.postprocess(batches[s])
            emb, act = orc.get_obs_cond(nb["obs"]), np.asarray(nb["actions"], np.float64)
            extra = {}
            if mixed:
                nbm = orc.postprocess(mixes[s])
                extra = dict(idm_obs_emb=orc.get_obs_cond(nbm["obs"]), idm_actions=np.asarray(nbm["actions"], np.float64))
            nz = noises[s]
            m = tr.update_step(emb, act, t_plan=nz["t_plan"].astype(np.int64), noise_plan=nz["noise_plan"], t_idm=nz["t_idm"].astype(np.int64),
                               noise_idm=nz["noise_idm"], **extra)
            for k in series:
                series[k].append(m[k])
            if s == 0:
                out["grads_planner"] = tree_digest(tr.last["grads_planner"], 11)
                out["grads_idm"] = tree_digest(tr.last["grads_idm"], 12)
                out["planner_after_1"], out["idm_after_1"] = tree_digest(tr.pp, 13), tree_digest(tr.ip, 14)
                out.update(emb_min=emb.min(), emb_max=emb.max(), emb_mean=emb.mean(), emb_std=emb.std(), action_min=act.min(), action_max=act.max())
        out["planner_after_n"], out["idm_after_n"] = tree_digest(tr.pp, 15), tree_digest(tr.ip, 16)
        out["planner_moved"] = np.asarray(max(float(np.abs(tr.pp[k] - np.asarray(v, np.float64)).max()) for k, v in planner_params(D=D).items()))
        for k, v in series.items():
            out[k] = np.asarray(v, np.float64)
        return out
    return inp, compute


HIER_IDM_DOWN = (256, 512)


def hier_idm_params(A=7, D=25, seed=5):
    """The hierarchical agent's IDM: a ConditionalUnet1D over action chunks (input_dim A, cond 2 D, down_dims [256, 512])."""
    from latent_diffusion_planning_amd import weights as W
    from tests.util import _cache
    key = ("hier_idm", A, D, seed)
    if key not in _cache:
        _cache[key] = W.init_planner_params(W.PlannerSpec(A, 2 * D, down_dims=HIER_IDM_DOWN), seed)
    return _cache[key]


def hier_idm_fn(params, trans, a_init, step_noise, n_train, n_steps, sampler):
    """float64 torch restatement of the IDM U-Net's loop (planner_sample with the two-level dims)."""
    from oracle import torch32
    P = torch32.TorchParams(params, dtype=torch.float64)
    return torch32.planner_sample(P, _t64(trans), _t64(a_init), None if step_noise is None else _t64(step_noise),
                                  n_train=n_train, n_steps=n_steps, sampler=sampler, down_dims=HIER_IDM_DOWN).numpy()


def hier_idm_params_heavy(A=7, D=25, seed=5):
    from tests.util import _cache, trained_like, PLANNER_STREAM
    key = ("hier_idm_heavy", A, D, seed)
    if key not in _cache:
        _cache[key] = trained_like(hier_idm_params(A, D, seed), 1000 + seed, heads=("Conv_0",), out_scale=1.0 / 300.0, keep=PLANNER_STREAM)
    return _cache[key]


def hier_idm_loop_heavy(sampler, n_steps, R=3, A=7, D=25):
    """The hierarchical agent's IDM -- a two-level ConditionalUnet1D over chunks of 4 actions (agent/ldp_hier_agent.yaml:18-26) -- on a trained-like
    weight set (tests/util.py trained_like), with `ref32_err` like the other heavy cases."""
    g = rng(1500 + n_steps)
    inp = dict(cond=g.uniform(-1, 1, (R, 2 * D)), x0=g.standard_normal((R, 4, A)), nz=g.standard_normal((n_steps, R, 4, A)))

    def compute():
        from oracle import torch32
        pp = hier_idm_params_heavy(A, D)
        nz = inp["nz"] if sampler == "ddpm" else None
        def run(dtype):
            P = torch32.TorchParams(pp, dtype=dtype)
            t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)      # noqa: E731
            return torch32.planner_sample(P, t(inp["cond"]), t(inp["x0"]), t(nz), n_train=100, n_steps=n_steps, sampler=sampler,
                                          down_dims=HIER_IDM_DOWN).double().numpy()
        plan = run(torch.float64)
        return dict(plan=plan, ref32_err=np.abs(run(torch.float32) - plan).max())
    return inp, compute


def agent_hier_sample_viz(cfg="rm", B=2, pred_horizon=32, ih=4, ah=4, sampler="ddpm", n_steps=100):
    """LDPHierAgent.sample_viz (agent/ldp_hier_agent.py:405-461): planner over pred_horizon // idm_horizon states,
    U-Net IDM over chunks of idm_horizon actions."""
    D, A, data = DIMS[cfg]
    Tp = pred_horizon // ih
    batch = cfgs.synth_latent_batch(data, B, 1, 450 + B)
    g = rng(460 + B + D + n_steps)
    inp = dict(x_init=g.standard_normal((B, Tp, D)), a_init=g.standard_normal((B * ah, ih, A)), **_flat_obs(batch))
    if sampler == "ddpm":
        inp.update(x_noise=g.standard_normal((n_steps, B, Tp, D)), a_noise=g.standard_normal((n_steps, B * ah, ih, A)))

    def compute():
        from oracle import np64
        conf = dict(planner_n_diffusion_steps=100, idm_n_diffusion_steps=100, lowdim_obs=data["lowdim_obs"],
                    rgb_obs=data["rgb_obs"], obs_horizon=1, pred_horizon=pred_horizon, action_horizon=ah, idm_horizon=ih,
                    obs_dim=D, action_dim=A, vae_feature_dim=16)
        orc = np64.HierAgentOracle(conf, planner_params(D=D), hier_idm_params(A, D), None, data["obs_normalization"],
                                   planner_sample_fn=planner_fn, idm_sample_fn=hier_idm_fn)
        a, m = orc.sample_viz(batch, inp["x_init"], inp.get("x_noise"), inp["a_init"], inp.get("a_noise"), decode=False,
                              sampler=sampler, n_steps=n_steps)
        return dict(action=a, plan=m["plan"])
    return inp, compute


def agent_hier_sample_action(cfg="rm", B=2, H=5, ih=4, n_steps=50):
    """LDPHierAgent.sample_action (agent/ldp_hier_agent.py:345-383): the IDM U-Net on the batch's own consecutive frames, DDIM-50."""
    D, A, data = DIMS[cfg]
    batch = cfgs.synth_latent_batch(data, B, H, 470 + B)
    g = rng(480 + B + D)
    inp = dict(a_init=g.standard_normal((B * (H - 1), ih, A)), **_flat_obs(batch))

    def compute():
        from oracle import np64
        conf = dict(planner_n_diffusion_steps=100, idm_n_diffusion_steps=100, lowdim_obs=data["lowdim_obs"], rgb_obs=data["rgb_obs"],
                    obs_horizon=1, pred_horizon=32, action_horizon=4, idm_horizon=ih, obs_dim=D, action_dim=A, vae_feature_dim=16)
        orc = np64.HierAgentOracle(conf, planner_params(D=D), hier_idm_params(A, D), None, data["obs_normalization"],
                                   planner_sample_fn=planner_fn, idm_sample_fn=hier_idm_fn)
        return dict(action=orc.sample_action(batch, inp["a_init"], None, sampler="ddim", n_steps=n_steps))
    return inp, compute


CASES = {}
def agent_hier_update(cfg="rm", B=3, pred_horizon=32, ih=4, steps=4, mixed=False):
    """LDPHierAgent.update / update_mixed (agent/ldp_hier_agent.py:223-322): `steps` training steps of the planner (every ih-th future state) and the
    U-Net IDM (chunks of ih actions per (state, state + ih) pair) on seeded batches with explicit timesteps and noise.  Kept like agent_update."""
    from tests.util import tree_digest
    D, A, data = DIMS[cfg]
    H = 1 + pred_horizon
    Tp, K = pred_horizon // ih, pred_horizon // ih
    inp, batches, mixes, noises = {}, [], [], []
    for s in range(steps):
        b = cfgs.synth_latent_batch(data, B, H, 8100 + 17 * s + D, with_actions=True)
        g = rng(8200 + 13 * s + D)
        nz = dict(t_plan=g.integers(0, 100, B).astype(np.float64), noise_plan=g.standard_normal((B, Tp, D)),
                  t_idm=g.integers(0, 100, B * K).astype(np.float64), noise_idm=g.standard_normal((B * K, ih, A)))
        batches.append(b)
        noises.append(nz)
        for k, v in _flat_obs(b).items():
            inp[f"s{s}_{k}"] = v
        for k, v in nz.items():
            inp[f"s{s}_{k}"] = v
        if mixed:
            mb = cfgs.synth_latent_batch(data, B, H, 8300 + 19 * s + D, with_actions=True)
            mixes.append(mb)
            for k, v in _flat_obs(mb).items():
                inp[f"s{s}_mixed_{k}"] = v

    def compute():
        from oracle import train as OT
        orc = _agent_oracle(cfg)
        kw = cfgs.HIER_KW
        tr = OT.TrainOracle(planner_params(D=D), hier_idm_params(A, D), lr=kw["lr"], end_lr=kw["end_lr"], idm_lr=kw["idm_lr"],
                            idm_end_lr=kw["idm_end_lr"], warmup_steps=kw["warmup_steps"], decay_steps=kw["decay_steps"], idm_horizon=ih,
                            idm_unet_kw=dict(down_dims=HIER_IDM_DOWN))
        out = {}
        series = {k: [] for k in ("plan_loss", "idm_loss", "g_norm", "planner_lr", "idm_lr")}
        for s in range(steps):
            nb = orc.postprocess(batches[s])
            emb, act = orc.get_obs_cond(nb["obs"]), np.asarray(nb["actions"], np.float64)
            extra = {}
            if mixed:
                nbm = orc.postprocess(mixes[s])
                extra = dict(idm_obs_emb=orc.get_obs_cond(nbm["obs"]), idm_actions=np.asarray(nbm["actions"], np.float64))
            nz = noises[s]
            m = tr.update_step(emb, act, t_plan=nz["t_plan"].astype(np.int64), noise_plan=nz["noise_plan"], t_idm=nz["t_idm"].astype(np.int64),
                               noise_idm=nz["noise_idm"], **extra)
            for k in series:
                series[k].append(m[k])
            if s == 0:
                out["grads_planner"] = tree_digest(tr.last["grads_planner"], 11)
                out["grads_idm"] = tree_digest(tr.last["grads_idm"], 12)
                out["planner_after_1"], out["idm_after_1"] = tree_digest(tr.pp, 13), tree_digest(tr.ip, 14)
                out.update(emb_min=emb.min(), emb_max=emb.max(), emb_mean=emb.mean(), emb_std=emb.std(), action_min=act.min(), action_max=act.max())
        out["planner_after_n"], out["idm_after_n"] = tree_digest(tr.pp, 15), tree_digest(tr.ip, 16)
        out["planner_moved"] = np.asarray(max(float(np.abs(tr.pp[k] - np.asarray(v, np.float64)).max()) for k, v in planner_params(D=D).items()))
        for k, v in series.items():
            out[k] = np.asarray(v, np.float64)
        return out
    return inp, compute


CASES["agent_hier_update_rm"] = (agent_hier_update, ())
CASES["agent_hier_update_mixed_rm"] = (agent_hier_update, ("rm", 3, 32, 4, 2, True))
CASES["agent_hier_sample_action_rm_ddim50_b2"] = (agent_hier_sample_action, ())
CASES["agent_hier_sample_viz_rm_b2"] = (agent_hier_sample_viz, ())
CASES["agent_hier_sample_viz_rm_ddim50_b3"] = (agent_hier_sample_viz, ("rm", 3, 32, 4, 4, "ddim", 50))
for _s, _n in (("ddpm", 100), ("ddim", 100), ("ddim", 50)):
    CASES[f"planner_loop_{_s}{_n}"] = (planner_loop, (_s, _n))
CASES["bench_rows_b256_ddim100"] = (bench_rows, ())
# trained-like (heavy-tailed) weight sets: the stress cases of every arithmetic form (tests/test_hip_stress.py)
CASES["planner_loop_heavy_ddpm100"] = (planner_loop_heavy, ("ddpm", 100))
CASES["planner_loop_heavy_ddim50"] = (planner_loop_heavy, ("ddim", 50))
CASES["planner_loop_heavy_t16_ddim50"] = (planner_loop_heavy, ("ddim", 50, 3, 16))
CASES["planner_loop_heavy_wide_ddim50"] = (planner_loop_heavy, ("ddim", 50, 3, 8, 25, True))      # residual stream at 2e7: beyond the fp16 planes
CASES["idm_loop_heavy_rm_ddpm100"] = (idm_loop_heavy, ("rm", "ddpm", 100))
CASES["idm_loop_heavy_rm_ddim50"] = (idm_loop_heavy, ("rm", "ddim", 50))
# round 6: the trained-like sets on the aloha dimensions (D = 30, A = 14) and on the hierarchical agent's U-Net IDM
CASES["planner_loop_heavy_aloha_ddpm100"] = (planner_loop_heavy, ("ddpm", 100, 3, 8, 30))
CASES["idm_loop_heavy_aloha_ddpm100"] = (idm_loop_heavy, ("aloha", "ddpm", 100))
CASES["hier_idm_loop_heavy_ddim50"] = (lambda: hier_idm_loop_heavy("ddim", 50), ())
CASES["planner_loop_t16_ddpm100"] = (planner_loop, ("ddpm", 100, 3, 16))
CASES["agent_sample_viz_rm_t16_b2"] = (agent_sample_viz_t16, ())
CASES["agent_raw_image_aloha_b2"] = (agent_raw_image, ())
CASES["agent_sample_viz_rm_ddim50_b3"] = (agent_sample_viz_ddim, ())
# the same two calls on the tasks BASELINE.json names, with THEIR normalisation tables (data/cfg/rm_square|rm_can/latent_img.yaml)
CASES["agent_sample_viz_rm_square_t16_b2"] = (agent_sample_viz_t16, ("rm_square",))
CASES["agent_sample_viz_rm_can_ddim50_b3"] = (agent_sample_viz_ddim, ("rm_can",))
for _c in ("rm", "aloha"):
    for _s, _n in (("ddpm", 100), ("ddim", 50)):
        CASES[f"idm_loop_{_c}_{_s}{_n}"] = (idm_loop, (_c, _s, _n))
    for _b in (1, 5):
        CASES[f"agent_sample_viz_{_c}_b{_b}"] = (agent_sample_viz, (_c, _b))
    CASES[f"agent_training_batch_{_c}"] = (agent_training_batch, (_c,))
    CASES[f"agent_get_metrics_{_c}"] = (agent_get_metrics, (_c,))
    CASES[f"agent_update_{_c}"] = (agent_update, (_c,))
CASES["agent_update_mixed_rm"] = (agent_update, ("rm", 4, 9, 3, True))


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_case(name):
    """-> (inputs, expected) with float64 expected outputs.  Inputs are rebuilt from the seed and
    cross-checked against the stored float32 copies."""
    fn, args = CASES[name]
    inp, _ = fn(*args)
    path = golden_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `python tests/golden/make_golden.py {name}`")
    with np.load(path) as z:
        exp = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
        for k in z.files:
            if k.startswith("in_"):
                np.testing.assert_allclose(z[k], np.asarray(inp[k[3:]], dtype=np.float32), rtol=0, atol=0,
                                           err_msg=f"golden input {k} of {name} no longer matches its seed")
    return inp, exp
