"""Shared helpers for the test-suite (seeded synthetic inputs)."""
import numpy as np

from latent_diffusion_planning_amd import weights as W

RM = dict(D=25, A=7, T=8, ah=4)           # rm_lift / rm_can / rm_square
ALOHA = dict(D=30, A=14, T=8, ah=4)       # aloha sim_transfer_cube


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


_cache = {}


def planner_params(D=25, G=None, seed=0, down_dims=(256, 512, 1024)):
    key = ("p", D, G, seed, down_dims)
    if key not in _cache:
        spec = W.PlannerSpec(D, D if G is None else G, down_dims=tuple(down_dims))
        _cache[key] = W.init_planner_params(spec, seed)
    return _cache[key]


def idm_params(D=25, A=7, seed=1):
    key = ("i", D, A, seed)
    if key not in _cache:
        _cache[key] = W.init_idm_params(W.IDMSpec(D, A), seed)
    return _cache[key]


def maxdiff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    i = np.unravel_index(np.argmax(d), d.shape)
    return float(d.max()), tuple(int(x) for x in i), float(a[i]), float(b[i])


def assert_close(got, ref, atol, what=""):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values in result"
    d, idx, g, r = maxdiff(got, ref)
    assert d <= atol, f"{what}: max|diff| {d:.3e} > {atol:.1e} at {idx}: got {g!r} ref {r!r}"


# ---- agent construction shared by the GPU tests and tools/parity_margin.py (no oracle import here) --------------
def make_agent(name, pp, ip, T=8, vae=None):
    from latent_diffusion_planning_amd.agent import LDPAgent
    from tests import cfgs
    data = cfgs.BY_NAME[name]
    kw = cfgs.agent_kwargs(data)
    kw["pred_horizon"] = T
    ag = LDPAgent.create(0, None, data["shape_meta"], vae_params=vae, **kw)
    # load the synthetic "checkpoint" the way load_snapshot does (train_bc.py:210-240)
    ag = ag.replace(planner_state=ag.planner_state.replace(params=pp, ema_params=pp),
                    idm_state=ag.idm_state.replace(params=ip, ema_params=ip))
    return ag, data


def normalised(action, data):
    """Actions back in the IDM's own [-1, 1] space: the north-star tolerance (1e-4) is stated there; the
    un-normalisation multiplies errors by (max - min) / 2 (up to 1.75 for aloha)."""
    e = data["obs_normalization"]["actions"]
    a = np.asarray(action, dtype=np.float64)
    if "min" in e:
        lo, hi = np.asarray(e["min"], np.float64), np.asarray(e["max"], np.float64)
        return (a - lo) / (hi - lo) * 2 - 1
    return a


