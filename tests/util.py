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


# ---- "trained-like" weight sets: every other weight set of the suite is a seeded Flax-default init -- GroupNorm scales
# 1 +- 0.1, biases +- 0.02, FiLM outputs small -- so activations are O(1) everywhere.  Checkpoints are not like that.  This transform keeps the
# seeded kernels and gives them the heavy tails a trained network shows:
#   * every norm scale (GroupNorm / LayerNorm)  x 10^N(0, 0.67), clipped to [1e-2, 1e2]   (log-normal over four decades)
#   * every bias                                + N(0, 3), clipped to +-10                 (O(10); FiLM biases included)
#   * every kernel: ONE output channel          x 100   -- except, in the default ("in-range") sets, the convs whose output joins the
#     residual stream without a norm in between AND whose input is that stream (residual projections, stride-2 / transposed convs, conv_in):
#     boosted, they compound -- x 100 per un-normalised conv in a row -- and the planner's stream reaches 2e7, the StableVAE's 1e10.  fp32
#     (and the bf16 planes) carry that; the fp16 planes (|x| < 65504) do not, which is what the `wide=True` sets are for: the range guard
#     must catch them (tests/test_hip_stress.py).  The in-range sets still run at |x| ~ 5e3.
#   * the output heads (`heads`: kernel x out_scale, bias restored) re-calibrated so that the network's OUTPUT is O(1) again -- a trained
#     eps-network predicts unit-variance noise; without this every sample saturates at the +-1 clip and the loops test nothing.
# Pure NumPy PCG64: the GPU box regenerates the same trees from the seed.
def trained_like(params, seed, heads=(), out_scale=1.0, keep=()):
    g = rng(seed)
    out = {}
    for k, v in params.items():
        leaf = k.rsplit("/", 1)[1]
        w = np.array(v, dtype=np.float64)
        if leaf == "scale":
            w = w * 10.0 ** np.clip(g.normal(0.0, 0.67, w.shape), -2.0, 2.0)
        elif leaf == "bias":
            w = w + np.clip(g.normal(0.0, 3.0, w.shape), -10.0, 10.0)
        elif leaf == "kernel" and w.ndim >= 2:
            c = int(g.integers(0, w.shape[-1]))                  # (drawn for every kernel: the sets differ only in which ones use it)
            if not any(m in k for m in keep):
                w[..., c] *= 100.0
        out[k] = w
    for h in heads:
        out[h + "/kernel"] = out[h + "/kernel"] * out_scale
        out[h + "/bias"] = np.array(params[h + "/bias"], dtype=np.float64)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


# kernels that read AND write the un-normalised residual stream
PLANNER_STREAM = ("ConditionalResidualBlock1D_%d/Conv_0/" % i for i in range(16))
PLANNER_STREAM = tuple(PLANNER_STREAM) + ("Downsample1d_", "Upsample1d_")
VAE_STREAM = ("conv_in/", "downsamplers_", "upsamplers_", "conv_shortcut/")


def planner_params_heavy(D=25, seed=0, wide=False):
    key = ("ph", D, seed, wide)
    if key not in _cache:
        _cache[key] = trained_like(planner_params(D=D, seed=seed), 1000 + seed, heads=("Conv_0",), out_scale=1.0 / 300.0,
                                   keep=() if wide else PLANNER_STREAM)
    return _cache[key]


def idm_params_heavy(D=25, A=7, seed=1):
    key = ("ih", D, A, seed)
    if key not in _cache:
        _cache[key] = trained_like(idm_params(D=D, A=A, seed=seed), 1000 + seed, heads=("MLPResNet_0/Dense_1",), out_scale=1.0 / 300.0)
    return _cache[key]


def vae_params_heavy(seed=2, wide=False):
    key = ("vh", seed, wide)
    if key not in _cache:
        base = W.init_vae_params(seed=seed)
        _cache[key] = trained_like(base, 1000 + seed, heads=("quant_conv", "decoder/conv_out"), out_scale=1.0 / 100.0,
                                   keep=() if wide else VAE_STREAM)
    return _cache[key]


def rel_err(got, ref):
    """max |got - ref| / max(1, |ref|): the tolerance rule of the stress tests -- absolute 1e-4 where |ref| <= 1 (plans, normalised
    actions, latents: the north-star's statement), relative 1e-4 where values are large."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())


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




def leaf_digest(arr, seed: int, n_samples: int = 64) -> np.ndarray:
    """What a golden keeps of a (possibly 5-million-element) parameter / gradient leaf: [L2 norm, max |x|, <x, r> / sqrt(n) with r ~ N(0,1) seeded,
    then n_samples elements at seeded positions].  The same function digests the oracle's leaf (tests/golden/make_golden.py) and the HIP result."""
    a = np.asarray(arr, np.float64).reshape(-1)
    g = rng(seed)
    r = g.standard_normal(a.size)
    idx = g.integers(0, a.size, n_samples)
    return np.concatenate([[np.sqrt((a * a).sum()), np.abs(a).max(), float(a @ r) / np.sqrt(a.size)], a[idx]])


def tree_digest(tree, seed: int = 0) -> np.ndarray:
    """(n_leaves, 67) digests of a flat parameter tree in key order."""
    return np.stack([leaf_digest(tree[k], seed + i) for i, k in enumerate(tree)])
