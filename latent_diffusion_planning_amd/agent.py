"""LDPAgent -- the reference's agent surface (agent/ldp_agent.py:28-672) on the HIP engine.

Drop-in for the *sampling* side of the reference class: `create`, `.config[...]`, `.replace`,
`.planner_state/.idm_state` (`.params`, `.replace(params=, ema_params=)`), `sample`,
`sample_viz`, `sample_action`, `sample_action_from_plan`, `vae_encode`, `vae_decode`,
`get_obs_cond`, `get_params`, `get_metrics` (the losses, forward only), `update`, `update_mixed` (the
training step: losses, gradients, global norm, optax.adam -- csrc/train.hip) -- same names, argument
meaning, return structure and error behaviour (e.g. the `assert len(batch.keys()) == 1` of
agent/ldp_agent.py:439).

Differences that are deliberate and documented:
  * `rng`: the reference takes a JAX PRNGKey; here an int seed, a uint32[2] key array or a
    torch.Generator is accepted and used as the seed of the in-kernel Philox stream.  JAX's
    threefry stream is not reproduced; `noise=` gives the explicit-noise parity mode.
  * inputs may be numpy arrays, torch tensors or DeviceArrays.  Outputs (`action`, every value of
    `metrics`) are `arrays.DeviceArray`s: device-resident (`.tensor`) and numpy-like on demand, so
    the reference's call sites run verbatim -- `np.array(batch_action)`, `batch_action[idx]`,
    `np.array(plan_viz)` (utils/rm_env_utils.py:186-196), `((plan_viz + 1) / 2 * 255).astype(np.uint8)
    .transpose(0, 1, 3, 4, 2)` (utils/aloha_env_utils.py:96), `np.mean(np.square(actions - pred))`
    (eval_bc.py:135-151), `float(stats['plan_mse'])`.
  * `metrics['plan_viz']` is always present; `sample()` leaves it lazy (decoded on first access),
    `sample_viz()` enqueues the decode right away like the reference (agent/ldp_agent.py:483).
  * all arithmetic runs in libldp_hip.so; there is no CPU path.
"""
from __future__ import annotations

import copy
import itertools
import warnings
from typing import Any, Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import weights as W
from .arrays import CallRecord, DeviceArray, as_tensor
from .engine import HipEngine

_versions = itertools.count(1)        # tokens of parameter trees (never reused, unlike id())

# vae_feature_dim -> (latent side, latent channels) as agent/ldp_agent.py:69-80 reshapes them; the image side is
# 32 x the latent side (five stride-2 stages): 64-pixel frames for 16 / 32, 96-pixel frames for 36 (3x3x4), 128 for 64.
LATENT_SHAPES = {16: (2, 4), 32: (2, 8), 36: (3, 4), 64: (4, 4)}


# ------------------------------------------------------------------------------------------------
# small stand-ins for flax_utils.TrainStateEMA (utils/flax_utils.py:18-27) as seen by the callers
# ------------------------------------------------------------------------------------------------
def _as_flat(params) -> Dict[str, np.ndarray]:
    if params is None:
        return None
    if any(isinstance(v, Mapping) for v in params.values()):
        return W.flatten(params)
    return {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in params.items()}


class ParamState:
    """Stand-in for flax_utils.TrainStateEMA (utils/flax_utils.py:18-27) as the callers see it: `.params`, `.ema_params`, `.step`, `.opt_state`
    and the `.replace(...)` the reference's load_snapshot uses (train_bc.py:210-240, eval_bc.py:228).

    A state that came out of `LDPAgent.update` lives in the engine's training arenas (fp32 master parameters + the two Adam moments on the
    GPU); its `.params` / `.opt_state` are fetched on first access.  Like a jitted step with donated buffers, only the NEWEST state of a
    module is materialisable: reading a state that a later `update` superseded raises (keep `agent = agent.update(...)[0]` as the reference's
    training loop does, train_bc.py:107).

    version: token of `params` (fresh on every construction / .replace(params=...)): the engine records the token of what it holds, so two
    agents sharing one engine can never sample or train with each other's weights.  Like a flax pytree the dict is treated as immutable --
    publish changes with .replace."""

    def __init__(self, params=None, ema_params=None, step: int = 0, version: Optional[int] = None, opt_state=None, _fetch=None):
        self._params = params
        self.ema_params = ema_params
        self.step = int(step)
        self.version = next(_versions) if version is None else version
        self._opt_state = opt_state
        self._fetch = _fetch                  # (what) -> tree, for a state that lives in the engine

    @property
    def params(self):
        if self._params is None and self._fetch is not None:
            self._params = self._fetch("params")
        return self._params

    @property
    def opt_state(self):
        """{'mu': tree, 'nu': tree, 'count': step} (optax ScaleByAdamState) or None for a state that was never trained."""
        if self._opt_state is None and self._fetch is not None:
            self._opt_state = dict(mu=self._fetch("mu"), nu=self._fetch("nu"), count=self.step)
        return self._opt_state

    def replace(self, **kw):
        new = ParamState(self._params, self.ema_params, self.step, self.version, self._opt_state, self._fetch)
        if "params" in kw:
            new.version = kw.pop("version", next(_versions))
            new._params = _as_flat(kw.pop("params"))
            new._fetch = None
            new._opt_state = None if "opt_state" not in kw else new._opt_state      # new parameters start a new optimiser state unless one is given
        if "ema_params" in kw:
            e = kw.pop("ema_params")
            new.ema_params = _as_flat(e) if e is not None else None
        if "opt_state" in kw:
            o = kw.pop("opt_state")
            new._opt_state = None if o is None else dict(mu=_as_flat(o["mu"]), nu=_as_flat(o["nu"]), count=int(o.get("count", new.step)))
            new.version = next(_versions) if new._fetch is None else new.version
        if "step" in kw:
            new.step = int(kw.pop("step"))
        if "version" in kw:
            new.version = kw.pop("version")
        if kw:
            raise AttributeError(f"ParamState has no field(s) {sorted(kw)}")
        return new


def _philox_normal(seed, elem0, step, stream_id, n, device):
    from .engine import philox_normal
    return philox_normal(seed, elem0, step, stream_id, n, device)


class _HostScalar:
    """A metric scalar that lives on the device until somebody looks at it: float()-able and np.asarray-able like the jnp 0-d arrays of the
    reference's metrics dict (eval_bc.py:152: `float(np.mean([m[k] for m in all_metrics]))`).  `fn` reads DeviceArrays (the first read is the
    call's completion point: fault poll, recompute) and does whatever scalar arithmetic is left on the host -- nothing of get_metrics is
    computed by torch."""

    def __init__(self, fn):
        self._fn = fn
        self.shape, self.ndim, self.dtype = (), 0, np.dtype(np.float32)

    def __float__(self):
        return float(self._fn())

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._fn(), dtype=np.float32)
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return "_HostScalar(...)"


def _Elem(vec, i):
    return _HostScalar(lambda: vec.numpy()[i])


def _seed_of(rng) -> int:
    if rng is None:
        return 0
    if isinstance(rng, torch.Generator):
        # a stateful generator: every call consumes one draw, so successive policy calls see fresh noise
        return int(torch.randint(0, 2**62, (1,), generator=rng, device=rng.device).item())
    if isinstance(rng, (int, np.integer)):
        return int(rng)
    a = np.asarray(rng)
    if a.shape == (2,):                                   # a JAX-style uint32[2] key
        return (int(a[0]) << 32) | int(a[1])
    if a.shape == ():
        return int(a)
    raise TypeError(f"rng must be an int seed, a uint32[2] key or a torch.Generator, got {type(rng)}")


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _norm_entry(entry) -> dict:
    """obs_normalization leaf: {'min','max'} (scalars or vectors) or {'clip_min','clip_max'}."""
    e = dict(entry)
    if "mean" in e:
        raise NotImplementedError                        # utils/data_utils.py:31-32
    if "min" in e or "clip_min" in e:
        return {k: (float(v) if np.ndim(v) == 0 else np.asarray(v, dtype=np.float32)) for k, v in e.items()}
    raise NotImplementedError                            # utils/data_utils.py:66-67


def load_pretrained_vae(vp: str):
    """`vae_pretrain_path` -> the VAE parameter tree.  The containers this package added are recognised by their
    suffix first (.safetensors, .npz -- also when they sit under a '....ckpt/' directory); anything else with 'ckpt' in
    its path follows the reference's rule (agent/ldp_agent.py:543-551: an orbax checkpoint written by train_vae.py)."""
    if vp.endswith(".safetensors"):
        loaded = W.load_safetensors(vp)
    elif vp.endswith(".npz"):
        loaded = W.load_npz(vp)
    elif "ckpt" in vp:
        from . import checkpoint
        loaded = checkpoint.param_trees(checkpoint.restore(vp))
        if "vae_params" not in loaded:
            raise KeyError(f"{vp}: no vae_params tree in this checkpoint (has {sorted(loaded)})")
    else:
        loaded = W.load_npz(vp)
    return loaded.get("vae_params") or loaded.get("vae") or next(iter(loaded.values()))


class LDPAgent:
    # ---------------------------------------------------------------------------------------------
    def __init__(self, planner_state, idm_state, vae_params, obs_normalization, use_planner, use_idm,
                 alpha_planner, alpha_idm, config, engine: Optional[HipEngine], planner_spec, idm_spec,
                 vae_spec, device, lr_schedules=None):
        self.planner_state = planner_state
        self.idm_state = idm_state
        self.vae_params = vae_params
        self.obs_normalization = obs_normalization
        self.use_planner, self.use_idm = use_planner, use_idm
        self.alpha_planner, self.alpha_idm = alpha_planner, alpha_idm
        self.config = config
        self._engine = engine
        self._planner_spec, self._idm_spec, self._vae_spec = planner_spec, idm_spec, vae_spec
        self._device = device
        self._vae_version = next(_versions)
        # {"planner": f(count), "idm": f(count)}: optax.warmup_cosine_decay_schedule per network (agent/ldp_agent.py:583-589, 621-627); the
        # reference object also keeps `self.lr_schedule` -- whichever of the two `create` built LAST (:669) -- and reports THAT one's value for
        # both `planner_lr` and `idm_lr` (:257,266): kept as `lr_schedule` for the metrics only
        self._lr_schedules = lr_schedules or {}
        self.lr_schedule = self._lr_schedules.get("idm" if use_idm else "planner")

    # ---------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, rng, batch, shape_meta,
               # Hydra config (agent/ldp_agent.yaml)
               name, planner, idm_net, preprocess_time, cond_encoder,
               vae_pretrain_path, vae_feature_dim,
               use_planner, use_idm,
               lowdim_obs, rgb_obs, obs_normalization, data_name,
               obs_horizon, pred_horizon, action_horizon,
               planner_n_diffusion_steps, idm_n_diffusion_steps,
               alpha_planner=1, alpha_idm=1,
               lr=None, end_lr=None, idm_lr=None, idm_end_lr=None,
               warmup_steps=None, decay_steps=None,
               update_planner_every=1, update_idm_every=1, update_idm_after=-1,
               update_planner_until=-1, update_planner_after=-1,
               grad_clip=None, device=None, vae_params=None, exclusive_gpu=True):
        """agent/ldp_agent.py:516-672.  `batch` is accepted for signature parity (the reference
        traces shapes from it); dims come from `shape_meta` exactly as there (:534-540).

        exclusive_gpu=False: another engine process computes on the same GPU.  Results are bit-stable either way
        (DESIGN.md 4.5), but the work-groups of the in-launch exchanges (column split, K split) then spin while the other
        process holds the CUs; this runs the handle without them (`safe_mode`, about half the speed at <= 256 plans)."""
        lowdim_obs, rgb_obs = list(lowdim_obs), list(rgb_obs)
        if len(rgb_obs) > 1:
            # get_obs_cond concatenates cameras on the time axis (:93-94): only defined for one
            raise NotImplementedError("more than one rgb_obs key: the reference's get_obs_cond "
                                      "concatenates cameras on axis 1 and is only well-defined for one")
        lowdim_dim = sum(int(np.prod(shape_meta["all_shapes"][k])) for k in lowdim_obs)
        # vae_feature_dim 16 (2x2x4 latent of 64x64 frames) is what every shipped config uses; agent/ldp_agent.py:69-80
        # also lists 32 / 36 / 64 (other latent shapes / image sizes: 64 / 96 / 128 pixel frames): all four are built
        if int(vae_feature_dim) not in LATENT_SHAPES:
            raise NotImplementedError(f"vae_feature_dim={vae_feature_dim}: the latent shapes of agent/ldp_agent.py:69-80 are "
                                      f"{sorted(LATENT_SHAPES)} (2x2x4, 2x2x8, 3x3x4, 4x4x4)")
        side, latent_ch = LATENT_SHAPES[int(vae_feature_dim)]
        image_size = 32 * side
        for k in rgb_obs:
            raw = k[len("latent_"):] if k.startswith("latent_") else k
            shp = shape_meta.get("all_shapes", {}).get(raw)
            if shp is not None and tuple(int(v) for v in shp) != (image_size, image_size, 3):
                raise NotImplementedError(f"image key {raw!r} has shape {tuple(shp)}: vae_feature_dim={vae_feature_dim} "
                                          f"means {image_size}x{image_size}x3 frames for the StableVAE")
        obs_dim = lowdim_dim + int(vae_feature_dim) * len(rgb_obs)
        if obs_dim > 128:
            raise NotImplementedError(f"obs_dim={obs_dim}: the planner's first conv is packed for observation "
                                      "embeddings of at most 128 features")
        action_dim = int(shape_meta["ac_dim"])
        seed = _seed_of(rng)

        down_dims = tuple(int(d) for d in _get(planner, "down_dims", (256, 512, 1024)))
        pspec = W.PlannerSpec(input_dim=obs_dim, global_cond_dim=obs_dim * int(obs_horizon),
                              diffusion_step_embed_dim=int(_get(planner, "diffusion_step_embed_dim", 256)),
                              down_dims=down_dims, kernel_size=int(_get(planner, "kernel_size", 5)),
                              n_groups=int(_get(planner, "n_groups", 8)),
                              downsample=bool(_get(planner, "downsample", True)))
        if not pspec.downsample:
            raise NotImplementedError("downsample=False planners are not built")
        if any(d < 256 or d % 128 for d in down_dims) or pspec.kernel_size != 5 or pspec.n_groups != 8:
            raise NotImplementedError(f"planner down_dims={down_dims} kernel_size={pspec.kernel_size} n_groups="
                                      f"{pspec.n_groups}: the MFMA conv tiles are built for kernel_size 5, 8 groups and "
                                      "levels that are multiples of 128 channels and at least 256 wide")
        if not bool(_get(idm_net, "use_layer_norm", True)) or _get(idm_net, "dropout_rate", None):
            raise NotImplementedError("IDM variants other than LayerNorm / no dropout are not built")
        if str(_get(cond_encoder, "activations", "mish")) != "mish" or bool(_get(cond_encoder, "activate_final", False)):
            raise NotImplementedError("cond_encoder must be MLP(mish, activate_final=False)")
        if bool(_get(preprocess_time, "learnable", False)):
            raise NotImplementedError("learnable FourierFeatures are not built")
        ispec = W.IDMSpec(obs_dim=obs_dim, action_dim=action_dim,
                          time_dim=int(_get(preprocess_time, "output_size", 256)),
                          cond_hidden=tuple(int(h) for h in _get(cond_encoder, "hidden_dims", (256, 256))),
                          hidden_dim=int(_get(idm_net, "hidden_dim", 256)),
                          n_blocks=int(_get(idm_net, "n_blocks", 3)))
        vspec = W.VAESpec(latent_channels=latent_ch)

        planner_state = idm_state = None
        if use_planner:
            planner_state = ParamState(W.init_planner_params(pspec, seed=seed * 3 + 1, perturb=False))
        if use_idm:
            idm_state = ParamState(W.init_idm_params(ispec, seed=seed * 3 + 2, perturb=False))
        if vae_params is None and vae_pretrain_path is not None:
            vae_params = load_pretrained_vae(str(vae_pretrain_path))
        vae_params = _as_flat(vae_params) if vae_params is not None else None

        config = dict(planner_n_diffusion_steps=int(planner_n_diffusion_steps),
                      idm_n_diffusion_steps=int(idm_n_diffusion_steps),
                      lowdim_obs=lowdim_obs, rgb_obs=rgb_obs, obs_horizon=int(obs_horizon),
                      name=name, action_dim=action_dim,
                      pred_horizon=int(pred_horizon), action_horizon=int(action_horizon),
                      obs_dim=obs_dim,
                      update_planner_every=update_planner_every, update_idm_every=update_idm_every,
                      update_planner_until=update_planner_until,
                      update_planner_after=update_planner_after,
                      update_idm_after=update_idm_after,
                      vae_feature_dim=int(vae_feature_dim), data_name=data_name)
        norm = {"obs": {k: _norm_entry(v) for k, v in dict(obs_normalization["obs"]).items()}}
        if "actions" in obs_normalization:
            norm["actions"] = _norm_entry(obs_normalization["actions"])

        if not torch.cuda.is_available():
            from ._lib import LDPHipUnavailable
            raise LDPHipUnavailable("no HIP device visible: LDPAgent has no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        engine = HipEngine(obs_dim=obs_dim, action_dim=action_dim, global_cond_dim=pspec.global_cond_dim,
                           pred_horizon=int(pred_horizon), action_horizon=int(action_horizon),
                           down_dims=down_dims, kernel_size=pspec.kernel_size, n_groups=pspec.n_groups,
                           step_embed_dim=pspec.diffusion_step_embed_dim,
                           planner_train_steps=int(planner_n_diffusion_steps),
                           idm_train_steps=int(idm_n_diffusion_steps), idm_hidden=ispec.hidden_dim,
                           idm_blocks=ispec.n_blocks, idm_time_dim=ispec.time_dim, image_size=image_size,
                           vae_latent_channels=latent_ch, device=dev)
        if not exclusive_gpu:
            engine.set_option("safe_mode", 1)
        from .schedule import warmup_cosine_decay_schedule
        sched = {}
        if lr is not None and warmup_steps is not None and decay_steps is not None:
            if use_planner:
                sched["planner"] = warmup_cosine_decay_schedule(float(end_lr), float(lr), int(warmup_steps), int(decay_steps), float(end_lr))
            if use_idm:
                sched["idm"] = warmup_cosine_decay_schedule(float(idm_end_lr if idm_end_lr is not None else end_lr),
                                                            float(idm_lr if idm_lr is not None else lr), int(warmup_steps), int(decay_steps),
                                                            float(idm_end_lr if idm_end_lr is not None else end_lr))
        return cls(planner_state, idm_state, vae_params, norm, use_planner, use_idm, alpha_planner,
                   alpha_idm, config, engine, pspec, ispec, vspec, dev, lr_schedules=sched)

    # ---------------------------------------------------------------------------------------------
    def replace(self, **fields):
        """flax.struct `.replace`: a shallow copy sharing the engine (weights re-upload lazily)."""
        new = copy.copy(self)
        for k, v in fields.items():
            if not hasattr(new, k):
                raise AttributeError(f"LDPAgent has no field {k!r}")
            if k == "vae_params":
                v = _as_flat(v)
                new._vae_version = next(_versions)
            setattr(new, k, v)
        return new

    def get_params(self):
        """agent/ldp_agent.py:508-514."""
        params = {}
        if self.use_planner:
            params["planner_params"] = self.planner_state.params
        if self.use_idm:
            params["idm_params"] = self.idm_state.params
        return params

    def _sync_weights(self, need_vae=False):
        """Upload whatever the (possibly shared) engine does not hold for THIS agent: the engine keeps the
        version token of each module's tree, the agent compares it with its own."""
        up, ver = {}, {}
        held = self._engine.loaded
        for name, use, st, shapes in (("planner", self.use_planner, self.planner_state, self._planner_shapes),
                                      ("idm", self.use_idm, self.idm_state, self._idm_shapes)):
            if not use or held[name] == st.version:
                continue
            if self._engine.train_token.get(name) == st.version:
                # the state an update() left in the training arenas: master parameters -> packed sampling layouts, on the device side
                self._engine.train_publish([name], versions={name: st.version})
                continue
            W.check_params(st.params, shapes())
            up[name], ver[name] = st.params, st.version
        if need_vae and held["vae"] != self._vae_version:
            if self.vae_params is None:
                raise ValueError("raw image observations need VAE weights (vae_pretrain_path / vae_params)")
            up["vae"], ver["vae"] = self.vae_params, self._vae_version
        if up:
            self._engine.load_params(**up, versions=ver)

    def _planner_shapes(self):
        return W.planner_shapes(self._planner_spec)

    def _idm_shapes(self):
        return W.idm_shapes(self._idm_spec)

    # ---- pre/post-processing (utils/data_utils.py:18-80) ----------------------------------------
    def _t(self, v) -> torch.Tensor:
        v = as_tensor(v)
        return v.to(device=self._device, dtype=torch.float32).contiguous()

    def _apply_norm(self, v: torch.Tensor, entry: dict, normalize: bool) -> torch.Tensor:
        if "min" in entry:
            lo, hi = entry["min"], entry["max"]
            if np.ndim(lo) != 0:
                lo, hi = np.asarray(lo), np.asarray(hi)
                diff = v.dim() - lo.ndim
                assert diff in (0, 1, 2, 3, 4, 5), "shape length mismatch in normalize_obs"
                assert tuple(v.shape[diff:]) == tuple(lo.shape), \
                    f"shape mismatch in normalize obs. {tuple(v.shape)}, {lo.shape}"
                lo, hi = lo.reshape(-1), hi.reshape(-1)
                return self._engine.normalize_bounds(v.reshape(-1, lo.size), lo, hi, normalize).reshape(v.shape)
            return self._engine.normalize_bounds(v, [lo], [hi], normalize)
        # clip_min / clip_max: plain clip in both directions (utils/data_utils.py:61-65)
        return self._engine.normalize_bounds(v, [entry["clip_min"]], [entry["clip_max"]], 2)

    def _normalize_dict(self, d, table, normalize=True):
        assert set(d.keys()).issubset(table), \
            f"obs_normalization keys {table.keys()} do not match batch keys {d.keys()}"
        return {k: self._apply_norm(self._t(v), table[k], normalize) for k, v in d.items()}

    def _postprocess(self, batch):
        """postprocess_batch / postprocess_batch_obs selection of agent/ldp_agent.py:436-440."""
        if "actions" in batch.keys():
            out = {"obs": self._normalize_dict(batch["obs"], self.obs_normalization["obs"])}
            out["actions"] = self._apply_norm(self._t(batch["actions"]), self.obs_normalization["actions"], True)
            return out
        assert len(batch.keys()) == 1
        return {"obs": self._normalize_dict(batch["obs"], self.obs_normalization["obs"])}

    # ---- agent/ldp_agent.py:46-64 -----------------------------------------------------------------
    def vae_encode(self, batch):
        """agent/ldp_agent.py:46-64.  Called on its own (the reference's public method) the encode is a recorded, guarded call like vae_decode: the
        StableVAE's stride-2 / split convs sit behind the range guard, and a fault is recomputed and reported instead of leaving inf latents with
        the caller.  The agent's own policy calls use _vae_encode_t inside their recorded run."""
        keys = [k for k in batch.keys() if f"latent_{k}" in self.config["rgb_obs"]]
        if not keys:
            return {k: self._t(v) for k, v in batch.items()}

        def run():
            out = self._vae_encode_t(batch)
            return [out[k] for k in out]
        rec = self._record(run)
        res = self._guarded(run)
        rec.seqs = self._seqs()
        order = [(f"latent_{k}" if k in keys else k) for k in batch.keys()]
        return {k: DeviceArray(t, record=rec) for k, t in zip(order, res)}

    def _vae_encode_t(self, batch):
        new_batch = {}
        for key in batch.keys():
            if f"latent_{key}" not in self.config["rgb_obs"]:
                new_batch[key] = self._t(batch[key])
                continue
            self._sync_weights(need_vae=True)
            init_obs = self._t(batch[key])
            B, H = init_obs.shape[:2]
            z = self._engine.vae_encode(init_obs.reshape(-1, *init_obs.shape[-3:]))   # NHWC in, NHWC mean out
            feats = z.reshape(B, H, -1)
            feats = self._apply_norm(feats, self.obs_normalization["obs"][f"latent_{key}"], True)
            new_batch[f"latent_{key}"] = feats
        return new_batch

    # ---- agent/ldp_agent.py:66-85 -----------------------------------------------------------------
    def _vae_decode_t(self, feats: torch.Tensor) -> torch.Tensor:
        B, H = feats.shape[:2]
        fd = self.config["vae_feature_dim"]
        side, latent_ch = LATENT_SHAPES[fd]
        if self.vae_params is None or "decoder/conv_in/kernel" not in self.vae_params:
            raise ValueError("plan_viz needs the StableVAE decoder weights (vae_pretrain_path / vae_params with "
                             "decoder/... and post_quant_conv/...)")
        self._sync_weights(need_vae=True)
        z = feats[:, :, :fd].reshape(B * H, side, side, latent_ch)        # agent/ldp_agent.py:69-80
        key = self.config["rgb_obs"][0]
        z = self._apply_norm(z.contiguous(), self.obs_normalization["obs"][key], False)
        img = self._engine.vae_decode(z)                      # (B*H, 3, S, S), like decode(...).sample
        return img.reshape(B, H, *img.shape[1:])

    def vae_decode(self, feats):
        t = self._t(feats)

        def run():
            return [self._vae_decode_t(t)]
        rec = self._record(run)                                # the decoder's split convs sit behind the range guard
        res = self._guarded(run)
        rec.seqs = self._seqs()
        return DeviceArray(res[0], record=rec)

    # ---- agent/ldp_agent.py:88-97 -----------------------------------------------------------------
    def get_obs_cond(self, batch):
        lowdim = torch.cat([self._t(batch[k]) for k in self.config["lowdim_obs"]], dim=-1)
        B, H = lowdim.shape[:2]
        img = torch.cat([self._t(batch[k]) for k in self.config["rgb_obs"]], dim=1)
        return torch.cat([img.reshape(B, H, -1), lowdim.reshape(B, H, -1)], dim=-1)

    # ---- completion hook of a policy call (fault protocol, include/ldp_hip.h) --------------------------
    def _engines(self):
        """Every engine handle a policy call of this agent enqueues on (LDPHierAgent has two)."""
        return [self._engine]

    def _seqs(self):
        return tuple(e.call_seq for e in self._engines())

    def _record(self, recompute):
        """recompute() -> list of replacement tensors (None for lazy arrays), in the order the call's
        DeviceArrays were created."""
        engines = self._engines()

        def on_complete(rec: CallRecord):
            # calls are asynchronous: a recorded fault may stem from any call enqueued so far, so the poll marks
            # them all suspect (engine.fault_upto) and each is recomputed when ITS results are first read
            for e in engines:
                e.poll_fault_kinds()
            if not any(s <= e.fault_upto for s, e in zip(rec.seqs, engines)):
                return
            kinds = 0
            for s_, e in zip(rec.seqs, engines):               # the kind(s) of the poll that made THIS call suspect, not every kind the handle ever saw
                if s_ <= e.fault_upto:
                    kinds |= e.last_fault_kinds
            if kinds & HipEngine.FAULT_RANGE:
                warnings.warn("libldp_hip: an operand left the range of the two-fp16-plane convolutions (|x| >= 65504); the call "
                              "is recomputed on three bf16 planes (fp32 range; the IDM on its exact-fp32 kernel), which this engine keeps from now on",
                              RuntimeWarning, stacklevel=3)
            if kinds & HipEngine.FAULT_EXCHANGE:
                warnings.warn("libldp_hip: a split work-group timed out on its peer (GPU shared with another "
                              "kernel?); the call is recomputed in safe mode, which this engine keeps from now on",
                              RuntimeWarning, stacklevel=3)
            # a recompute may meet the OTHER kind of fault for the first time (safe mode first, then the range guard): bounded retries
            for attempt in range(3):
                fresh = self._guarded(recompute)
                torch.cuda.current_stream(self._device).synchronize()
                again = 0
                for e in engines:
                    again |= e.poll_fault_kinds()
                if not again:
                    break
            else:
                raise RuntimeError("libldp_hip faulted again after switching to safe mode / bf16 planes")
            for arr, t in zip(rec.arrays, fresh):
                if arr is not None:
                    arr._swap(t)
        return CallRecord(on_complete)

    def _guarded(self, run):
        """Run the engine calls of one policy call; if an engine refuses because an earlier (unread) call
        faulted, acknowledge on every engine -- that marks the earlier calls suspect, they are recomputed when read -- and retry."""
        from ._lib import LDPHipFault
        for attempt in range(4):                     # in-flight pre-safe-mode launches may still fault after the first acknowledge
            try:
                return run()
            except LDPHipFault:
                if attempt == 3:
                    raise
                torch.cuda.current_stream(self._device).synchronize()
                for e in self._engines():
                    e.poll_fault_kinds()

    def _action_bounds(self):
        """(lo, hi, mode) of utils/data_utils.py:61-68 for the un-normalisation of actions."""
        e = self.obs_normalization["actions"]
        if "min" in e:
            return np.atleast_1d(np.asarray(e["min"], np.float32)), np.atleast_1d(np.asarray(e["max"], np.float32)), 0
        return np.asarray([e["clip_min"]], np.float32), np.asarray([e["clip_max"]], np.float32), 2

    # ---- IDM loop shared by the action samplers ---------------------------------------------------
    def _idm_actions(self, first, second, seed, B, noise=None, row_offset=0, sampler="ddpm", n_steps=None):
        trans = torch.cat([first, second], dim=-1)
        trans = trans.reshape(-1, trans.shape[-1]).contiguous()          # 'B H D -> (B H) D'
        a_init = a_noise = None
        if noise is not None:
            a_init, a_noise = noise.get("a_init"), noise.get("a_noise")
        rows_per = trans.shape[0] // B
        a = self._engine.idm_sample(trans, a_init=a_init, step_noise=a_noise, seed=seed,
                                    row_offset=row_offset * rows_per, sampler=sampler, n_steps=n_steps)
        a = a.reshape(B, -1, a.shape[-1])
        return self._apply_norm(a, self.obs_normalization["actions"], False)

    # ---- agent/ldp_agent.py:350-389 ---------------------------------------------------------------
    def sample_action_from_plan(self, batch, next_plan, eval_rng, noise=None):
        seed = _seed_of(eval_rng)

        def run():
            self._sync_weights()
            nb = self._postprocess(batch)
            obs = self._vae_encode_t(nb["obs"])
            start = self.get_obs_cond(obs)
            return [self._idm_actions(start, self._t(next_plan), seed, start.shape[0], noise)]
        rec = self._record(run)
        res = self._guarded(run)
        rec.seqs = self._seqs()
        return DeviceArray(res[0], record=rec)

    # ---- agent/ldp_agent.py:391-430 ---------------------------------------------------------------
    def sample_action(self, batch, eval_rng, noise=None):
        seed = _seed_of(eval_rng)

        def run():
            self._sync_weights()
            nb = self._postprocess(batch)
            obs = self._vae_encode_t(nb["obs"])
            plan = self.get_obs_cond(obs)
            return [self._idm_actions(plan[:, :-1], plan[:, 1:], seed, plan.shape[0], noise)]
        rec = self._record(run)
        res = self._guarded(run)
        rec.seqs = self._seqs()
        return DeviceArray(res[0], record=rec)

    # ---- agent/ldp_agent.py:432-506 ---------------------------------------------------------------
    def sample(self, batch, eval_rng, **kw):
        """agent/ldp_agent.py:432-433.  Same return value as sample_viz; the only difference is that the
        image decode behind metrics['plan_viz'] is deferred until (unless) the caller reads it."""
        kw.setdefault("decode", False)
        return self.sample_viz(batch, eval_rng, **kw)

    def get_action(self, batch, eval_rng, **kw):
        """Alias named by BASELINE.json's north_star (the reference has no such method)."""
        return self.sample(batch, eval_rng, **kw)[0]

    def _sample_core(self, batch, seed, noise, row_offset, sampler, n_steps, idm_steps):
        """normalize -> vae_encode -> get_obs_cond -> ONE ldp_agent_sample (planner loop, plan assembly,
        IDM loop, action un-normalisation as one captured graph).  -> (action, plan, x, obs_emb) tensors."""
        self._sync_weights()
        cfg = self.config
        nb = self._postprocess(batch)
        obs = self._vae_encode_t(nb["obs"])
        obs_emb = self.get_obs_cond(obs).contiguous()
        lo, hi, mode = self._action_bounds()
        nz = noise or {}
        x, plan, action = self._engine.agent_sample(
            obs_emb, cfg["obs_horizon"], x_init=nz.get("x_init"), x_noise=nz.get("x_noise"),
            a_init=nz.get("a_init"), a_noise=nz.get("a_noise"), seed=seed, row_offset=row_offset, sampler=sampler,
            planner_steps=n_steps, idm_steps=idm_steps, action_bounds=(lo, hi), action_mode=mode)
        return action, plan, x, obs_emb

    def sample_viz(self, batch, eval_rng, noise=None, decode=True, row_offset=0,
                   sampler="ddpm", n_steps=None, idm_steps=None):
        """-> (action (B, ah, A), {'plan' (B, ah+1, D), 'plan_viz' (B, ah+1, 3, S, S)[, 'plan_mse']}), all
        DeviceArrays.
        noise: optional dict(x_init (B,T,D), x_noise (S,B,T,D), a_init (B*ah,A), a_noise (S,B*ah,A))
        for explicit-noise parity runs.  decode: True = enqueue the VAE decode of the plan now (the
        reference always does, agent/ldp_agent.py:483); False = leave metrics['plan_viz'] lazy (the
        decode is 5 x 24.9 GFLOP per plan and callers such as eval_bc.py:144 discard it).
        row_offset: global index of this batch's first plan (keeps the Philox stream independent of how
        candidates are sharded over GPUs).  n_steps / idm_steps: denoising steps (DDIM only; DDPM
        visits every training timestep)."""
        if not (self.use_planner and self.use_idm):
            raise NotImplementedError("sample() needs both the planner and the IDM (use_planner / use_idm)")
        seed = _seed_of(eval_rng)
        oh = self.config["obs_horizon"]
        if idm_steps is None and n_steps is not None and sampler != "ddpm":
            idm_steps = n_steps                                # one schedule for both loops unless told otherwise

        def run():
            action, plan, x, obs_emb = self._sample_core(batch, seed, noise, row_offset, sampler, n_steps, idm_steps)
            out = [action, plan]
            if obs_emb.shape[1] > oh:                          # from a training batch, not inference (:447-448)
                out.append(self._engine.mean_sq_diff(x, obs_emb[:, oh:]))
            return out
        rec = self._record(lambda: run() + [None])             # plan_viz re-decodes itself from the new plan
        res = self._guarded(run)
        rec.seqs = self._seqs()
        action = DeviceArray(res[0], record=rec)
        plan = DeviceArray(res[1], record=rec)
        metrics = {"plan": plan}
        if len(res) > 2:
            metrics["plan_mse"] = DeviceArray(res[2], record=rec)
        S = self._engine.image_size
        viz = DeviceArray(thunk=lambda: self._vae_decode_t(plan.tensor), shape=tuple(plan.shape[:2]) + (3, S, S),
                          record=rec)
        if decode:
            viz.tensor                                         # enqueue the decode now (no host sync)
        metrics["plan_viz"] = viz
        return action, metrics

    # ---- agent/ldp_agent.py:223-323: the training step ----------------------------------------------------------------
    def _gates(self, step):
        """The python-side schedule gates of `update` / `update_mixed` (agent/ldp_agent.py:224-230, 275-281)."""
        cfg = self.config
        use_planner = bool(self.use_planner) and step % cfg["update_planner_every"] == 0
        use_idm = bool(self.use_idm) and step % cfg["update_idm_every"] == 0
        use_idm = use_idm and step >= cfg["update_idm_after"]
        update_planner = cfg["update_planner_until"] < 0 or step < cfg["update_planner_until"]
        update_planner = update_planner and step >= cfg["update_planner_after"]
        return use_planner and update_planner, use_idm

    def update(self, batch, rng, step, noise=None):
        """agent/ldp_agent.py:223-272 -> (new agent, metrics).  One jax.grad(loss) + optax.adam step per network, on the GPU (csrc/train.hip).
        rng: seed of the timesteps (host PCG64) and of the noise (device Philox), as in get_metrics; noise: optional explicit
        dict(t_plan, noise_plan, t_idm, noise_idm) for parity runs."""
        use_planner, use_idm = self._gates(int(step))
        return self._update_step(batch, None, rng, use_planner, use_idm, noise)

    def update_mixed(self, batch, mixed_batch, rng, step, noise=None):
        """agent/ldp_agent.py:274-323: the planner learns from `batch`, the IDM from `mixed_batch`."""
        use_planner, use_idm = self._gates(int(step))
        return self._update_step(batch, mixed_batch, rng, use_planner, use_idm, noise)

    def _train_sync(self, name, state, shapes, eng=None):
        """The engine's training arenas must hold THIS agent's state of the module (another agent sharing the engine, a load_snapshot or a
        fresh create may have left something else there).  eng: the handle that trains the module (default: the agent's first)."""
        eng = self._engine if eng is None else eng
        if eng.train_token.get(name) == state.version:
            return
        W.check_params(state.params, shapes)
        o = state.opt_state
        eng.train_load(name, state.params, mu=None if o is None else o["mu"], nu=None if o is None else o["nu"], step=state.step,
                       token=state.version)

    def _trained_state(self, name, old, shapes, eng=None):
        """The state after this step: parameters / moments stay on the GPU and are fetched on demand (while it is the newest state)."""
        eng = self._engine if eng is None else eng
        token = next(_versions)
        eng.train_token[name] = token
        which = {"params": eng.TRAIN_PARAMS, "mu": eng.TRAIN_MU, "nu": eng.TRAIN_NU}

        def fetch(what):
            if eng.train_token.get(name) != token:
                raise RuntimeError(f"this {name} state was superseded by a later update(): its buffers were donated to the next step "
                                   "(keep the agent that update() returned, as train_bc.py:107 does)")
            return eng.train_read(name, which[what], shapes)
        return ParamState(None, old.ema_params, old.step + 1, token, None, fetch)

    def _update_step(self, batch, mixed_batch, rng, use_planner, use_idm, noise, shard=None):
        """shard (dist.update_sharded): dict(group, rows=(lo, n), mixed_rows=(lo, n)) -- `batch` / `mixed_batch` are rows [lo, lo + B) of a
        global batch of n rows split over the ranks of `group`.  Timesteps and noise are those of the global rows (so the step does not
        depend on the world size), each rank's loss is weighted B / n, the gradient arenas are summed over the ranks with one all-reduce per
        module, and everything after it (global norm, Adam) runs replicated."""
        cfg, eng = self.config, self._engine
        if not self._lr_schedules:
            raise ValueError("update() needs the optimiser settings of LDPAgent.create (lr, end_lr, idm_lr, idm_end_lr, warmup_steps, decay_steps)")
        seed = _seed_of(rng)
        oh = cfg["obs_horizon"]
        nz = noise or {}
        nb = self._postprocess(batch)
        if "actions" not in nb:
            raise KeyError("update needs batch['actions'] (utils/data_utils.py:73)")
        obs_emb = self.get_obs_cond(nb["obs"]).contiguous()
        action = nb["actions"]
        emb_i, action_i = obs_emb, action
        if mixed_batch is not None:
            nbm = self._postprocess(mixed_batch)
            emb_i, action_i = self.get_obs_cond(nbm["obs"]).contiguous(), nbm["actions"]
        B, Bi = obs_emb.shape[0], emb_i.shape[0]
        lo_p, n_p = (0, B) if shard is None else shard["rows"]
        lo_i, n_i = (0, Bi) if shard is None else shard.get("mixed_rows", shard["rows"]) if mixed_batch is not None else shard["rows"]
        w_p, w_i = np.float32(B) / np.float32(n_p), np.float32(Bi) / np.float32(n_i)     # 1 without shards

        def rows_of(x, lo, n_loc, n_glob, per=1):
            """An explicit parity input given for the global batch -> this rank's rows."""
            return x[lo * per:(lo + n_loc) * per] if len(x) == n_glob * per and n_glob != n_loc else x
        hg = np.random.Generator(np.random.PCG64(seed & (2**63 - 1)))
        zero = torch.zeros((), dtype=torch.float32, device=self._device)
        plan_loss = idm_loss = zero
        mods = []
        # The two networks' gradients are independent: the IDM's tape (0.5 ms at 256 samples) is enqueued FIRST, on a second stream, and runs next to
        # the planner's (csrc/train.hip keeps one workspace lane per module); the statistics scalars read inputs only and go to a third.  The
        # main stream waits for both before anything reads a gradient or a scalar.
        if use_planner:                                               # (a (re)load of a module's training state synchronises the device: before anything is in flight)
            self._train_sync("planner", self.planner_state, self._planner_shapes())
        if use_idm:
            self._train_sync("idm", self.idm_state, self._idm_shapes())
        main = torch.cuda.current_stream(self._device)
        side = eng.aux_streams() if eng.get_option("train_streams") else {}
        idm_stream = side.get("idm") if (use_planner and use_idm) else None
        stats_stream = side.get("stats")
        if stats_stream is not None:
            stats_stream.wait_stream(main)
        with torch.cuda.stream(stats_stream if stats_stream is not None else main):
            stats = [eng.reduce_stats(obs_emb), eng.reduce_stats(action)] + [eng.reduce_stats(nb["obs"][k]) for k in nb["obs"]]
        # host draws in the reference's order of use (planner, then IDM), whatever the enqueue order below
        t_plan = t_idm = None
        if use_planner:
            npl = int(cfg["planner_n_diffusion_steps"])
            t_plan = nz.get("t_plan")
            t_plan = rows_of(np.asarray(hg.integers(0, npl, size=n_p) if t_plan is None else t_plan).reshape(-1), lo_p, B, n_p)
        if use_idm:                                                   # idm_loss, :129-140
            s = torch.cat([emb_i[:, oh - 1:-1], emb_i[:, oh:]], dim=-1)
            s = s.reshape(-1, s.shape[-1]).contiguous()               # 'B H D -> (B H) D'
            a = action_i[:, :-1].reshape(-1, action_i.shape[-1]).contiguous()
            if a.shape[0] != s.shape[0]:
                raise ValueError(f"idm_loss pairs {s.shape[0]} transitions with {a.shape[0]} actions: the batch needs "
                                 "actions.shape[1] - 1 == obs.shape[1] - obs_horizon (agent/ldp_agent.py:130-131)")
            nid = int(cfg["idm_n_diffusion_steps"])
            H = a.shape[0] // Bi                                      # transitions per sample
            t_idm = nz.get("t_idm")
            t_idm = rows_of(np.asarray(hg.integers(0, nid, size=n_i * H) if t_idm is None else t_idm).reshape(-1), lo_i, Bi, n_i, H)
            eps_i = nz.get("noise_idm")
            eps_i = (self._t(rows_of(eps_i, lo_i, Bi, n_i, H)) if eps_i is not None
                     else _philox_normal(seed, lo_i * H * a.shape[-1], 0, 8, a.numel(), self._device).reshape(a.shape))
            if idm_stream is not None:
                idm_stream.wait_stream(main)                          # its inputs were written on the main stream
            with torch.cuda.stream(idm_stream if idm_stream is not None else main):
                idm_loss = eng.train_idm_grad(s, a, eps_i, t_idm, float(np.float32(self.alpha_idm) * w_i))
        if use_planner:                                               # plan_loss, :113-127
            nxt = obs_emb[:, oh:].contiguous()
            eps = nz.get("noise_plan")
            per = nxt.numel() // B
            eps = (self._t(rows_of(eps, lo_p, B, n_p)) if eps is not None
                   else _philox_normal(seed, lo_p * per, 0, 7, nxt.numel(), self._device).reshape(nxt.shape))
            cond = obs_emb[:, :oh].reshape(B, -1).contiguous()
            plan_loss = eng.train_planner_grad(nxt, eps, t_plan, cond, float(np.float32(self.alpha_planner) * w_p))
            mods.append("planner")
        if use_idm:
            mods.append("idm")
        for st in (idm_stream, stats_stream):
            if st is not None:
                main.wait_stream(st)
        if shard is not None:                                         # data parallel: sum of the B / n weighted shard gradients = the global batch's
            import torch.distributed as tdist
            for name in mods:
                tdist.all_reduce(eng.train_arena(name, eng.TRAIN_GRADS), group=shard.get("group"))
            both = torch.stack([plan_loss.reshape(()), idm_loss.reshape(())])
            tdist.all_reduce(both, group=shard.get("group"))
            plan_loss, idm_loss = both[0], both[1]
        rep = self.lr_schedule
        new_p, new_i = self.planner_state, self.idm_state
        m = {}
        if use_planner:
            st = self.planner_state
            eng.train_apply("planner", float(np.float32(self._lr_schedules["planner"](st.step))))
            m["planner_lr"], m["planner_step"] = np.float32(rep(st.step)), st.step        # the OLD state's step, the LAST-built schedule (:257-258)
            new_p = self._trained_state("planner", st, self._planner_shapes())
        else:
            m.update(planner_lr=0, planner_step=0, noise_diff=0)
        if use_idm:
            st = self.idm_state
            eng.train_apply("idm", float(np.float32(self._lr_schedules["idm"](st.step))))
            m["idm_lr"], m["idm_step"] = np.float32(rep(st.step)), st.step
            new_i = self._trained_state("idm", st, self._idm_shapes())
        else:
            m.update(idm_lr=0, idm_step=0)
        # linear_algebra.global_norm(grads), :253 -- a metric only (nothing is clipped), taken after the optimiser launches, which leave the
        # per-stripe sums of squares of the gradients they consumed behind (one small launch instead of a second pass over the arenas)
        g_norm = eng.train_grad_norm(mods) if mods else zero
        arrs = [DeviceArray(x) for x in (plan_loss, idm_loss, g_norm)] + [DeviceArray(x) for x in stats]
        # (alpha_planner / alpha_idm are already inside the two device scalars: the gradients are those of the weighted losses)
        m.update(plan_loss=_HostScalar(lambda: arrs[0].numpy()), idm_loss=_HostScalar(lambda: arrs[1].numpy()),
                 loss=_HostScalar(lambda: arrs[0].numpy() + arrs[1].numpy()), g_norm=_HostScalar(lambda: arrs[2].numpy()))
        m["emb_min"], m["emb_max"], m["emb_mean"], m["emb_std"] = (_Elem(arrs[3], i) for i in range(4))
        m["action_min"], m["action_max"] = _Elem(arrs[4], 0), _Elem(arrs[4], 1)
        for j, k in enumerate(nb["obs"]):
            m[f"{k}_min"], m[f"{k}_max"] = _Elem(arrs[5 + j], 0), _Elem(arrs[5 + j], 1)
        return self.replace(planner_state=new_p, idm_state=new_i), m

    # ---- agent/ldp_agent.py:113-180, 328-349: the training losses, FORWARD ONLY ---------------------------
    def get_metrics(self, batch, rng, noise=None):
        """`get_metrics_step`: postprocess_batch -> get_obs_cond -> plan_loss / idm_loss (add_noise at a random timestep per sample, ONE
        network evaluation each, MSE against the noise) -> the statistics scalars.  No gradient anywhere: this is the call eval_bc.py:127
        makes before its sampling metrics.  Keys as the reference's `loss` (:141-180): plan_loss, idm_loss, loss, emb_min / max / mean /
        std, action_min / max, `<obs key>_min` / `_max` for every key of the (normalised) batch.

        rng: seed of the timesteps (host PCG64) and of the noise (the device Philox primitive); JAX's threefry stream is a non-goal.
        noise: optional dict(t_plan (B,), noise_plan (B, T, D), t_idm (R,), noise_idm (R, A)) -- explicit inputs for parity runs."""
        cfg, eng = self.config, self._engine
        seed = _seed_of(rng)
        oh = cfg["obs_horizon"]
        nz = noise or {}

        def run():
            self._sync_weights()
            nb = self._postprocess(batch)                           # needs 'actions' like postprocess_batch (utils/data_utils.py:70-74)
            if "actions" not in nb:
                raise KeyError("get_metrics needs batch['actions'] (utils/data_utils.py:73)")
            obs_emb = self.get_obs_cond(nb["obs"]).contiguous()     # (the reference's loss reads pre-encoded latents: no vae_encode here)
            action = nb["actions"]
            B = obs_emb.shape[0]
            hg = np.random.Generator(np.random.PCG64(seed & (2**63 - 1)))
            plan_loss = idm_loss = None
            if self.use_planner:                                    # :113-127
                nxt = self._planner_targets(obs_emb)
                npl = int(cfg["planner_n_diffusion_steps"])
                t = nz.get("t_plan")
                t = torch.as_tensor(hg.integers(0, npl, size=B) if t is None else np.asarray(t)).to(self._device)
                eps = nz.get("noise_plan")
                eps = (self._t(eps) if eps is not None else
                       _philox_normal(seed, 0, 0, 7, nxt.numel(), self._device).reshape(nxt.shape))
                noisy = eng.add_noise(nxt, eps, t, npl)
                cond = obs_emb[:, :oh].reshape(B, -1).contiguous()
                pred = eng.unet_forward(noisy, t, cond)
                plan_loss = eng.mean_sq_diff(pred, eps)
            if self.use_idm:                                        # :129-140
                s, a = self._idm_pairs(obs_emb, action)
                nid = int(cfg["idm_n_diffusion_steps"])
                t = nz.get("t_idm")
                t = torch.as_tensor(hg.integers(0, nid, size=a.shape[0]) if t is None else np.asarray(t).reshape(-1)).to(self._device)
                eps = nz.get("noise_idm")
                eps = (self._t(eps) if eps is not None else
                       _philox_normal(seed, 0, 0, 8, a.numel(), self._device).reshape(a.shape))
                noisy = eng.add_noise(a, eps, t, nid)
                pred = self._idm_eps(s, noisy, t)
                idm_loss = eng.mean_sq_diff(pred, eps)
            zero = torch.zeros((), dtype=torch.float32, device=self._device)
            out = [zero if plan_loss is None else plan_loss, zero if idm_loss is None else idm_loss,
                   eng.reduce_stats(obs_emb), eng.reduce_stats(action)]
            out += [eng.reduce_stats(nb["obs"][k]) for k in nb["obs"]]
            return out
        rec = self._record(run)
        res = self._guarded(run)
        rec.seqs = self._seqs()
        keys = list(self._postprocess_keys(batch))
        arrs = [DeviceArray(t, record=rec) for t in res]
        # the two mean-squared errors are device scalars; alpha_planner / alpha_idm and the sum (agent/ldp_agent.py:146-158) are applied in
        # float32 when a value is read
        ap, ai = np.float32(self.alpha_planner if self.use_planner else 0), np.float32(self.alpha_idm if self.use_idm else 0)
        m = dict(plan_loss=_HostScalar(lambda: ap * arrs[0].numpy()), idm_loss=_HostScalar(lambda: ai * arrs[1].numpy()),
                 loss=_HostScalar(lambda: ap * arrs[0].numpy() + ai * arrs[1].numpy()))
        # the statistics live in 4-vectors (min, max, mean, std) on the device; a metric is one element of its vector
        m["emb_min"], m["emb_max"], m["emb_mean"], m["emb_std"] = (_Elem(arrs[2], i) for i in range(4))
        m["action_min"], m["action_max"] = _Elem(arrs[3], 0), _Elem(arrs[3], 1)
        for j, k in enumerate(keys):
            m[f"{k}_min"], m[f"{k}_max"] = _Elem(arrs[4 + j], 0), _Elem(arrs[4 + j], 1)
        return m

    # what the two losses read of a training batch (LDPHierAgent overrides the three: strided targets, a U-Net IDM)
    def _planner_targets(self, obs_emb):
        return obs_emb[:, self.config["obs_horizon"]:].contiguous()                                       # agent/ldp_agent.py:117

    def _idm_pairs(self, obs_emb, action):
        oh = self.config["obs_horizon"]
        s = torch.cat([obs_emb[:, oh - 1:-1], obs_emb[:, oh:]], dim=-1)
        s = s.reshape(-1, s.shape[-1]).contiguous()                                                       # 'B H D -> (B H) D'
        a = action[:, :-1].reshape(-1, action.shape[-1]).contiguous()
        if a.shape[0] != s.shape[0]:
            raise ValueError(f"idm_loss pairs {s.shape[0]} transitions with {a.shape[0]} actions: the batch needs "
                             "actions.shape[1] - 1 == obs.shape[1] - obs_horizon (agent/ldp_agent.py:130-131)")
        return s, a

    def _idm_eps(self, s, noisy, t):
        return self._engine.idm_forward(s, noisy, t)

    @staticmethod
    def _postprocess_keys(batch):
        return batch["obs"].keys()
