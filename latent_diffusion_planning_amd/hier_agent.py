"""LDPHierAgent -- the sampling surface of the reference's hierarchical variant (agent/ldp_hier_agent.py:385-468)
on the HIP engine.  SURVEY.md section 8(f), unranked tail: built last; sampling (round 4) and the training step (round 6).

The planner (`ConditionalUnet1D`, down_dims [256, 512, 1024]) predicts every `idm_horizon`-th state -- a trajectory
of pred_horizon // idm_horizon latent states --, and the inverse-dynamics model is a SECOND `ConditionalUnet1D`
(agent/ldp_hier_agent.yaml:18-26: down_dims [256, 512], input_dim = action_dim, global_cond_dim = 2 obs_dim) that
denoises a chunk of `idm_horizon` actions per (state, next state) transition.  Both run on `ldp::tconv_kernel`: two
engine handles, each replaying its loop from one hipGraph.

Same names / argument meaning / return structure as the reference class: `create` (with `idm_horizon`), `sample`,
`sample_viz` -> (action (B, action_horizon * idm_horizon, A), {'plan_viz'[, 'plan_mse']}), `sample_action` (round 5), `get_params`, `.config`,
`.replace`, `.planner_state / .idm_state`, `vae_encode / vae_decode / get_obs_cond` (inherited from LDPAgent: the
reference's two classes share them line for line).  `metrics` additionally carries 'plan' ((B, action_horizon + 1, D),
the states the actions connect) -- the reference pops it.  `update` / `update_mixed` (round 6) train both U-Nets; `get_metrics` is the
forward-only evaluation of the same two losses.

What the reference's shipped configuration cannot do is refused with the reason: train_bc.yaml gives pred_horizon 15
and idm_horizon 4, i.e. a planner trajectory of 3 states, which the three-level U-Net cannot process (its skip
connections need a multiple of 4: networks/diffusion_nets_v2.py:141-156 would concatenate lengths 1 and 2).
"""
from __future__ import annotations

import numpy as np
import torch

from . import weights as W
from .agent import LATENT_SHAPES, LDPAgent, ParamState, _as_flat, _get, _norm_entry, _seed_of, load_pretrained_vae
from .arrays import DeviceArray
from .engine import HipEngine


class LDPHierAgent(LDPAgent):
    @classmethod
    def create(cls, rng, batch, shape_meta,
               name, planner, idm_net,
               vae_pretrain_path, vae_feature_dim,
               use_planner, use_idm,
               lowdim_obs, rgb_obs, obs_normalization, data_name,
               obs_horizon, pred_horizon, action_horizon,
               planner_n_diffusion_steps, idm_n_diffusion_steps,
               alpha_planner=1, alpha_idm=1,
               lr=None, end_lr=None, idm_lr=None, idm_end_lr=None,
               warmup_steps=None, decay_steps=None, idm_horizon=4,
               update_planner_every=1, update_idm_every=1, update_idm_after=-1,
               update_planner_until=-1, update_planner_after=-1, grad_clip=None,
               device=None, vae_params=None, exclusive_gpu=True):
        """agent/ldp_hier_agent.py:471-640."""
        lowdim_obs, rgb_obs = list(lowdim_obs), list(rgb_obs)
        if len(rgb_obs) > 1:
            raise NotImplementedError("more than one rgb_obs key: the reference's get_obs_cond concatenates cameras on "
                                      "axis 1 and is only well-defined for one")
        if int(vae_feature_dim) not in LATENT_SHAPES:
            raise NotImplementedError(f"vae_feature_dim={vae_feature_dim}: built latent shapes are {sorted(LATENT_SHAPES)}")
        idm_horizon, pred_horizon, action_horizon = int(idm_horizon), int(pred_horizon), int(action_horizon)
        assert action_horizon % idm_horizon == 0                                       # agent/ldp_hier_agent.py:618
        p_down = tuple(int(d) for d in _get(planner, "down_dims", (256, 512, 1024)))
        i_down = tuple(int(d) for d in _get(idm_net, "down_dims", (256, 512)))
        t_plan = pred_horizon // idm_horizon
        if t_plan % (1 << (len(p_down) - 1)) != 0:
            raise ValueError(f"pred_horizon {pred_horizon} // idm_horizon {idm_horizon} = {t_plan} planner states: the "
                             f"{len(p_down)}-level ConditionalUnet1D needs a multiple of {1 << (len(p_down) - 1)} (its skip "
                             "connections concatenate equal lengths, networks/diffusion_nets_v2.py:141-156); the reference's "
                             "own train_bc.yaml (horizon 16, idm_horizon 4 -> 3 states) fails there too")
        if idm_horizon % (1 << (len(i_down) - 1)) != 0:
            raise ValueError(f"idm_horizon {idm_horizon}: the {len(i_down)}-level IDM U-Net needs a multiple of {1 << (len(i_down) - 1)}")
        if action_horizon > t_plan:
            raise ValueError(f"action_horizon {action_horizon} states are taken from a plan of {t_plan} "
                             "(agent/ldp_hier_agent.py:431-433)")
        for nm, dd in (("planner", p_down), ("idm_net", i_down)):
            if any(d < 256 or d % 128 for d in dd):
                raise NotImplementedError(f"{nm} down_dims={dd}: the MFMA conv tiles are built for levels that are multiples "
                                          "of 128 channels and at least 256 wide")
        side, latent_ch = LATENT_SHAPES[int(vae_feature_dim)]
        lowdim_dim = sum(int(np.prod(shape_meta["all_shapes"][k])) for k in lowdim_obs)
        obs_dim = lowdim_dim + int(vae_feature_dim) * len(rgb_obs)
        action_dim = int(shape_meta["ac_dim"])
        if obs_dim > 128 or 2 * obs_dim * 1 > 4096:
            raise NotImplementedError(f"obs_dim={obs_dim}: at most 128 features")
        seed = _seed_of(rng)
        pspec = W.PlannerSpec(input_dim=obs_dim, global_cond_dim=obs_dim * int(obs_horizon),
                              diffusion_step_embed_dim=int(_get(planner, "diffusion_step_embed_dim", 256)), down_dims=p_down)
        ispec = W.PlannerSpec(input_dim=action_dim, global_cond_dim=2 * obs_dim,
                              diffusion_step_embed_dim=int(_get(idm_net, "diffusion_step_embed_dim", 256)), down_dims=i_down)
        planner_state = ParamState(W.init_planner_params(pspec, seed=seed * 3 + 1, perturb=False)) if use_planner else None
        idm_state = ParamState(W.init_planner_params(ispec, seed=seed * 3 + 2, perturb=False)) if use_idm else None
        if vae_params is None and vae_pretrain_path is not None:
            vae_params = load_pretrained_vae(str(vae_pretrain_path))
        vae_params = _as_flat(vae_params) if vae_params is not None else None
        config = dict(planner_n_diffusion_steps=int(planner_n_diffusion_steps), idm_n_diffusion_steps=int(idm_n_diffusion_steps),
                      lowdim_obs=lowdim_obs, rgb_obs=rgb_obs, obs_horizon=int(obs_horizon), name=name, action_dim=action_dim,
                      pred_horizon=pred_horizon, action_horizon=action_horizon, idm_horizon=idm_horizon, obs_dim=obs_dim,
                      update_planner_every=update_planner_every, update_idm_every=update_idm_every,
                      update_planner_until=update_planner_until, update_planner_after=update_planner_after,
                      update_idm_after=update_idm_after, vae_feature_dim=int(vae_feature_dim), data_name=data_name)
        norm = {"obs": {k: _norm_entry(v) for k, v in dict(obs_normalization["obs"]).items()}}
        if "actions" in obs_normalization:
            norm["actions"] = _norm_entry(obs_normalization["actions"])
        if not torch.cuda.is_available():
            from ._lib import LDPHipUnavailable
            raise LDPHipUnavailable("no HIP device visible: LDPHierAgent has no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # handle 1: planner U-Net + StableVAE; handle 2: the IDM U-Net (its "planner" module)
        eng = HipEngine(obs_dim=obs_dim, action_dim=action_dim, global_cond_dim=pspec.global_cond_dim, pred_horizon=t_plan,
                        action_horizon=min(action_horizon, t_plan), down_dims=p_down, planner_train_steps=int(planner_n_diffusion_steps),
                        idm_train_steps=int(idm_n_diffusion_steps), image_size=32 * side, vae_latent_channels=latent_ch, device=dev)
        idm_eng = HipEngine(obs_dim=action_dim, action_dim=action_dim, global_cond_dim=2 * obs_dim, pred_horizon=idm_horizon,
                            action_horizon=idm_horizon, down_dims=i_down, planner_train_steps=int(idm_n_diffusion_steps),
                            idm_train_steps=int(idm_n_diffusion_steps), device=dev)
        if not exclusive_gpu:
            eng.set_option("safe_mode", 1)
            idm_eng.set_option("safe_mode", 1)
        from .schedule import warmup_cosine_decay_schedule
        sched = {}                                                                      # agent/ldp_hier_agent.py:533-540, 567-574
        if lr is not None and warmup_steps is not None and decay_steps is not None:
            if use_planner:
                sched["planner"] = warmup_cosine_decay_schedule(float(end_lr), float(lr), int(warmup_steps), int(decay_steps), float(end_lr))
            if use_idm:
                sched["idm"] = warmup_cosine_decay_schedule(float(idm_end_lr if idm_end_lr is not None else end_lr),
                                                            float(idm_lr if idm_lr is not None else lr), int(warmup_steps), int(decay_steps),
                                                            float(idm_end_lr if idm_end_lr is not None else end_lr))
        self = cls(planner_state, idm_state, vae_params, norm, use_planner, use_idm, alpha_planner, alpha_idm, config, eng,
                   pspec, None, W.VAESpec(latent_channels=latent_ch), dev, lr_schedules=sched)
        self._idm_engine, self._idm_unet_spec = idm_eng, ispec
        return self

    def _engines(self):
        """Both handles take part in every policy call: the fault protocol (LDPAgent._record / _guarded) polls and
        acknowledges each of them -- a fault on the IDM handle marks the call suspect like one on the planner handle."""
        return [self._engine, self._idm_engine]

    # the IDM is a U-Net here: it lives in the second handle's planner slot
    def _sync_weights(self, need_vae=False):
        held = self._engine.loaded
        up, ver = {}, {}
        if self.use_planner and held["planner"] != self.planner_state.version:
            if self._engine.train_token.get("planner") == self.planner_state.version:       # the state an update() left in the training arenas
                self._engine.train_publish(["planner"], versions={"planner": self.planner_state.version})
            else:
                W.check_params(self.planner_state.params, W.planner_shapes(self._planner_spec))
                up["planner"], ver["planner"] = self.planner_state.params, self.planner_state.version
        if need_vae and held["vae"] != self._vae_version:
            if self.vae_params is None:
                raise ValueError("raw image observations need VAE weights (vae_pretrain_path / vae_params)")
            up["vae"], ver["vae"] = self.vae_params, self._vae_version
        if up:
            self._engine.load_params(**up, versions=ver)
        if self.use_idm and self._idm_engine.loaded["planner"] != self.idm_state.version:
            if self._idm_engine.train_token.get("planner") == self.idm_state.version:
                self._idm_engine.train_publish(["planner"], versions={"planner": self.idm_state.version})
            else:
                W.check_params(self.idm_state.params, W.planner_shapes(self._idm_unet_spec))
                self._idm_engine.load_params(planner=self.idm_state.params, versions={"planner": self.idm_state.version})

    # ---- agent/ldp_hier_agent.py:385-461 -----------------------------------------------------------
    def sample(self, batch, eval_rng, **kw):
        kw.setdefault("decode", False)
        return self.sample_viz(batch, eval_rng, **kw)

    def sample_viz(self, batch, eval_rng, noise=None, decode=True, row_offset=0, sampler="ddpm", n_steps=None,
                   idm_steps=None):
        """noise: optional dict(x_init (B, Tp, D), x_noise (S, B, Tp, D), a_init (B*ah, ih, A), a_noise (S, B*ah, ih, A))."""
        if not (self.use_planner and self.use_idm):
            raise NotImplementedError("sample() needs both the planner and the IDM")
        cfg = self.config
        seed = _seed_of(eval_rng)
        oh, ih, ah, D = cfg["obs_horizon"], cfg["idm_horizon"], cfg["action_horizon"], cfg["obs_dim"]
        if idm_steps is None and n_steps is not None and sampler != "ddpm":
            idm_steps = n_steps
        nz = noise or {}

        def run():
            self._sync_weights()
            nb = self._postprocess(batch)
            obs = self._vae_encode_t(nb["obs"])
            obs_emb = self.get_obs_cond(obs).contiguous()
            B = obs_emb.shape[0]
            cond = obs_emb[:, :oh].reshape(B, -1).contiguous()
            nxt = self._engine.plan_sample(cond, x_init=nz.get("x_init"), step_noise=nz.get("x_noise"), seed=seed,
                                           row_offset=row_offset, sampler=sampler, n_steps=n_steps)        # (B, Tp, D)
            plan = torch.cat([obs_emb[:, oh - 1:oh], nxt[:, :ah]], dim=1)                                 # :431-436
            trans = torch.cat([plan[:, :-1], plan[:, 1:]], dim=-1).reshape(-1, 2 * D).contiguous()        # 'B H D -> (B H) D'
            a = self._idm_engine.plan_sample(trans, x_init=nz.get("a_init"), step_noise=nz.get("a_noise"), seed=seed + 1,
                                             row_offset=row_offset * ah, sampler=sampler, n_steps=idm_steps)   # (B*ah, ih, A)
            action = self._apply_norm(a.reshape(B, ah * ih, -1), self.obs_normalization["actions"], False)
            out = [action, plan]
            if obs_emb.shape[1] > oh:                                                                     # :399-400
                out.append(self._engine.mean_sq_diff(nxt, obs_emb[:, oh:].contiguous()))
            return out
        rec = self._record(lambda: run() + [None])
        res = self._guarded(run)
        rec.seqs = self._seqs()
        action, plan = DeviceArray(res[0], record=rec), DeviceArray(res[1], record=rec)
        metrics = {"plan": plan}
        if len(res) > 2:
            metrics["plan_mse"] = DeviceArray(res[2], record=rec)
        S = self._engine.image_size
        viz = DeviceArray(thunk=lambda: torch.repeat_interleave(self._vae_decode_t(plan.tensor)[:, 1:], ih, dim=1),   # :437-438
                          shape=(plan.shape[0], ah * ih, 3, S, S), record=rec)
        if decode:
            viz.tensor
        metrics["plan_viz"] = viz
        return action, metrics

    # ---- agent/ldp_hier_agent.py:345-383 -----------------------------------------------------------
    def sample_action(self, batch, eval_rng, noise=None, row_offset=0, sampler="ddpm", n_steps=None):
        """The IDM on the batch's OWN consecutive frames (ground-truth "plan"): every (frame, next frame) pair of the (B, H, obs_dim)
        observation embedding gets a chunk of `idm_horizon` actions from the IDM U-Net -> (B, (H - 1) * idm_horizon, A), un-normalised.
        noise: optional dict(a_init (B*(H-1), ih, A), a_noise (S, B*(H-1), ih, A))."""
        if not self.use_idm:
            raise NotImplementedError("sample_action needs the IDM (use_idm)")
        seed = _seed_of(eval_rng)
        ih, D = self.config["idm_horizon"], self.config["obs_dim"]
        nz = noise or {}

        def run():
            self._sync_weights()
            nb = self._postprocess(batch)
            plan = self.get_obs_cond(self._vae_encode_t(nb["obs"]))                                          # (B, H, D)
            B = plan.shape[0]
            trans = torch.cat([plan[:, :-1], plan[:, 1:]], dim=-1).reshape(-1, 2 * D).contiguous()        # 'B H D -> (B H) D' (:363-364)
            a = self._idm_engine.plan_sample(trans, x_init=nz.get("a_init"), step_noise=nz.get("a_noise"), seed=seed,
                                             row_offset=row_offset * (plan.shape[1] - 1), sampler=sampler, n_steps=n_steps)
            return [self._apply_norm(a.reshape(B, -1, a.shape[-1]), self.obs_normalization["actions"], False)]   # '(B H) T D -> B (H T) D'
        rec = self._record(run)
        res = self._guarded(run)
        rec.seqs = self._seqs()
        return DeviceArray(res[0], record=rec)

    # ---- agent/ldp_hier_agent.py:324-343 `get_metrics`: LDPAgent's, forward only, with this agent's three readings of a training batch -----------
    # (the reference's own evaluation skips the call for this agent -- eval_bc.py:107-109 returns an empty dict, and so does harness.eval_loss_metrics)
    def _planner_targets(self, obs_emb):
        return obs_emb[:, self.config["obs_horizon"]::self.config["idm_horizon"]].contiguous()            # :115

    def _idm_pairs(self, obs_emb, action):
        oh, ih = self.config["obs_horizon"], self.config["idm_horizon"]
        s = torch.cat([obs_emb[:, oh - 1:-1:ih], obs_emb[:, oh - 1 + ih::ih]], dim=-1)
        s = s.reshape(-1, s.shape[-1]).contiguous()                                                       # 'B H D -> (B H) D', :126
        a = action[:, oh - 1:-1]
        if a.shape[1] % ih != 0 or a.shape[0] * (a.shape[1] // ih) != s.shape[0]:
            raise ValueError(f"idm_loss pairs {s.shape[0]} (state, state + {ih}) transitions with {a.shape[1]} actions per sample: the batch needs "
                             f"actions.shape[1] - obs_horizon a multiple of idm_horizon and one chunk per transition (agent/ldp_hier_agent.py:126-128)")
        return s, a.reshape(a.shape[0], -1, ih, a.shape[-1]).reshape(-1, ih, a.shape[-1]).contiguous()    # 'B K H D -> (B K) H D'

    def _idm_eps(self, s, noisy, t):
        return self._idm_engine.unet_forward(noisy, t, s)

    # ---- agent/ldp_hier_agent.py:111-137, 223-322: the training step ------------------------------------------------
    # `update` / `update_mixed` are LDPAgent's (the gating is the same code, :223-232 / :274-283); what differs is the step:
    def _update_step(self, batch, mixed_batch, rng, use_planner, use_idm, noise, shard=None):
        """plan_loss on every `idm_horizon`-th future state (:111-123), idm_loss of the action U-Net on chunks of `idm_horizon` actions per
        (state, state + idm_horizon) pair (:125-137), jax.grad + global_norm + one optax.adam step per network (:234-272).  Both networks are
        ConditionalUnet1Ds: each trains in its own engine handle's planner slot (csrc/train.hip's U-Net tape), the two tapes on two streams.
        noise: optional dict(t_plan (B,), noise_plan (B, Tp, D), t_idm (B K,), noise_idm (B K, ih, A)) for parity runs."""
        from .agent import _Elem, _HostScalar, _philox_normal
        if not self._lr_schedules:
            raise ValueError("update() needs the optimiser settings of LDPHierAgent.create (lr, end_lr, idm_lr, idm_end_lr, warmup_steps, decay_steps)")
        cfg, eng, ieng = self.config, self._engine, self._idm_engine
        seed = _seed_of(rng)
        oh, ih = cfg["obs_horizon"], cfg["idm_horizon"]
        nz = noise or {}
        nb = self._postprocess(batch)
        if "actions" not in nb:
            raise KeyError("update needs batch['actions'] (utils/data_utils.py:73)")
        obs_emb = self.get_obs_cond(nb["obs"]).contiguous()
        action = nb["actions"]
        emb_i, action_i = obs_emb, action
        if mixed_batch is not None:                                                     # loss_mixed, :180-203
            nbm = self._postprocess(mixed_batch)
            emb_i, action_i = self.get_obs_cond(nbm["obs"]).contiguous(), nbm["actions"]
        B, Bi = obs_emb.shape[0], emb_i.shape[0]
        # shard (dist.update_sharded): this rank holds rows [lo, lo + B) of a global batch of n -- global-row timesteps and noise, B / n weighted
        # losses, one all-reduce per handle's gradient arena (LDPAgent._update_step has the same contract)
        lo_p, n_p = (0, B) if shard is None else shard["rows"]
        lo_i, n_i = (0, Bi) if shard is None else (shard.get("mixed_rows", shard["rows"]) if mixed_batch is not None else shard["rows"])
        w_p, w_i = np.float32(B) / np.float32(n_p), np.float32(Bi) / np.float32(n_i)

        def rows_of(x, lo, n_loc, n_glob, per=1):
            return x[lo * per:(lo + n_loc) * per] if len(x) == n_glob * per and n_glob != n_loc else x
        hg = np.random.Generator(np.random.PCG64(seed & (2**63 - 1)))
        zero = torch.zeros((), dtype=torch.float32, device=self._device)
        plan_loss = idm_loss = zero
        if use_planner:
            self._train_sync("planner", self.planner_state, W.planner_shapes(self._planner_spec))
        if use_idm:
            self._train_sync("planner", self.idm_state, W.planner_shapes(self._idm_unet_spec), eng=ieng)
        main = torch.cuda.current_stream(self._device)
        side = eng.aux_streams() if eng.get_option("train_streams") else {}
        idm_stream = side.get("idm") if (use_planner and use_idm) else None
        stats_stream = side.get("stats")
        if stats_stream is not None:
            stats_stream.wait_stream(main)
        with torch.cuda.stream(stats_stream if stats_stream is not None else main):
            stats = [eng.reduce_stats(obs_emb), eng.reduce_stats(action)] + [eng.reduce_stats(nb["obs"][k]) for k in nb["obs"]]
        t_plan = None
        if use_planner:
            t_plan = nz.get("t_plan")
            t_plan = rows_of(np.asarray(hg.integers(0, int(cfg["planner_n_diffusion_steps"]), size=n_p) if t_plan is None else t_plan).reshape(-1), lo_p, B, n_p)
        if use_idm:                                                                     # :125-137
            s, a = self._idm_pairs(emb_i, action_i)
            K = a.shape[0] // Bi                                                        # chunks per sample
            t_idm = nz.get("t_idm")
            t_idm = rows_of(np.asarray(hg.integers(0, int(cfg["idm_n_diffusion_steps"]), size=n_i * K) if t_idm is None else t_idm).reshape(-1), lo_i, Bi, n_i, K)
            eps_i = nz.get("noise_idm")
            eps_i = (self._t(rows_of(eps_i, lo_i, Bi, n_i, K)) if eps_i is not None
                     else _philox_normal(seed, lo_i * K * a.shape[1] * a.shape[2], 0, 8, a.numel(), self._device).reshape(a.shape))
            if idm_stream is not None:
                idm_stream.wait_stream(main)
            with torch.cuda.stream(idm_stream if idm_stream is not None else main):
                idm_loss = ieng.train_planner_grad(a, eps_i, t_idm, s, float(np.float32(self.alpha_idm) * w_i))
        if use_planner:                                                                 # :111-123
            nxt = self._planner_targets(obs_emb)
            eps = nz.get("noise_plan")
            eps = (self._t(rows_of(eps, lo_p, B, n_p)) if eps is not None
                   else _philox_normal(seed, lo_p * (nxt.numel() // B), 0, 7, nxt.numel(), self._device).reshape(nxt.shape))
            cond = obs_emb[:, :oh].reshape(B, -1).contiguous()
            plan_loss = eng.train_planner_grad(nxt, eps, t_plan, cond, float(np.float32(self.alpha_planner) * w_p))
        for st in (idm_stream, stats_stream):
            if st is not None:
                main.wait_stream(st)
        if shard is not None:
            import torch.distributed as tdist
            if use_planner:
                tdist.all_reduce(eng.train_arena("planner", eng.TRAIN_GRADS), group=shard.get("group"))
            if use_idm:
                tdist.all_reduce(ieng.train_arena("planner", ieng.TRAIN_GRADS), group=shard.get("group"))
            both = torch.stack([plan_loss.reshape(()), idm_loss.reshape(())])
            tdist.all_reduce(both, group=shard.get("group"))
            plan_loss, idm_loss = both[0], both[1]
        rep = self.lr_schedule
        new_p, new_i = self.planner_state, self.idm_state
        m = {}
        norms = []
        if use_planner:
            st = self.planner_state
            eng.train_apply("planner", float(np.float32(self._lr_schedules["planner"](st.step))))
            m["planner_lr"], m["planner_step"] = np.float32(rep(st.step)), st.step       # the OLD state's step, the LAST-built schedule (:254-255)
            new_p = self._trained_state("planner", st, W.planner_shapes(self._planner_spec))
            norms.append(eng.train_grad_norm(["planner"]))
        else:
            m.update(planner_lr=0, planner_step=0, noise_diff=0)
        if use_idm:
            st = self.idm_state
            ieng.train_apply("planner", float(np.float32(self._lr_schedules["idm"](st.step))))
            m["idm_lr"], m["idm_step"] = np.float32(rep(st.step)), st.step
            new_i = self._trained_state("planner", st, W.planner_shapes(self._idm_unet_spec), eng=ieng)
            norms.append(ieng.train_grad_norm(["planner"]))
        else:
            m.update(idm_lr=0, idm_step=0)
        # linear_algebra.global_norm over BOTH gradient trees (:250): the two handles' norms combine as sqrt(a^2 + b^2)
        g_norm = zero if not norms else norms[0] if len(norms) == 1 else torch.sqrt(norms[0].double() ** 2 + norms[1].double() ** 2).float()
        arrs = [DeviceArray(x) for x in (plan_loss, idm_loss, g_norm)] + [DeviceArray(x) for x in stats]
        m.update(plan_loss=_HostScalar(lambda: arrs[0].numpy()), idm_loss=_HostScalar(lambda: arrs[1].numpy()),
                 loss=_HostScalar(lambda: arrs[0].numpy() + arrs[1].numpy()), g_norm=_HostScalar(lambda: arrs[2].numpy()))
        m["emb_min"], m["emb_max"], m["emb_mean"], m["emb_std"] = (_Elem(arrs[3], i) for i in range(4))
        m["action_min"], m["action_max"] = _Elem(arrs[4], 0), _Elem(arrs[4], 1)
        for j, k in enumerate(nb["obs"]):
            m[f"{k}_min"], m[f"{k}_max"] = _Elem(arrs[5 + j], 0), _Elem(arrs[5 + j], 1)
        return self.replace(planner_state=new_p, idm_state=new_i), m

    # (the reference's hierarchical class has no sample_action_from_plan)
    def sample_action_from_plan(self, *a, **k):
        raise NotImplementedError("LDPHierAgent has no sample_action_from_plan (neither has agent/ldp_hier_agent.py)")
