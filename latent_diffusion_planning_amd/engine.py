"""HipEngine: PyTorch-ROCm tensors in, libldp_hip.so (hand-written gfx950 kernels) out.

torch is plumbing here -- device memory, the current stream, `torch.distributed` -- the
arithmetic of the hot path all happens inside the C-ABI calls.  Every method enqueues on
`torch.cuda.current_stream()` and returns without synchronising.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import (LDPHipError, LDPHipFault, LdpConfig, MOD_IDM, MOD_PLANNER, MOD_VAE, SAMPLER_DDIM, SAMPLER_DDPM, check)

_SAMPLERS = {"ddpm": SAMPLER_DDPM, "ddim": SAMPLER_DDIM}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, device) -> torch.Tensor:
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _want(name: str, t: Optional[torch.Tensor], shape) -> None:
    """Raw pointers cross the C ABI next: a wrong shape would be an out-of-bounds device read there,
    where the JAX reference raises a shape error."""
    if t is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")


class HipEngine:
    """One `ldp_handle` on one GPU."""

    def __init__(self, *, obs_dim: int, action_dim: int, global_cond_dim: int, pred_horizon: int,
                 action_horizon: int, down_dims: Sequence[int] = (256, 512, 1024), kernel_size: int = 5,
                 n_groups: int = 8, step_embed_dim: int = 256, planner_train_steps: int = 100,
                 idm_train_steps: int = 100, idm_hidden: int = 256, idm_blocks: int = 3,
                 idm_time_dim: int = 256, image_size: int = 64, vae_latent_channels: int = 4,
                 device: Optional[torch.device] = None):
        self.lib = _lib.load()                     # raises loudly when the extension is missing
        if not torch.cuda.is_available():
            raise _lib.LDPHipUnavailable("no HIP device visible: the LDP hot path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cfg = LdpConfig()
        cfg.obs_dim, cfg.action_dim, cfg.global_cond_dim = obs_dim, action_dim, global_cond_dim
        cfg.pred_horizon, cfg.action_horizon = pred_horizon, action_horizon
        cfg.n_levels = len(down_dims)
        for i, d in enumerate(down_dims):
            cfg.down_dims[i] = int(d)
        cfg.kernel_size, cfg.n_groups, cfg.step_embed_dim = kernel_size, n_groups, step_embed_dim
        cfg.planner_train_steps, cfg.idm_train_steps = planner_train_steps, idm_train_steps
        cfg.idm_hidden, cfg.idm_blocks, cfg.idm_time_dim = idm_hidden, idm_blocks, idm_time_dim
        cfg.image_size, cfg.vae_latent_channels = image_size, vae_latent_channels
        cfg.device = self.device.index or 0
        self.cfg = cfg
        self.D, self.A, self.G, self.T = obs_dim, action_dim, global_cond_dim, pred_horizon
        self.ah = action_horizon
        self.image_size, self.latent_channels = int(image_size), int(vae_latent_channels)
        self.planner_train_steps, self.idm_train_steps = planner_train_steps, idm_train_steps
        # version token of the parameter tree last uploaded per module (LDPAgent compares it with its
        # ParamState.version: agents sharing one engine can never run on each other's weights)
        self.loaded = {"planner": None, "idm": None, "vae": None}
        # likewise for the training arenas (ldp_train_*): the token of the ParamState they currently represent
        self.train_token = {"planner": None, "idm": None}
        # fault bookkeeping for callers that keep results on the device: every sampling call gets a sequence
        # number; a detected fault marks every call enqueued so far as suspect (calls are asynchronous: the word
        # may have been set by any of them)
        self.call_seq = 0
        self.fault_upto = -1
        self.fault_kinds = 0                          # every kind of fault this handle ever reported (poll_fault_kinds)
        self.last_fault_kinds = 0
        self._bounds_cache = {}
        self._h = C.c_void_p()
        check(self.lib.ldp_create(C.byref(cfg), C.byref(self._h)))

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ldp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_params(self, planner: Optional[Dict[str, np.ndarray]] = None,
                    idm: Optional[Dict[str, np.ndarray]] = None,
                    vae: Optional[Dict[str, np.ndarray]] = None, versions: Optional[dict] = None) -> None:
        """Upload flat Flax-path parameter dicts and build the packed layouts / tables.
        versions: optional {module: token} recorded in `self.loaded` (None = anonymous upload)."""
        mods = 0
        for name, tree, bit in (("planner", planner, MOD_PLANNER), ("idm", idm, MOD_IDM),
                                ("vae", vae, MOD_VAE)):
            if tree is None:
                continue
            mods |= bit
            for path, arr in tree.items():
                a = np.ascontiguousarray(np.asarray(arr), dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                check(self.lib.ldp_set_weight(self._h, f"{name}/{path}".encode(),
                                              a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        with torch.cuda.device(self.device):
            check(self.lib.ldp_finalize(self._h, mods, self._stream()))
        for name, tree in (("planner", planner), ("idm", idm), ("vae", vae)):
            if tree is not None:
                self.loaded[name] = (versions or {}).get(name, object())

    # -- options / fault protocol (include/ldp_hip.h) -----------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        check(self.lib.ldp_set_option(self._h, name.encode(), C.c_int64(int(value))))

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        check(self.lib.ldp_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def active_debug_options(self) -> str:
        """Names of the timing-ablation options that are set (results are wrong while any is)."""
        on = []
        if self.get_option("dbg"):
            on.append(f"dbg={self.get_option('dbg')}")
        if self.get_option("repeat") != 1:
            on.append(f"repeat={self.get_option('repeat')}")
        return ",".join(on)

    FAULT_EXCHANGE, FAULT_RANGE = 1, 2

    def poll_fault_kinds(self) -> int:
        """No stream is synchronised.  Bit mask of the faults recorded since the last poll (include/ldp_hip.h):
        FAULT_EXCHANGE -- a split work-group timed out on its peer (the handle now runs in safe mode);
        FAULT_RANGE -- an operand left the range of the two-fp16-plane convolutions (the handle now runs them on three
        bf16 planes, which have fp32's range).  Either way the results enqueued since the previous poll are invalid."""
        f = C.c_int32()
        check(self.lib.ldp_poll_fault(self._h, C.byref(f)))
        if f.value:
            self.fault_upto = self.call_seq
            self.fault_kinds |= int(f.value)
            self.last_fault_kinds = int(f.value)       # of the poll that marked the calls up to fault_upto suspect (the warning names THIS fault)
        return int(f.value)

    def poll_fault(self) -> bool:
        """True if any fault was recorded since the last poll: results enqueued since then must be recomputed."""
        return self.poll_fault_kinds() != 0

    def vae_encode_checked(self, img_nhwc: torch.Tensor) -> torch.Tensor:
        """vae_encode for callers that keep no CallRecord (bulk pre-encoding): synchronises, and when the range guard
        of the fp16-plane convolutions fired (an activation beyond 65504: the result holds inf / NaN) encodes again --
        the handle has switched to the bf16 planes by then."""
        for _ in range(3):
            out = self.vae_encode(img_nhwc)
            torch.cuda.current_stream(self.device).synchronize()
            if not self.poll_fault_kinds():
                return out
        raise RuntimeError("libldp_hip: vae_encode faulted three times in a row")

    # -- planner --------------------------------------------------------------------------------
    def unet_forward(self, x: torch.Tensor, k, cond: Optional[torch.Tensor]) -> torch.Tensor:
        x = _f32(x, self.device)
        B = x.shape[0]
        cond_t = None if cond is None else _f32(cond, self.device)
        _want("x", x, (B, self.T, self.D))
        _want("cond", cond_t, (B, self.G))
        eps = torch.empty_like(x)
        if torch.is_tensor(k) or isinstance(k, np.ndarray):
            kd = torch.as_tensor(k).to(device=self.device, dtype=torch.int32).reshape(-1)
            if kd.numel() == 1:
                kd = kd.expand(B)
            kd = kd.contiguous()
            check(self.lib.ldp_unet_forward(self._h, _ptr(x), _ptr(kd), 0, _ptr(cond_t), _ptr(eps), B,
                                            self._stream()))
        else:
            check(self.lib.ldp_unet_forward(self._h, _ptr(x), None, int(k), _ptr(cond_t), _ptr(eps), B,
                                            self._stream()))
        self.call_seq += 1              # (a single evaluation can fault like a loop: column split, fp16-plane range)
        return eps

    def plan_sample(self, cond: Optional[torch.Tensor], B: Optional[int] = None,
                    x_init: Optional[torch.Tensor] = None, step_noise: Optional[torch.Tensor] = None,
                    seed: int = 0, row_offset: int = 0, sampler: str = "ddpm",
                    n_steps: Optional[int] = None, use_graph: bool = True) -> torch.Tensor:
        cond_t = None if cond is None else _f32(cond, self.device)
        if B is None:
            B = cond_t.shape[0] if cond_t is not None else x_init.shape[0]
        n_steps = self.planner_train_steps if n_steps is None else int(n_steps)
        xi = None if x_init is None else _f32(x_init, self.device)
        nz = None if step_noise is None else _f32(step_noise, self.device)
        _want("cond", cond_t, (B, self.G))
        _want("x_init", xi, (B, self.T, self.D))
        _want("step_noise", nz, (n_steps, B, self.T, self.D))
        out = torch.empty((B, self.T, self.D), device=self.device, dtype=torch.float32)
        check(self.lib.ldp_plan_sample(self._h, _ptr(cond_t), _ptr(xi), _ptr(nz), C.c_uint64(seed & (2**64 - 1)),
                                       C.c_int64(row_offset), _SAMPLERS[sampler], n_steps, _ptr(out), B,
                                       1 if use_graph else 0, self._stream()))
        self.call_seq += 1
        return out

    # -- IDM ------------------------------------------------------------------------------------
    def idm_forward(self, s: torch.Tensor, a: torch.Tensor, k) -> torch.Tensor:
        s, a = _f32(s, self.device), _f32(a, self.device)
        R = s.shape[0]
        _want("s", s, (R, 2 * self.D))
        _want("a", a, (R, self.A))
        eps = torch.empty_like(a)
        if torch.is_tensor(k) or isinstance(k, np.ndarray):
            kd = torch.as_tensor(k).to(device=self.device, dtype=torch.int32).reshape(-1)
            if kd.numel() == 1:
                kd = kd.expand(R)
            kd = kd.contiguous()
            check(self.lib.ldp_idm_forward(self._h, _ptr(s), _ptr(a), _ptr(kd), 0, _ptr(eps), R, self._stream()))
        else:
            check(self.lib.ldp_idm_forward(self._h, _ptr(s), _ptr(a), None, int(k), _ptr(eps), R, self._stream()))
        self.call_seq += 1
        return eps

    def idm_sample(self, transition: torch.Tensor, a_init: Optional[torch.Tensor] = None,
                   step_noise: Optional[torch.Tensor] = None, seed: int = 0, row_offset: int = 0,
                   sampler: str = "ddpm", n_steps: Optional[int] = None, use_graph: bool = True) -> torch.Tensor:
        tr = _f32(transition, self.device)
        R = tr.shape[0]
        n_steps = self.idm_train_steps if n_steps is None else int(n_steps)
        ai = None if a_init is None else _f32(a_init, self.device)
        nz = None if step_noise is None else _f32(step_noise, self.device)
        _want("transition", tr, (R, 2 * self.D))
        _want("a_init", ai, (R, self.A))
        _want("step_noise", nz, (n_steps, R, self.A))
        out = torch.empty((R, self.A), device=self.device, dtype=torch.float32)
        check(self.lib.ldp_idm_sample(self._h, _ptr(tr), _ptr(ai), _ptr(nz), C.c_uint64(seed & (2**64 - 1)),
                                      C.c_int64(row_offset), _SAMPLERS[sampler], n_steps, _ptr(out), R,
                                      1 if use_graph else 0, self._stream()))
        self.call_seq += 1
        return out

    # -- planner + IDM as one call / one captured graph ---------------------------------------------
    def agent_sample(self, obs_emb: torch.Tensor, obs_horizon: int, *, x_init=None, x_noise=None, a_init=None,
                     a_noise=None, seed: int = 0, row_offset: int = 0, sampler: str = "ddpm",
                     planner_steps: Optional[int] = None, idm_steps: Optional[int] = None,
                     action_bounds=None, action_mode: int = 0, use_graph: bool = True):
        """sample_viz_step without the decode (agent/ldp_agent.py:452-505): -> (x, plan, action) with
        x (B,T,D), plan (B,ah+1,D), action (B,ah,A).  action_bounds = (lo, hi) device tensors of length
        1 or A (None: actions stay normalised); action_mode 0 = unnormalize + clip, 2 = clip."""
        ob = _f32(obs_emb, self.device)
        B, H = ob.shape[0], ob.shape[1]
        _want("obs_emb", ob, (B, H, self.D))
        ps = self.planner_train_steps if planner_steps is None else int(planner_steps)
        is_ = self.idm_train_steps if idm_steps is None else int(idm_steps)
        R = B * self.ah
        xi = None if x_init is None else _f32(x_init, self.device)
        xn = None if x_noise is None else _f32(x_noise, self.device)
        ai = None if a_init is None else _f32(a_init, self.device)
        an = None if a_noise is None else _f32(a_noise, self.device)
        _want("x_init", xi, (B, self.T, self.D))
        _want("x_noise", xn, (ps, B, self.T, self.D))
        _want("a_init", ai, (R, self.A))
        _want("a_noise", an, (is_, R, self.A))
        lo = hi = None
        adim = 0
        if action_bounds is not None:
            lo, hi = self._bounds(*action_bounds)
            adim = lo.numel()
            if adim not in (1, self.A) or hi.numel() != adim:
                raise ValueError(f"action bounds must have length 1 or {self.A}")
        x = torch.empty((B, self.T, self.D), device=self.device, dtype=torch.float32)
        plan = torch.empty((B, self.ah + 1, self.D), device=self.device, dtype=torch.float32)
        act = torch.empty((B, self.ah, self.A), device=self.device, dtype=torch.float32)
        check(self.lib.ldp_agent_sample(self._h, _ptr(ob), H, int(obs_horizon), _ptr(xi), _ptr(xn), _ptr(ai), _ptr(an),
                                        C.c_uint64(seed & (2**64 - 1)), C.c_int64(row_offset), _SAMPLERS[sampler], ps, is_,
                                        _ptr(x), _ptr(plan), _ptr(act), _ptr(lo), _ptr(hi), adim, int(action_mode), B,
                                        1 if use_graph else 0, self._stream()))
        self.call_seq += 1
        return x, plan, act

    # -- VAE ------------------------------------------------------------------------------------
    def vae_encode(self, img_nhwc: torch.Tensor) -> torch.Tensor:
        img = _f32(img_nhwc, self.device)
        n, s = img.shape[0], img.shape[1]
        _want("img_nhwc", img, (n, self.image_size, self.image_size, 3))
        out = torch.empty((n, s // 32, s // 32, self.cfg.vae_latent_channels), device=self.device,
                          dtype=torch.float32)
        check(self.lib.ldp_vae_encode(self._h, _ptr(img), _ptr(out), n, self._stream()))
        self.call_seq += 1              # (the fp16-plane range guard can fault this call)
        return out

    def vae_decode(self, z_nhwc: torch.Tensor) -> torch.Tensor:
        z = _f32(z_nhwc, self.device)
        n = z.shape[0]
        s = int(self.cfg.image_size)
        _want("z_nhwc", z, (n, s // 32, s // 32, self.latent_channels))
        out = torch.empty((n, 3, s, s), device=self.device, dtype=torch.float32)
        check(self.lib.ldp_vae_decode(self._h, _ptr(z), _ptr(out), n, self._stream()))
        self.call_seq += 1
        return out

    # -- elementwise ----------------------------------------------------------------------------
    def normalize_bounds(self, x: torch.Tensor, lo, hi, normalize) -> torch.Tensor:
        """normalize: True/1 -> to [-1,1]; False/0 -> back (+clip); 2 -> plain clip to [lo, hi]."""
        x = _f32(x, self.device)
        lo_t, hi_t = self._bounds(lo, hi)
        dim = lo_t.numel()
        if dim != 1 and x.shape[-1] != dim:
            raise ValueError(f"bounds of length {dim} do not match trailing axis {x.shape[-1]}")
        y = torch.empty_like(x)
        check(self.lib.ldp_normalize_bounds(_ptr(x), _ptr(y), x.numel(), _ptr(lo_t), _ptr(hi_t), dim,
                                            int(normalize), self._stream()))
        return y

    def mean_sq_diff(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """mean((a - b)^2) as a device scalar (the `plan_mse` of agent/ldp_agent.py:497-499)."""
        a, b = _f32(a, self.device), _f32(b, self.device)
        if a.shape != b.shape:
            raise ValueError(f"shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty((), dtype=torch.float32, device=self.device)
        check(self.lib.ldp_mean_sq_diff(_ptr(a), _ptr(b), a.numel(), _ptr(out), self._stream()))
        return out

    def add_noise(self, x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor, n_train: int) -> torch.Tensor:
        """FlaxDDPMScheduler.add_noise (agent/ldp_agent.py:119,136): x0 / noise (rows, ...), t (rows,) int."""
        x0, noise = _f32(x0, self.device), _f32(noise, self.device)
        rows = x0.shape[0]
        td = torch.as_tensor(t).to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        _want("noise", noise, x0.shape)
        _want("t", td, (rows,))
        out = torch.empty_like(x0)
        check(self.lib.ldp_add_noise(_ptr(x0), _ptr(noise), _ptr(td), int(n_train), _ptr(out), rows, x0.numel() // rows,
                                     self._stream()))
        return out

    def reduce_stats(self, x: torch.Tensor) -> torch.Tensor:
        """(min, max, mean, population std) of x as a device tensor of 4 (agent/ldp_agent.py:163-178)."""
        x = _f32(x, self.device)
        out = torch.empty((4,), dtype=torch.float32, device=self.device)
        check(self.lib.ldp_reduce_stats(_ptr(x), x.numel(), _ptr(out), self._stream()))
        return out

    def _bounds(self, lo, hi):
        """Device copies of normalisation bounds, cached by value: a policy call normalises 4-5 keys and would
        otherwise pay two small (synchronous, pageable) host-to-device copies for each."""
        if torch.is_tensor(lo) and torch.is_tensor(hi):
            return _f32(lo, self.device).reshape(-1), _f32(hi, self.device).reshape(-1)
        lo_a = np.atleast_1d(np.asarray(lo, dtype=np.float32))
        hi_a = np.atleast_1d(np.asarray(hi, dtype=np.float32))
        key = (lo_a.tobytes(), hi_a.tobytes())
        hit = self._bounds_cache.get(key)
        if hit is None:
            if len(self._bounds_cache) > 256:
                self._bounds_cache.clear()
            hit = (_f32(lo_a, self.device), _f32(hi_a, self.device))
            self._bounds_cache[key] = hit
        return hit

    # -- training step (include/ldp_hip.h "training step"; agent/ldp_agent.py:113-180, 223-323) -----------------------
    _MODS = {"planner": MOD_PLANNER, "idm": MOD_IDM}

    def _mask(self, modules) -> int:
        if isinstance(modules, str):
            modules = [modules]
        m = 0
        for name in modules:
            m |= self._MODS[name]
        return m

    def train_init(self, modules) -> None:
        """TrainState.create for the listed modules from the parameters last uploaded (load_params): moments zero, step 0."""
        with torch.cuda.device(self.device):
            check(self.lib.ldp_train_init(self._h, self._mask(modules), self._stream()))

    def train_load(self, module: str, params: Dict[str, np.ndarray], mu=None, nu=None, step: int = 0, token=None) -> None:
        """TrainState.create(params) -- or a restored TrainState (mu / nu / step from a checkpoint) -- for one module.  The parameters go through
        ldp_set_weight, so the SAMPLING side of that module is un-finalized afterwards (train_publish / load_params rebuild it)."""
        for path, arr in params.items():
            a = np.ascontiguousarray(np.asarray(arr), dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(self.lib.ldp_set_weight(self._h, f"{module}/{path}".encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        self.loaded[module] = None
        self.train_init([module])
        if mu is not None:
            self.train_write(module, self.TRAIN_MU, mu)
        if nu is not None:
            self.train_write(module, self.TRAIN_NU, nu)
        if step:
            self.train_step_count(module, set_to=step)
        self.train_token[module] = token if token is not None else object()

    def _timesteps(self, t, n_train: int, what: str) -> torch.Tensor:
        """int32 device vector of training timesteps; the range is checked where the values live (host values: no device round trip)."""
        tt = torch.as_tensor(t)
        if tt.numel() == 0 or int(tt.min()) < 0 or int(tt.max()) >= n_train:
            raise ValueError(f"{what} timesteps must lie in [0, {n_train})")
        return tt.to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()

    def aux_streams(self) -> Dict[str, "torch.cuda.Stream"]:
        """Two more streams of this engine's device for the training step: the IDM's tape next to the planner's, the statistics scalars next to both."""
        if getattr(self, "_aux_streams", None) is None:
            self._aux_streams = {"idm": torch.cuda.Stream(device=self.device), "stats": torch.cuda.Stream(device=self.device)}
        return self._aux_streams

    def train_planner_grad(self, x0: torch.Tensor, noise: torch.Tensor, t, cond: Optional[torch.Tensor], alpha: float = 1.0) -> torch.Tensor:
        """alpha * plan_loss and its gradients (agent/ldp_agent.py:113-127): x0 / noise (B, T, D), t (B,), cond (B, G) -> device scalar."""
        x0, noise = _f32(x0, self.device), _f32(noise, self.device)
        B = x0.shape[0]
        td = self._timesteps(t, self.planner_train_steps, "planner")
        cond_t = None if cond is None else _f32(cond, self.device)
        _want("x0", x0, (B, self.T, self.D)); _want("noise", noise, (B, self.T, self.D)); _want("t", td, (B,)); _want("cond", cond_t, (B, self.G))
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        check(self.lib.ldp_train_planner_grad(self._h, _ptr(x0), _ptr(noise), _ptr(td), _ptr(cond_t), C.c_float(alpha), _ptr(loss), B, self._stream()))
        self._keep = (x0, noise, td, cond_t)           # the launches are asynchronous: the inputs must outlive them
        return loss

    def train_idm_grad(self, s: torch.Tensor, a0: torch.Tensor, noise: torch.Tensor, t, alpha: float = 1.0) -> torch.Tensor:
        """alpha * idm_loss and its gradients (agent/ldp_agent.py:129-140): s (R, 2D), a0 / noise (R, A), t (R,) -> device scalar."""
        s, a0, noise = _f32(s, self.device), _f32(a0, self.device), _f32(noise, self.device)
        R = s.shape[0]
        td = self._timesteps(t, self.idm_train_steps, "IDM")
        _want("s", s, (R, 2 * self.D)); _want("a0", a0, (R, self.A)); _want("noise", noise, (R, self.A)); _want("t", td, (R,))
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        check(self.lib.ldp_train_idm_grad(self._h, _ptr(s), _ptr(a0), _ptr(noise), _ptr(td), C.c_float(alpha), _ptr(loss), R, self._stream()))
        self._keep_i = (s, a0, noise, td)
        return loss

    def train_grad_norm(self, modules) -> torch.Tensor:
        out = torch.empty((), dtype=torch.float32, device=self.device)
        check(self.lib.ldp_train_grad_norm(self._h, self._mask(modules), _ptr(out), self._stream()))
        return out

    def train_apply(self, module: str, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
        check(self.lib.ldp_train_apply(self._h, self._MODS[module], C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), self._stream()))

    def train_step_count(self, module: str, set_to: Optional[int] = None) -> int:
        v = C.c_int64()
        check(self.lib.ldp_train_step_count(self._h, self._MODS[module], C.c_int64(-1 if set_to is None else int(set_to)), C.byref(v)))
        return int(v.value)

    TRAIN_PARAMS, TRAIN_GRADS, TRAIN_MU, TRAIN_NU = 0, 1, 2, 3

    def train_read(self, module: str, which: int, shapes) -> Dict[str, np.ndarray]:
        """{flax path: array} of the module's parameters / gradients / Adam moments; `shapes` = {path: shape} (weights.planner_shapes / idm_shapes)."""
        out = {}
        for path, shape in shapes.items():
            a = np.empty(tuple(shape), dtype=np.float32)
            check(self.lib.ldp_train_read(self._h, self._MODS[module], int(which), path.encode(), a.ctypes.data_as(C.c_void_p), a.size, self._stream()))
            out[path] = a
        return out

    def train_write(self, module: str, which: int, tree) -> None:
        for path, arr in tree.items():
            a = np.ascontiguousarray(np.asarray(arr), dtype=np.float32)
            check(self.lib.ldp_train_write(self._h, self._MODS[module], int(which), path.encode(), a.ctypes.data_as(C.c_void_p), a.size, self._stream()))

    def train_arena(self, module: str, which: int) -> torch.Tensor:
        """The module's flat parameter / gradient / moment arena as a 1-D float32 device tensor that ALIASES the engine's memory (no copy):
        what dist.update_sharded hands to the gradient all-reduce.  Valid until the next train_init."""
        ptr, n = C.c_void_p(), C.c_int64()
        check(self.lib.ldp_train_arena(self._h, self._MODS[module], int(which), C.byref(ptr), C.byref(n)))
        return _alias_device_f32(ptr.value, n.value, self.device)

    def train_publish(self, modules, versions: Optional[dict] = None) -> None:
        """The sampling path takes over the trained parameters (packed layouts and tables are rebuilt)."""
        with torch.cuda.device(self.device):
            check(self.lib.ldp_train_publish(self._h, self._mask(modules), self._stream()))
        for name in ([modules] if isinstance(modules, str) else modules):
            self.loaded[name] = (versions or {}).get(name, object())

    def check_fault(self) -> None:
        """Synchronises the current stream and raises LDPHipFault if a fault (exchange time-out or fp16-plane range,
        poll_fault_kinds) was recorded since the last check / poll."""
        check(self.lib.ldp_check_fault(self._h, self._stream()))

    def launch_counts(self):
        n_conv, n_all = C.c_int64(), C.c_int64()
        check(self.lib.ldp_launch_count(self._h, 0, C.byref(n_conv)))
        check(self.lib.ldp_launch_count(self._h, 1, C.byref(n_all)))
        return n_conv.value, n_all.value


# ---- unit-testable primitives (no handle) -------------------------------------------------------
def _host(a):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def conv1d_gn_mish_film(x: torch.Tensor, kernel, bias, gn_scale, gn_bias,
                        film: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().float()
    B, T, cin = x.shape
    k, kp = _host(kernel)
    b, bp = _host(bias)
    gs, gsp = _host(gn_scale)
    gb, gbp = _host(gn_bias)
    cout = k.shape[2]
    y = torch.empty((B, T, cout), device=x.device, dtype=torch.float32)
    f = None if film is None else film.contiguous().float()
    check(lib.ldp_conv1d_gn_mish_film_f32(_ptr(x), kp, bp, gsp, gbp, _ptr(f), _ptr(y), B, T, cin, cout,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


def conv2d_3x3(x: torch.Tensor, kernel, bias, stride: int = 1) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().float()
    n, h, w, cin = x.shape
    k, kp = _host(kernel)
    b, bp = _host(bias)
    cout = k.shape[3]
    y = torch.empty((n, h // stride, w // stride, cout), device=x.device, dtype=torch.float32)
    check(lib.ldp_conv2d_3x3_f32(_ptr(x), kp, bp, _ptr(y), n, h, w, cin, cout, stride,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


def conv2d_3x3_split(x: torch.Tensor, kernel, bias, res: Optional[torch.Tensor] = None, dual: bool = True,
                     with_stats: bool = False):
    """The stride-1 3x3 convolution on split bf16 operands (ldp_conv2d_3x3_bf16x3: what the StableVAE's 64 / 32 / 16
    pixel ResnetBlock2D convolutions run on).  dual = 2: on two fp16 planes / three products instead of three bf16 planes / six.
    -> y, or (y, per-256-pixel-tile column (sum, sum of squares))."""
    lib = _lib.load()
    x = x.contiguous().float()
    n, h, w, cin = x.shape
    k, kp = _host(kernel)
    b, bp = _host(bias)
    cout = k.shape[3]
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    st = torch.empty((n * h * w // 256, cout, 2), device=x.device, dtype=torch.float32) if with_stats else None
    if res is not None:
        res = res.contiguous().float()
    check(lib.ldp_conv2d_3x3_bf16x3(_ptr(x), kp, bp, _ptr(res) if res is not None else None, _ptr(y),
                                    _ptr(st) if st is not None else None, n, h, w, cin, cout, 2 if dual == 2 else 1 if dual else 0,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return (y, st) if with_stats else y


def downsample1d(x: torch.Tensor, kernel, bias) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().float()
    B, T, c = x.shape
    k, kp = _host(kernel)
    b, bp = _host(bias)
    y = torch.empty((B, T // 2, c), device=x.device, dtype=torch.float32)
    check(lib.ldp_downsample1d_f32(_ptr(x), kp, bp, _ptr(y), B, T, c,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


def upsample1d(x: torch.Tensor, kernel, bias) -> torch.Tensor:
    lib = _lib.load()
    x = x.contiguous().float()
    B, T, c = x.shape
    k, kp = _host(kernel)
    b, bp = _host(bias)
    y = torch.empty((B, 2 * T, c), device=x.device, dtype=torch.float32)
    check(lib.ldp_upsample1d_f32(_ptr(x), kp, bp, _ptr(y), B, T, c,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


class _DevSpan:
    """__cuda_array_interface__ (v2) over a span of device memory somebody else owns."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


def _alias_device_f32(ptr: int, numel: int, device) -> torch.Tensor:
    with torch.cuda.device(device):
        t = torch.as_tensor(_DevSpan(ptr, numel), device=torch.device("cuda", torch.cuda.current_device()))
    if t.data_ptr() != ptr:
        raise LDPHipError(-2, "torch copied the arena instead of aliasing it")
    return t


# ---- noise-source primitives (tests/test_philox.py) -------------------------------------------------
def philox_raw(seed: int, elem0: int, step: int, stream_id: int, n: int, device=None) -> torch.Tensor:
    """(n, 4) uint32 words of Philox4x32-10 at counters (elem0 + i, step, stream_id), key = seed."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty((n, 4), device=dev, dtype=torch.int32)
    check(lib.ldp_philox_raw(C.c_uint64(seed & (2**64 - 1)), C.c_uint64(elem0 & (2**64 - 1)), C.c_uint32(step),
                             C.c_uint32(stream_id), _ptr(out), n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def philox_normal(seed: int, elem0: int, step: int, stream_id: int, n: int, device=None) -> torch.Tensor:
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.empty((n,), device=dev, dtype=torch.float32)
    check(lib.ldp_philox_normal(C.c_uint64(seed & (2**64 - 1)), C.c_uint64(elem0 & (2**64 - 1)), C.c_uint32(step),
                                C.c_uint32(stream_id), _ptr(out), n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
