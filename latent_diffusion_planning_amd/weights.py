"""Parameter trees of the LDP denoising hot path (planner U-Net, IDM MLP, StableVAE).

Host-side logic only (numpy): names, shapes and the synthetic initialiser used by
`LDPAgent.create`, `bench.py` and the tests.  Every tree is a *flat* dict
``{"<flax path>/<leaf>": float32 ndarray}`` whose keys follow the names Flax / diffusers
auto-assign to the reference modules, so that a checkpoint exported elsewhere with
``flax.traverse_util.flatten_dict(params, sep="/")`` loads unchanged.

Reference anchors (behavioural spec, nothing is copied):
  * planner tree   : networks/diffusion_nets_v2.py:66-169   
  * IDM tree       : networks/mlp_diffusion_nets.py:8-68, networks/mlp_nets.py:49-97 (B.2)
  * VAE tree       : model/stable_vae_model.yaml:4-16 -> diffusers 0.27.2 FlaxAutoencoderKL (A.3)
  * init families  : agent/ldp_agent.py:566-614 (`create`), Flax defaults

The Flax RNG stream (threefry) is *not* reproduced: initial values come from NumPy PCG64,
drawn from the same distribution families.  Biases and norm affine parameters get a small
non-zero perturbation so that a kernel which forgets one of them fails its parity test.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

Params = Dict[str, np.ndarray]


# --------------------------------------------------------------------------------------
# architecture descriptions
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class PlannerSpec:
    """ConditionalUnet1D hyper-parameters (agent/ldp_agent.yaml:7-15)."""
    input_dim: int                      # D  (obs_dim)
    global_cond_dim: int                # obs_horizon * D  (width actually fed, ldp_agent.py:573-575)
    diffusion_step_embed_dim: int = 256
    down_dims: Tuple[int, ...] = (256, 512, 1024)
    kernel_size: int = 5
    n_groups: int = 8
    downsample: bool = True

    @property
    def cond_dim(self) -> int:
        return self.diffusion_step_embed_dim + self.global_cond_dim

    def blocks(self) -> List[Tuple[int, int, bool]]:
        """(Cin, Cout, residual_proj) of ConditionalResidualBlock1D_0..N in construction order."""
        out: List[Tuple[int, int, bool]] = []
        cin = self.input_dim
        for c in self.down_dims:                       # down path
            out.append((cin, c, True))
            out.append((c, c, False))
            cin = c
        mid = self.down_dims[-1]
        out.append((mid, mid, False))                  # mid
        out.append((mid, mid, False))
        for c in reversed(self.down_dims[:-1]):        # up path: input is concat(x, skip)
            out.append((2 * cin, c, True))             # skip h.pop() has as many channels as x
            out.append((c, c, False))
            cin = c
        return out

    def n_levels(self) -> int:
        return len(self.down_dims)


@dataclass(frozen=True)
class IDMSpec:
    """MLPDiffusion hyper-parameters (agent/ldp_agent.yaml:17-34)."""
    obs_dim: int                        # D
    action_dim: int                     # A
    time_dim: int = 256                 # FourierFeatures.output_size
    cond_hidden: Tuple[int, ...] = (256, 256)
    hidden_dim: int = 256
    n_blocks: int = 3

    @property
    def in_dim(self) -> int:
        return self.action_dim + 2 * self.obs_dim + self.cond_hidden[-1]


@dataclass(frozen=True)
class VAESpec:
    """FlaxAutoencoderKL configuration (model/stable_vae_model.yaml:4-16)."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 256, 256, 256, 256)
    layers_per_block: int = 2
    norm_num_groups: int = 32


# --------------------------------------------------------------------------------------
# shape tables
# --------------------------------------------------------------------------------------
def planner_shapes(spec: PlannerSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    e = spec.diffusion_step_embed_dim
    k = spec.kernel_size
    s["Dense_0/kernel"] = (e, 4 * e)
    s["Dense_0/bias"] = (4 * e,)
    s["Dense_1/kernel"] = (4 * e, e)
    s["Dense_1/bias"] = (e,)
    for i, (cin, cout, proj) in enumerate(spec.blocks()):
        p = f"ConditionalResidualBlock1D_{i}"
        for j, ci in enumerate((cin, cout)):
            s[f"{p}/Conv1dBlock_{j}/Conv_0/kernel"] = (k, ci, cout)
            s[f"{p}/Conv1dBlock_{j}/Conv_0/bias"] = (cout,)
            s[f"{p}/Conv1dBlock_{j}/GroupNorm_0/scale"] = (cout,)
            s[f"{p}/Conv1dBlock_{j}/GroupNorm_0/bias"] = (cout,)
        s[f"{p}/Dense_0/kernel"] = (spec.cond_dim, 2 * cout)
        s[f"{p}/Dense_0/bias"] = (2 * cout,)
        if proj:
            s[f"{p}/Conv_0/kernel"] = (1, cin, cout)
            s[f"{p}/Conv_0/bias"] = (cout,)
    if spec.downsample:
        for i, c in enumerate(spec.down_dims[:-1]):
            s[f"Downsample1d_{i}/Conv_0/kernel"] = (3, c, c)
            s[f"Downsample1d_{i}/Conv_0/bias"] = (c,)
        for i, c in enumerate(reversed(spec.down_dims[:-1])):
            s[f"Upsample1d_{i}/ConvTranspose_0/kernel"] = (4, c, c)
            s[f"Upsample1d_{i}/ConvTranspose_0/bias"] = (c,)
    c0 = spec.down_dims[0]
    s["Conv1dBlock_0/Conv_0/kernel"] = (k, c0, c0)
    s["Conv1dBlock_0/Conv_0/bias"] = (c0,)
    s["Conv1dBlock_0/GroupNorm_0/scale"] = (c0,)
    s["Conv1dBlock_0/GroupNorm_0/bias"] = (c0,)
    s["Conv_0/kernel"] = (1, c0, spec.input_dim)
    s["Conv_0/bias"] = (spec.input_dim,)
    return s


def idm_shapes(spec: IDMSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    w = spec.time_dim
    for i, h in enumerate(spec.cond_hidden):
        s[f"MLP_0/Dense_{i}/kernel"] = (w, h)
        s[f"MLP_0/Dense_{i}/bias"] = (h,)
        w = h
    hd = spec.hidden_dim
    s["MLPResNet_0/Dense_0/kernel"] = (spec.in_dim, hd)
    s["MLPResNet_0/Dense_0/bias"] = (hd,)
    for i in range(spec.n_blocks):
        p = f"MLPResNet_0/MLPResNetBlock_{i}"
        s[f"{p}/LayerNorm_0/scale"] = (hd,)
        s[f"{p}/LayerNorm_0/bias"] = (hd,)
        s[f"{p}/Dense_0/kernel"] = (hd, 4 * hd)
        s[f"{p}/Dense_0/bias"] = (4 * hd,)
        s[f"{p}/Dense_1/kernel"] = (4 * hd, hd)
        s[f"{p}/Dense_1/bias"] = (hd,)
    s["MLPResNet_0/Dense_1/kernel"] = (hd, spec.action_dim)
    s["MLPResNet_0/Dense_1/bias"] = (spec.action_dim,)
    return s


def _resnet2d(s, p, cin, cout):
    s[f"{p}/norm1/scale"] = (cin,)
    s[f"{p}/norm1/bias"] = (cin,)
    s[f"{p}/conv1/kernel"] = (3, 3, cin, cout)
    s[f"{p}/conv1/bias"] = (cout,)
    s[f"{p}/norm2/scale"] = (cout,)
    s[f"{p}/norm2/bias"] = (cout,)
    s[f"{p}/conv2/kernel"] = (3, 3, cout, cout)
    s[f"{p}/conv2/bias"] = (cout,)
    if cin != cout:
        s[f"{p}/conv_shortcut/kernel"] = (1, 1, cin, cout)
        s[f"{p}/conv_shortcut/bias"] = (cout,)


def _mid_block(s, p, c):
    _resnet2d(s, f"{p}/resnets_0", c, c)
    a = f"{p}/attentions_0"
    s[f"{a}/group_norm/scale"] = (c,)
    s[f"{a}/group_norm/bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{a}/{n}/kernel"] = (c, c)
        s[f"{a}/{n}/bias"] = (c,)
    _resnet2d(s, f"{p}/resnets_1", c, c)


def vae_encoder_shapes(spec: VAESpec) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = spec.block_out_channels
    s["encoder/conv_in/kernel"] = (3, 3, spec.in_channels, ch[0])
    s["encoder/conv_in/bias"] = (ch[0],)
    cin = ch[0]
    for i, c in enumerate(ch):
        for j in range(spec.layers_per_block):
            _resnet2d(s, f"encoder/down_blocks_{i}/resnets_{j}", cin, c)
            cin = c
        if i != len(ch) - 1:
            s[f"encoder/down_blocks_{i}/downsamplers_0/conv/kernel"] = (3, 3, c, c)
            s[f"encoder/down_blocks_{i}/downsamplers_0/conv/bias"] = (c,)
    _mid_block(s, "encoder/mid_block", ch[-1])
    s["encoder/conv_norm_out/scale"] = (ch[-1],)
    s["encoder/conv_norm_out/bias"] = (ch[-1],)
    s["encoder/conv_out/kernel"] = (3, 3, ch[-1], 2 * spec.latent_channels)
    s["encoder/conv_out/bias"] = (2 * spec.latent_channels,)
    s["quant_conv/kernel"] = (1, 1, 2 * spec.latent_channels, 2 * spec.latent_channels)
    s["quant_conv/bias"] = (2 * spec.latent_channels,)
    return s


def vae_decoder_shapes(spec: VAESpec) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    lc = spec.latent_channels
    rev = tuple(reversed(spec.block_out_channels))
    s["post_quant_conv/kernel"] = (1, 1, lc, lc)
    s["post_quant_conv/bias"] = (lc,)
    s["decoder/conv_in/kernel"] = (3, 3, lc, rev[0])
    s["decoder/conv_in/bias"] = (rev[0],)
    _mid_block(s, "decoder/mid_block", rev[0])
    cin = rev[0]
    for i, c in enumerate(rev):
        for j in range(spec.layers_per_block + 1):
            _resnet2d(s, f"decoder/up_blocks_{i}/resnets_{j}", cin, c)
            cin = c
        if i != len(rev) - 1:
            s[f"decoder/up_blocks_{i}/upsamplers_0/conv/kernel"] = (3, 3, c, c)
            s[f"decoder/up_blocks_{i}/upsamplers_0/conv/bias"] = (c,)
    s["decoder/conv_norm_out/scale"] = (rev[-1],)
    s["decoder/conv_norm_out/bias"] = (rev[-1],)
    s["decoder/conv_out/kernel"] = (3, 3, rev[-1], spec.out_channels)
    s["decoder/conv_out/bias"] = (spec.out_channels,)
    return s


def vae_shapes(spec: VAESpec, decoder: bool = True):
    s = vae_encoder_shapes(spec)
    if decoder:
        s.update(vae_decoder_shapes(spec))
    return s


# --------------------------------------------------------------------------------------
# synthetic initialisation (distribution families of the Flax defaults)
# --------------------------------------------------------------------------------------
# Dense layers the reference builds with kernel_init=xavier_uniform;
# every other kernel uses Flax's default lecun_normal.
def _is_xavier(path: str) -> bool:
    if path.startswith("Dense_0/") or path.startswith("Dense_1/"):          # planner time MLP
        return True
    if "ConditionalResidualBlock1D_" in path and "/Dense_0/" in path:       # FiLM
        return True
    if path.startswith("MLP_0/"):                                            # IDM cond encoder
        return True
    if path in ("MLPResNet_0/Dense_0/kernel", "MLPResNet_0/Dense_1/kernel"):  # IDM in / out
        return True
    return False


def _fans(shape: Sequence[int]) -> Tuple[int, int]:
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return rf * shape[-2], rf * shape[-1]


def init_from_shapes(shapes, seed: int, perturb: bool = True) -> Params:
    """Draw a parameter tree.  `perturb=False` gives exact Flax defaults for bias/affine
    (zeros / ones); the default adds small noise so omissions are visible in parity tests."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Params = OrderedDict()
    for path, shape in shapes.items():
        leaf = path.rsplit("/", 1)[1]
        if leaf == "kernel":
            fan_in, fan_out = _fans(shape)
            if _is_xavier(path):
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                w = rng.uniform(-lim, lim, size=shape)
            else:
                # lecun_normal: truncated normal (+-2 sigma) with variance 1/fan_in
                std = np.sqrt(1.0 / fan_in) / 0.87962566103423978
                w = np.clip(rng.standard_normal(size=shape), -2.0, 2.0) * std
        elif leaf == "scale":
            w = np.ones(shape) + (0.1 * rng.standard_normal(size=shape) if perturb else 0.0)
        elif leaf == "bias":
            w = 0.02 * rng.standard_normal(size=shape) if perturb else np.zeros(shape)
        else:
            raise KeyError(f"unknown leaf kind in {path}")
        out[path] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def init_planner_params(spec: PlannerSpec, seed: int = 0, perturb: bool = True) -> Params:
    return init_from_shapes(planner_shapes(spec), seed, perturb)


def init_idm_params(spec: IDMSpec, seed: int = 1, perturb: bool = True) -> Params:
    return init_from_shapes(idm_shapes(spec), seed, perturb)


def init_vae_params(spec: VAESpec = VAESpec(), seed: int = 2, perturb: bool = True,
                    decoder: bool = True) -> Params:
    return init_from_shapes(vae_shapes(spec, decoder), seed, perturb)


def check_params(params: Params, shapes) -> None:
    """Raise with a precise message on a missing / extra / mis-shaped leaf."""
    missing = [k for k in shapes if k not in params]
    extra = [k for k in params if k not in shapes]
    if missing or extra:
        raise KeyError(f"parameter tree mismatch: missing={missing[:5]} extra={extra[:5]}")
    for k, shp in shapes.items():
        if tuple(params[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected shape {tuple(shp)}, got {tuple(params[k].shape)}")


def count(params: Params) -> int:
    return int(sum(int(v.size) for v in params.values()))


# --------------------------------------------------------------------------------------
# flat <-> nested, npz I/O  (weight import)
# --------------------------------------------------------------------------------------
def unflatten(params: Params) -> dict:
    root: dict = {}
    for k, v in params.items():
        node = root
        parts = k.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return root


def flatten(tree: dict, prefix: str = "") -> Params:
    out: Params = OrderedDict()
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, key))
        else:
            out[key] = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
    return out


def save_npz(path: str, **trees: Params) -> None:
    flat = {}
    for name, tree in trees.items():
        for k, v in tree.items():
            flat[f"{name}:{k}"] = v
    np.savez(path, **flat)


def save_safetensors(path: str, **trees: Params) -> None:
    """Flat `<tree>:<flax path>` keys, same naming as save_npz (safetensors keeps insertion order
    only through the metadata, so the order is stored explicitly)."""
    from safetensors.numpy import save_file
    flat, order = {}, []
    for name, tree in trees.items():
        for k, v in tree.items():
            flat[f"{name}:{k}"] = np.ascontiguousarray(v, dtype=np.float32)
            order.append(f"{name}:{k}")
    save_file(flat, path, metadata={"order": "\n".join(order)})


def load_safetensors(path: str) -> Dict[str, Params]:
    from safetensors import safe_open
    out: Dict[str, Params] = {}
    with safe_open(path, framework="np") as f:
        meta = f.metadata() or {}
        keys = meta.get("order", "").split("\n") if meta.get("order") else sorted(f.keys())
        for key in keys:
            name, k = key.split(":", 1)
            out.setdefault(name, OrderedDict())[k] = np.ascontiguousarray(f.get_tensor(key), dtype=np.float32)
    return out


def load_npz(path: str) -> Dict[str, Params]:
    out: Dict[str, Params] = {}
    with np.load(path) as z:
        for key in z.files:
            name, k = key.split(":", 1)
            out.setdefault(name, OrderedDict())[k] = np.ascontiguousarray(z[key], dtype=np.float32)
    return out
