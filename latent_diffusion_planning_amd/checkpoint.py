"""Reference checkpoints: the orbax `PyTreeCheckpointer` directories train_bc.py:203-208 writes and
train_bc.py:210-240 / agent/ldp_agent.py:543-551 / process_sdvae_data.py:40-47 restore.

`save_snapshot` passes `save_args = orbax_utils.save_args_from_target(ckpt)`, i.e. `SaveArgs(aggregate=True)` for every
leaf of a single-device tree (flax 0.8.4), so the whole pytree -- `planner_params`, `idm_params`, (`vae_params` for
train_vae.py), `data`, `cfg` -- lands in ONE msgpack file `<step>.ckpt/checkpoint` in the flax.serialization wire
format (orbax-checkpoint 0.5.14 `msgpack_utils`, a copy of flax/serialization.py):

    dict            -> msgpack map, str keys
    ndarray / jax   -> ExtType(1, packb((shape, dtype.name, C-order bytes), use_bin_type=True))
    numpy scalar    -> ExtType(3, same tuple)           complex -> ExtType(2, packb((re, im)))
    arrays > 2**30 B-> {"__msgpack_chunked_array__": True, "shape": ..., "chunks": {"0": ext, ...}} (row-major pieces)
    a leaf saved outside the aggregate file (aggregate=False: tensorstore) -> the string "PLACEHOLDER://<key path>"

orbax / flax / jax are not installed here; the format is read with the `msgpack` package alone (a flax dependency,
so it is wherever the reference runs).  Pinned by a hand-assembled byte string in tests/test_checkpoint.py, not by a
file orbax wrote -- none exists in this environment (f-4 stays "names unverified").
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Iterable, Optional

import numpy as np

from . import weights as W

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_CHUNKED = "__msgpack_chunked_array__"
PLACEHOLDER = "PLACEHOLDER://"


class CheckpointError(RuntimeError):
    pass


def _msgpack():
    try:
        import msgpack
    except ImportError as e:          # pragma: no cover
        raise CheckpointError("reading orbax aggregate checkpoints needs the `msgpack` package (a flax dependency)") from e
    return msgpack


def _ndarray_from(payload: bytes) -> np.ndarray:
    msgpack = _msgpack()
    shape, dtype_name, buf = msgpack.unpackb(payload, raw=False)
    if dtype_name == "bfloat16":
        u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
        return u.view(np.float32).reshape(shape)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()


def _ext_hook(code: int, data: bytes):
    if code == _EXT_NDARRAY:
        return _ndarray_from(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from(data)[()]
    if code == _EXT_COMPLEX:
        re, im = _msgpack().unpackb(data, raw=False)
        return complex(re, im)
    return _msgpack().ExtType(code, data)


def _unchunk(node):
    """flax.serialization._unchunk_array_leaves_in_place."""
    if isinstance(node, dict):
        if node.get(_CHUNKED):
            chunks = node["chunks"]
            flat = np.concatenate([np.asarray(chunks[str(i)]).reshape(-1) for i in range(len(chunks))])
            return flat.reshape(tuple(node["shape"]))
        return {k: _unchunk(v) for k, v in node.items()}
    return node


def aggregate_file(path: str) -> str:
    """`<step>.ckpt` directory (or the file itself) -> the aggregate msgpack file."""
    p = os.fspath(path)
    if os.path.isdir(p):
        f = os.path.join(p, "checkpoint")
        if not os.path.isfile(f):
            kids = sorted(os.listdir(p))[:8]
            if "_METADATA" in os.listdir(p) or "_sharding" in os.listdir(p) or any(k.startswith("ocdbt") or k == "manifest.ocdbt" for k in os.listdir(p)):
                # what newer orbax-checkpoint releases write (per-leaf tensorstore / OCDBT + a _METADATA tree description);
                # this reader was written against orbax-checkpoint 0.5.14 (env.yml) with aggregate=True: ONE msgpack file
                raise CheckpointError(f"{p}: this directory has orbax's _METADATA / OCDBT layout (found {kids}) and no aggregate file "
                                      "'checkpoint': it was not written with save_args aggregate=True as train_bc.py:203-208 does under "
                                      "orbax-checkpoint 0.5.14, or by a newer orbax that ignores `aggregate`.  Its arrays live in "
                                      "tensorstore; read it with orbax where it was written and export with weights.save_npz / "
                                      "save_safetensors")
            raise CheckpointError(f"{p}: no aggregate file 'checkpoint' in this directory (found {kids}); the reference writes "
                                  "one (train_bc.py:207 save_args_from_target -> aggregate=True).  A checkpoint whose arrays "
                                  "went to tensorstore needs orbax to read: export it with weights.save_npz there")
        return f
    if not os.path.isfile(p):
        raise CheckpointError(f"{p}: no such checkpoint")
    return p


def restore(path: str) -> dict:
    """`PyTreeCheckpointer().restore(path)` for aggregate checkpoints: the nested dict with NumPy leaves."""
    msgpack = _msgpack()
    with open(aggregate_file(path), "rb") as f:
        raw = f.read()
    try:
        tree = msgpack.unpackb(raw, ext_hook=_ext_hook, raw=False, strict_map_key=False)
    except Exception as e:
        raise CheckpointError(f"{path}: not a flax/orbax msgpack aggregate file ({type(e).__name__}: {e})") from e
    if not isinstance(tree, dict):
        raise CheckpointError(f"{path}: the aggregate file holds a {type(tree).__name__}, not a pytree dict")
    return _unchunk(tree)


def _placeholders(tree, prefix="") -> Iterable[str]:
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            yield from _placeholders(v, key)
        elif isinstance(v, str) and v.startswith(PLACEHOLDER):
            yield key


def param_trees(raw_restored: dict, keys: Optional[Iterable[str]] = None) -> Dict[str, "OrderedDict[str, np.ndarray]"]:
    """The `*_params` trees of a restored checkpoint as flat float32 `{flax path: array}` dicts (weights.flatten),
    `ema` trees skipped like train_bc.py:230-232.  `keys` = cfg.restore_keys (empty / None: everything)."""
    keys = set(keys or ())
    out: Dict[str, OrderedDict] = {}
    for k, v in raw_restored.items():
        if (keys and k not in keys) or not str(k).endswith("_params") or "ema" in str(k) or not isinstance(v, dict):
            continue
        held = list(_placeholders(v))
        if held:
            raise CheckpointError(f"{k}: {len(held)} leaves (first: {held[0]}) were saved outside the aggregate file "
                                  "(tensorstore); this reader has the aggregate file only")
        out[k] = W.flatten(v)
    return out


def load_snapshot(agent, path: str, restore_keys: Iterable[str] = ()):
    """train_bc.py:210-240 on this agent: every restored `<prefix>_params` replaces `<prefix>_state`'s params and
    ema_params (`vae_params` replaces the agent's VAE tree); returns the new agent (flax-style, the old one is untouched)."""
    trees = param_trees(restore(path), restore_keys)
    fields = {}
    for k, flat in trees.items():
        prefix = k[:-len("_params")]
        if prefix == "vae":
            fields["vae_params"] = flat
            continue
        state_name = f"{prefix}_state"
        state = getattr(agent, state_name, None)
        if state is None:
            if prefix == "encoder":          # LDP agents have no learned encoder state (train_bc.py:216-229 is the BC agents')
                continue
            raise CheckpointError(f"checkpoint holds {k} but the agent has no {state_name}")
        fields[state_name] = state.replace(params=flat, ema_params=flat)
    if not fields:
        raise CheckpointError(f"{path}: no *_params tree selected (keys in the file: {sorted(restore(path))})")
    return agent.replace(**fields)


# ---- writer (the same wire format: hands trees back to `PyTreeCheckpointer().restore`) ------------------------------
def _pack_tree(node):
    msgpack = _msgpack()
    if isinstance(node, dict):
        return {str(k): _pack_tree(v) for k, v in node.items()}
    if isinstance(node, (np.ndarray, np.generic)) or hasattr(node, "__array__"):
        a = np.asarray(node)
        code = _EXT_NPSCALAR if isinstance(node, np.generic) else _EXT_NDARRAY
        return msgpack.ExtType(code, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))
    if isinstance(node, (list, tuple)):
        return {str(i): _pack_tree(v) for i, v in enumerate(node)}      # flax state dicts: sequences become str-keyed maps
    return node


def save(path: str, tree: dict) -> str:
    """Write `tree` as `<path>/checkpoint`.  Values may be nested dicts or this package's flat `{"a/b/kernel": array}`
    parameter trees (nested first, so the file holds what `agent.get_params()` holds in the reference)."""
    msgpack = _msgpack()
    nested = {k: (W.unflatten(v) if isinstance(v, dict) and any("/" in str(q) for q in v) else v) for k, v in tree.items()}
    os.makedirs(path, exist_ok=True)
    f = os.path.join(path, "checkpoint")
    with open(f, "wb") as fh:
        fh.write(msgpack.packb(_pack_tree(nested), use_bin_type=True))
    return f


def save_snapshot(agent, path: str, batch: Optional[dict] = None, cfg: Optional[dict] = None) -> str:
    """train_bc.py:203-208 on this agent: `ckpt = dict(data=batch, cfg=cfg); ckpt.update(agent.get_params())` written as one aggregate file
    (`<path>/checkpoint`) that `load_snapshot` -- here or the reference's `PyTreeCheckpointer().restore` -- reads back.  For an agent that came out
    of `update()` this is where the trained parameters leave the GPU (the reference's snapshot holds parameters only: no optimiser state)."""
    ckpt = {}
    if batch is not None:
        ckpt["data"] = {k: ({kk: np.asarray(vv) for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)) for k, v in batch.items()}
    if cfg is not None:
        ckpt["cfg"] = cfg
    ckpt.update(agent.get_params())
    return save(path, ckpt)
