"""Bulk latent pre-encoding -- counterpart of process_sdvae_data.py:55-121 (`run_rm`): encodes
every frame of every episode with the StableVAE encoder in shards of `shard` images (the ragged
tail is zero-padded to a full shard and trimmed, like the reference), and tracks the running
min / max of the latents (`min_z`, `max_z` attributes; the reference initialises both at 0).

Containers: `encode_hdf5` is the file-to-file form of `run_rm` / `run_aloha` -- it reads the robomimic image file
(`data/<ep>/obs/<rgb_key>`, last row of `data/<ep>/next_obs/<rgb_key>`) and writes the `latent.hdf5` the data layer
opens (data/robomimic_latent_data.py:94-96): `data/<ep>/latent/<rgb_key>` float32 datasets, `data` attributes
`total` / `min_z` / `max_z`.  h5py is not installed for this interpreter; hdf5_io.py drives the HDF5 C library
directly.  `encode_dataset` + `save_latents` is the in-memory form (arrays in, .npz or .hdf5 out).
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import HipEngine


def encode_frames(engine: HipEngine, frames: np.ndarray, shard: int = 128) -> torch.Tensor:
    """frames (T, H, W, 3) raw [0,255] -> latent means (T, h, w, c) on the GPU.
    Pre-processing as process_sdvae_data.py:88-90: x/255, then (x-0.5)/0.5."""
    x = torch.as_tensor(np.asarray(frames, dtype=np.float32)).to(engine.device)
    x = engine.normalize_bounds(x, [0.0], [255.0], True)             # == (x/255 - 0.5)/0.5
    outs = []
    for i in range(0, x.shape[0], shard):
        part = x[i:i + shard]
        n = part.shape[0]
        if n < shard:                                                # zero-pad the ragged tail
            part = torch.cat([part, torch.zeros((shard - n,) + tuple(part.shape[1:]), device=x.device)], dim=0)
        outs.append(engine.vae_encode_checked(part)[:n])       # range guard of the fp16-plane convs: re-encoded on bf16 planes if it fires
    return torch.cat(outs, dim=0)


def encode_dataset(engine: HipEngine, episodes: Mapping[str, Mapping[str, np.ndarray]], shard: int = 128
                   ) -> Tuple[Dict[str, np.ndarray], dict]:
    """episodes: {ep_name: {rgb_key: frames (T+1, H, W, 3)}} (obs frames + the final next_obs frame,
    process_sdvae_data.py:80-85).  Returns ({"data/<ep>/latent/<key>": (T+1, h, w, c)}, attrs)."""
    out: Dict[str, np.ndarray] = {}
    min_z, max_z = 0.0, 0.0
    for ep, obs in episodes.items():
        for key, frames in obs.items():
            z = encode_frames(engine, frames, shard)
            min_z = min(min_z, float(z.min()))
            max_z = max(max_z, float(z.max()))
            out[f"data/{ep}/latent/{key}"] = z.cpu().numpy()
    return out, {"total": len(episodes), "min_z": min_z, "max_z": max_z}


def save_latents(path: str, latents: Dict[str, np.ndarray], attrs: dict, fmt: Optional[str] = None) -> None:
    """fmt "hdf5" (default for *.hdf5 / *.h5): the reference's latent.hdf5 layout; "npz": dataset paths as keys."""
    fmt = fmt or ("hdf5" if path.endswith((".hdf5", ".h5")) else "npz")
    if fmt == "hdf5":
        from .hdf5_io import write_latent_file
        write_latent_file(path, latents, attrs)
    elif fmt == "npz":
        np.savez_compressed(path, **latents, **{f"data.attrs/{k}": np.asarray(v) for k, v in attrs.items()})
    else:
        raise ValueError(f"unknown latent container {fmt!r}")


def demo_order(names: Sequence[str]):
    """`demo_<n>` by n (data/robomimic_latent_data.py:45-47); other names keep name order."""
    try:
        return sorted(names, key=lambda e: int(e[5:]))
    except ValueError:
        return sorted(names)


def encode_hdf5(engine: HipEngine, image_path: str, latent_path: str, rgb_keys: Sequence[str], shard: int = 128,
                append_last_next_obs: bool = True, demos: Optional[Sequence[str]] = None) -> dict:
    """File-to-file counterpart of `run_rm` (append_last_next_obs=True, process_sdvae_data.py:78-85) and `run_aloha`
    (False, :147-148): one episode's frames are in memory at a time.  Returns the attributes written."""
    from .hdf5_io import File
    min_z, max_z = 0.0, 0.0
    with File(image_path, "r") as src, File(latent_path, "w") as dst:
        dst.require_group("data")
        eps = list(demos) if demos is not None else demo_order(src.keys("data"))
        for ep in eps:
            dst.require_group(f"data/{ep}")
            for key in rgb_keys:
                frames = src.read_dataset(f"data/{ep}/obs/{key}")
                if append_last_next_obs:
                    frames = np.concatenate([frames, src.read_dataset(f"data/{ep}/next_obs/{key}", -1)], axis=0)
                z = encode_frames(engine, frames, shard)
                min_z = min(min_z, float(z.min()))
                max_z = max(max_z, float(z.max()))
                dst.write_dataset(f"data/{ep}/latent/{key}", z.cpu().numpy())
        attrs = {"total": len(eps), "min_z": min_z, "max_z": max_z}
        for k, v in attrs.items():
            dst.write_attr("data", k, v)
    return attrs
