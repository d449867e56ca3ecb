"""Bulk latent pre-encoding -- counterpart of process_sdvae_data.py:55-121 (`run_rm`): encodes
every frame of every episode with the StableVAE encoder in shards of `shard` images (the ragged
tail is zero-padded to a full shard and trimmed, like the reference), and tracks the running
min / max of the latents (`min_z`, `max_z` attributes; the reference initialises both at 0).

h5py is not available in this environment, so episodes come in as arrays and the result goes out
as arrays / an .npz with the reference's dataset paths as keys
(`data/<ep>/latent/<rgb_key>`, `data.attrs/min_z`, ...).  A maintainer with h5py only has to
swap the container.
"""
from __future__ import annotations

from typing import Dict, Iterable, Mapping, Tuple

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
        outs.append(engine.vae_encode(part)[:n])
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


def save_latents(path: str, latents: Dict[str, np.ndarray], attrs: dict) -> None:
    np.savez_compressed(path, **latents, **{f"data.attrs/{k}": np.asarray(v) for k, v in attrs.items()})
