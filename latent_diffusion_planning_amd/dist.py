"""Multi-GPU sampling: one process per GPU, plans sharded by row, one RCCL all-gather.

The path is embarrassingly data-parallel: every plan is independent through VAE
encode, planner loop and IDM loop, so the only collective is the all-gather of the sampled
trajectories (and actions).  The reference's only multi-device construct is batch
PositionalSharding (utils/py_utils.py:27-39); this is its MI355X counterpart.

Each rank keys its Philox rows by the *global* plan index (row_offset), so the noise a plan sees does
not depend on the world size; the gathered result equals a single-GPU run of the whole batch bitwise
when shard and full batch run in the same launch regime, to fp32 round-off otherwise.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of n rows: the first n % world ranks get one extra row."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: dict, world: int, rank: int) -> Tuple[dict, int, int]:
    """Slice every leaf of {'obs': {...}[, 'actions': ...]} along axis 0."""
    n = len(next(iter(batch["obs"].values())))
    lo, hi = shard_bounds(n, world, rank)
    out = {"obs": {k: v[lo:hi] for k, v in batch["obs"].items()}}
    if "actions" in batch:
        out["actions"] = batch["actions"][lo:hi]
    return out, lo, n


def all_gather_rows(x: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather a row-sharded tensor whose shards follow shard_bounds (ragged allowed):
    shards are padded to the largest shard, gathered with ONE collective and trimmed."""
    world = dist.get_world_size(group)
    if world == 1:
        return x
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = torch.empty((world * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if all(hi - lo == mx for lo, hi in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sample_sharded(agent, batch: dict, eval_rng, group=None, **kw):
    """`agent.sample(batch, rng)` with the batch rows split over the ranks of `group`.
    Every rank passes the *full* batch and receives the *full* (action, {'plan': ...}) as DeviceArrays
    (device tensors under `.tensor`).  One host synchronisation per call: the local shard is completed (fault poll)
    before its rows are gathered."""
    from .arrays import DeviceArray, as_tensor
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    local, lo, n = shard_batch(batch, world, rank)
    nloc = len(next(iter(local["obs"].values())))
    if nloc > 0:
        action, metrics = agent.sample(local, eval_rng, row_offset=lo, **kw)
        # The rows are about to leave this rank: this is the local call's completion point.  Wait for it and poll the
        # fault word of the in-launch exchanges NOW (on a fault the shard is recomputed in safe mode and the tensors
        # are swapped) -- gathered record-less copies could never be recalled afterwards.
        for arr in (action, metrics["plan"]):
            if isinstance(arr, DeviceArray):
                arr.complete()
        action, plan = as_tensor(action), as_tensor(metrics["plan"])
    else:                                             # more ranks than rows
        cfg = agent.config
        dev = agent._device
        action = torch.zeros((0, cfg["action_horizon"], cfg["action_dim"]), device=dev)
        plan = torch.zeros((0, cfg["action_horizon"] + 1, cfg["obs_dim"]), device=dev)
    if world > 1:
        # synchronous collectives: the launch stream waits for them, so the next planner graph (whose split
        # work-groups want the whole chip) never overlaps an RCCL kernel
        action, plan = all_gather_rows(action, n, group), all_gather_rows(plan, n, group)
    return DeviceArray(action), {"plan": DeviceArray(plan)}


def update_sharded(agent, batch: dict, rng, step: int, group=None, mixed_batch=None, noise=None):
    """`agent.update(batch, rng, step)` / `agent.update_mixed(batch, mixed_batch, rng, step)` (agent/ldp_agent.py:223-323; LDPHierAgent alike) with the batch rows
    split over the ranks of `group`: the MI355X counterpart of the reference's training-time PositionalSharding (utils/py_utils.py:27-39 +
    jit: XLA inserts the gradient all-reduce).  Every rank passes the FULL batch(es) and keeps a full replica of parameters and Adam state;
    a rank computes forward / backward of its rows only, the flat gradient arena of each trained module crosses the wire ONCE (one RCCL
    all-reduce per module, in place on the engine's memory, on the launch stream) and global norm + Adam run replicated, so the replicas
    never diverge.  Timesteps and noise are keyed by the global row index: the step equals the one-GPU step of the whole batch to fp32
    round-off, for any world size.  Shards may be ragged (losses are weighted rows / total rows) but not empty.
    The statistics scalars of the metrics dict (emb_*, action_*, <obs key>_*) describe this rank's rows."""
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    use_planner, use_idm = agent._gates(int(step))
    local, lo, n = shard_batch(batch, world, rank)
    if world > n:
        raise ValueError(f"update_sharded: {n} batch rows cannot feed {world} ranks")
    shard = {"group": group, "rows": (lo, n)}
    local_m = None
    if mixed_batch is not None:
        local_m, lo_m, n_m = shard_batch(mixed_batch, world, rank)
        if world > n_m:
            raise ValueError(f"update_sharded: {n_m} mixed-batch rows cannot feed {world} ranks")
        shard["mixed_rows"] = (lo_m, n_m)
    if world == 1:
        shard = None
    return agent._update_step(local, local_m, rng, use_planner, use_idm, noise, shard)
