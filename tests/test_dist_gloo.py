"""world_size-2 CPU (gloo) test of the sharded sampling path: shard -> sample -> one all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from latent_diffusion_planning_amd.dist import all_gather_rows, sample_sharded, shard_bounds
from tests import cfgs


class FakeAgent:
    """Stands in for the GPU agent: a deterministic function of (row content, global row index),
    like the Philox stream keyed by row_offset."""
    config = dict(action_horizon=4, action_dim=7, obs_dim=25)
    _device = torch.device("cpu")

    def sample(self, batch, rng, row_offset=0, **kw):
        x = torch.as_tensor(batch["obs"]["robot0_eef_pos"]).float()
        n = x.shape[0]
        rows = torch.arange(row_offset, row_offset + n, dtype=torch.float32)
        base = x.sum(dim=(1, 2)) + 1000.0 * rows + float(rng)
        action = base[:, None, None] + torch.zeros(n, 4, 7)
        plan = -base[:, None, None] + torch.zeros(n, 5, 25)
        return action, {"plan": plan}

    sample_viz = sample


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, n, 1, 42)
        action, metrics = sample_sharded(FakeAgent(), batch, 7)
        lo, hi = shard_bounds(n, world, rank)
        ragged = all_gather_rows(torch.full((hi - lo, 3), float(rank)), n)
        q.put((rank, np.array(action), np.array(metrics["plan"]), ragged.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 5, 1])
def test_sharded_sampling_matches_single_process(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, n, 1, 42)
    ref_a, ref_m = FakeAgent().sample_viz(batch, 7, row_offset=0)
    for rank, a, plan, ragged in res:
        np.testing.assert_array_equal(a, ref_a.numpy())                 # every rank holds the full result
        np.testing.assert_array_equal(plan, ref_m["plan"].numpy())
        expect = np.concatenate([np.full((shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0], 3), float(r))
                                 for r in range(world)])
        np.testing.assert_array_equal(ragged, expect)


def test_single_process_path_needs_no_process_group():
    batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, 4, 1, 1)
    a, m = sample_sharded(FakeAgent(), batch, 3)
    ref_a, _ = FakeAgent().sample_viz(batch, 3)
    assert np.array_equal(np.array(a), ref_a.numpy())


class FaultyAgent(FakeAgent):
    """Returns record-carrying DeviceArrays whose completion hook swaps in the 'recomputed' tensors, like LDPAgent
    after a fault of the in-launch exchanges (the sharded path used to gather the unchecked tensors)."""

    def sample(self, batch, rng, row_offset=0, **kw):
        from latent_diffusion_planning_amd.arrays import CallRecord, DeviceArray
        action, m = super().sample(batch, rng, row_offset=row_offset, **kw)
        self.hook_runs = 0

        def hook(rec):
            self.hook_runs += 1
            for arr, t in zip(rec.arrays, (action, m["plan"])):
                arr._swap(t)
        rec = CallRecord(hook)
        bad_a, bad_p = torch.full_like(action, float("nan")), torch.full_like(m["plan"], float("nan"))
        return DeviceArray(bad_a, record=rec), {"plan": DeviceArray(bad_p, record=rec)}


def test_sharded_path_completes_the_local_call_before_gathering():
    batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, 4, 1, 1)
    ag = FaultyAgent()
    a, m = sample_sharded(ag, batch, 3)
    ref_a, ref_m = FakeAgent().sample_viz(batch, 3)
    assert ag.hook_runs == 1                                         # one record, completed once, before the tensors left
    assert np.array_equal(np.array(a), ref_a.numpy()) and np.array_equal(np.array(m["plan"]), ref_m["plan"].numpy())


def test_a_failed_completion_hook_leaves_the_record_incomplete():
    """completed=True used to be set before the hook ran, so a raising safe-mode recompute left the arrays
    serving the faulted tensors without any further check."""
    from latent_diffusion_planning_amd.arrays import CallRecord, DeviceArray
    calls = []

    def hook(rec):
        calls.append(1)
        if len(calls) == 1:
            raise RuntimeError("faulted again in safe mode")
        rec.arrays[0]._swap(torch.ones(2))
    rec = CallRecord(hook)
    arr = DeviceArray(torch.full((2,), float("nan")), record=rec)
    with pytest.raises(RuntimeError):
        arr.numpy()
    assert not rec.completed
    assert np.array_equal(arr.numpy(), np.ones(2, np.float32)) and rec.completed and len(calls) == 2
