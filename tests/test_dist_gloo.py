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


# ---- data-parallel training: dist.update_sharded --------------------------------------------------------------
class FakeTrainee(FakeAgent):
    """The shard contract of LDPAgent._update_step on a one-parameter least-squares 'module': loss = mean over the GLOBAL rows, so a rank
    weights its rows' mean by B / n, timesteps are drawn for the global batch and sliced, and the gradients are summed over the ranks."""

    def __init__(self, w=0.5):
        self.w = float(w)

    def _gates(self, step):
        return True, step % 2 == 0

    def _update_step(self, batch, mixed_batch, rng, use_planner, use_idm, noise, shard=None):
        x = torch.as_tensor(batch["obs"]["robot0_eef_pos"]).double().sum(dim=(1, 2))
        B = x.shape[0]
        lo, n = (0, B) if shard is None else shard["rows"]
        t = np.random.Generator(np.random.PCG64(int(rng))).integers(0, 100, size=n)[lo:lo + B]
        y = torch.as_tensor(t).double() / 100.0
        r = self.w * x - y
        loss, grad = (r * r).mean() * (B / n), (2 * r * x).mean() * (B / n)
        extra = {}
        if mixed_batch is not None:
            xm = torch.as_tensor(mixed_batch["obs"]["robot0_eef_pos"]).double().sum(dim=(1, 2))
            lo_m, n_m = (0, xm.shape[0]) if shard is None else shard["mixed_rows"]
            extra = dict(mixed_rows=(lo_m, xm.shape[0], n_m), mixed_sum=float(xm.sum()))
        both = torch.stack([loss, grad])
        if shard is not None:
            dist.all_reduce(both, group=shard.get("group"))
        return FakeTrainee(self.w - 0.1 * float(both[1])), dict(loss=float(both[0]), grad=float(both[1]), rows=(lo, B, n), gates=(use_planner, use_idm), **extra)


def _train_worker(rank, world, port, n, n_mixed, q):
    from latent_diffusion_planning_amd.dist import update_sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag, out = FakeTrainee(), []
        for step in range(3):
            batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, n, 2, 60 + step, with_actions=True)
            mixed = cfgs.synth_latent_batch(cfgs.RM_LIFT, n_mixed, 2, 80 + step, with_actions=True) if n_mixed else None
            ag, m = update_sharded(ag, batch, 7 + step, step, mixed_batch=mixed)
            out.append(m)
        q.put((rank, ag.w, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,n_mixed", [(8, 0), (5, 0), (7, 4), (2, 3)])
def test_sharded_update_equals_the_one_process_update(n, n_mixed):
    """Rows split 2 ways (ragged: 3 + 2, 4 + 3), gates from the global step, mixed batch sharded by its own row count: parameters and
    metrics on both ranks equal the one-process update of the whole batch."""
    from latent_diffusion_planning_amd.dist import update_sharded
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, n, n_mixed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ag, ref = FakeTrainee(), []
    for step in range(3):                                        # no process group: the same entry point, unsharded
        batch = cfgs.synth_latent_batch(cfgs.RM_LIFT, n, 2, 60 + step, with_actions=True)
        mixed = cfgs.synth_latent_batch(cfgs.RM_LIFT, n_mixed, 2, 80 + step, with_actions=True) if n_mixed else None
        ag, m = update_sharded(ag, batch, 7 + step, step, mixed_batch=mixed)
        ref.append(m)
    seen_mixed = [0.0] * 3
    for rank, w, out in res:
        assert abs(w - ag.w) <= 1e-12 * max(1.0, abs(ag.w))
        lo, hi = shard_bounds(n, world, rank)
        for step, (m, r) in enumerate(zip(out, ref)):
            assert m["rows"] == (lo, hi - lo, n) and r["rows"] == (0, n, n)
            assert m["gates"] == r["gates"] == (True, step % 2 == 0)
            assert abs(m["loss"] - r["loss"]) <= 1e-12 * max(1.0, r["loss"]) and abs(m["grad"] - r["grad"]) <= 1e-12 * max(1.0, abs(r["grad"]))
            if n_mixed:
                lm, hm = shard_bounds(n_mixed, world, rank)
                assert m["mixed_rows"] == (lm, hm - lm, n_mixed)
                seen_mixed[step] += m["mixed_sum"]
    if n_mixed:
        assert all(abs(a - r["mixed_sum"]) <= 1e-9 for a, r in zip(seen_mixed, ref))


def test_sharded_update_refuses_more_ranks_than_rows():
    from latent_diffusion_planning_amd.dist import update_sharded
    import latent_diffusion_planning_amd.dist as D
    old = (D.dist.is_initialized, D.dist.get_world_size, D.dist.get_rank)
    D.dist.is_initialized, D.dist.get_world_size, D.dist.get_rank = (lambda: True), (lambda g=None: 4), (lambda g=None: 1)
    try:
        with pytest.raises(ValueError, match="cannot feed 4 ranks"):
            update_sharded(FakeTrainee(), cfgs.synth_latent_batch(cfgs.RM_LIFT, 3, 2, 1, with_actions=True), 0, 0)
    finally:
        D.dist.is_initialized, D.dist.get_world_size, D.dist.get_rank = old
