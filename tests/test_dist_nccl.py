"""The RCCL leg on real hardware: `dist.sample_sharded` with the real LDPAgent, one process per GPU,
backend "nccl" (= RCCL over xGMI).  Needs >= 2 GPUs on the node: skipped on the single-GPU box the
-m gpu suite usually runs on (the driver's 8-GPU scaling bench exercises the same path through
bench.py --gpus N; the CPU/gloo twin of this test is tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    from latent_diffusion_planning_amd.dist import sample_sharded
    from tests import cfgs
    from tests.util import idm_params, make_agent, planner_params
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        ag, data = make_agent("rm", planner_params(), idm_params())
        batch = cfgs.synth_latent_batch(data, n, 1, 42)
        action, metrics = sample_sharded(ag, batch, 7)
        ag._engine.check_fault()
        q.put((rank, dist.get_world_size(), np.array(action), np.array(metrics["plan"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 MI355X on one node")
@pytest.mark.parametrize("n", [600, 37])
def test_sharded_sampling_over_rccl_matches_one_gpu(n):
    import torch.multiprocessing as mp
    from tests import cfgs
    from tests.util import assert_close, idm_params, make_agent, planner_params
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ag, data = make_agent("rm", planner_params(), idm_params())
    ag._engine.set_option("no_batch_split", 1)       # 600 plans as ONE loop (512 + 88 otherwise): the regime of the 300-plan shards
    ref_a, ref_m = ag.sample(cfgs.synth_latent_batch(data, n, 1, 42), 7)
    ref_a, ref_p = np.array(ref_a), np.array(ref_m["plan"])
    for rank, seen, a, plan in res:
        assert seen == world
        if n == 600:      # shards of 300 and the full 600 share a launch regime: bitwise
            np.testing.assert_array_equal(plan, ref_p)
        else:             # tiny shards split K / groups differently: equal to fp32 round-off
            assert_close(plan, ref_p, 1e-4, f"rank {rank} plans")
        assert_close(a, ref_a, 1e-4, f"rank {rank} actions")
    ag._engine.close()


_ONE_RANK = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from latent_diffusion_planning_amd.dist import sample_sharded
from tests import cfgs
from tests.util import idm_params, make_agent, planner_params
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[1]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
ag, data = make_agent("rm", planner_params(), idm_params())
batch = cfgs.synth_latent_batch(data, 37, 1, 42)
before = np.array(ag.sample(batch, 7)[0])                      # graphs captured before RCCL exists
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a, m = sample_sharded(ag, batch, 7)                            # replay + a second batch size captured with RCCL's threads alive
b2 = cfgs.synth_latent_batch(data, 300, 1, 43)
a2, m2 = sample_sharded(ag, b2, 9)
out = torch.empty_like(m2["plan"].tensor)
dist.all_gather_into_tensor(out, m2["plan"].tensor)
dist.barrier(); torch.cuda.synchronize()
ag._engine.check_fault()
ok = np.array_equal(np.array(a), before) and torch.equal(out, m2["plan"].tensor) and np.isfinite(np.array(a2)).all()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK" if ok else "RCCL_ONE_RANK_MISMATCH")
"""


def test_graphs_and_rccl_share_a_process():
    """One GPU is enough for this part of the multi-GPU path: the process initialises backend "nccl" (RCCL: its
    proxy / watchdog threads start), the planner+IDM graph is captured and replayed with them alive (thread-local
    capture mode), and the collectives bench.py and dist.sample_sharded use run on the launch stream."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ONE_RANK, str(_free_port())], cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _worker_one_gpu(rank, world, port, n, q):
    """Two ranks, ONE GPU: both drive their own engine on cuda:0 (two engine processes sharing a GPU: DESIGN 4.5) and
    gather their device tensors through backend gloo.  Everything of dist.sample_sharded is real except the wire."""
    import torch.distributed as dist
    from latent_diffusion_planning_amd.dist import sample_sharded
    from tests import cfgs
    from tests.util import idm_params, make_agent, planner_params
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag, data = make_agent("rm", planner_params(), idm_params())
        batch = cfgs.synth_latent_batch(data, n, 1, 42)
        action, metrics = sample_sharded(ag, batch, 7)
        again, _ = sample_sharded(ag, batch, 7)                  # graph replay + second gather
        ag._engine.check_fault()
        q.put((rank, dist.get_world_size(), np.array(action), np.array(metrics["plan"]), np.array(again)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [600, 37])
def test_two_ranks_on_one_gpu_shard_and_gather_the_real_agent(n):
    """the gloo test uses a FakeAgent (no HIP), the RCCL test needs two GPUs.  This one runs on the
    single-GPU box: world size 2, real LDPAgent + real engines in both ranks (sharing cuda:0), row shards keyed by
    global plan index, the all-gather on device tensors (gloo stages them).  Result on every rank == the one-process
    run of the whole batch: bitwise for shards in the batch's launch regime (300 + 300 of 600), round-off otherwise."""
    import torch.multiprocessing as mp
    from tests import cfgs
    from tests.util import assert_close, idm_params, make_agent, planner_params
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_one_gpu, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ag, data = make_agent("rm", planner_params(), idm_params())
    ag._engine.set_option("no_batch_split", 1)       # 600 plans as ONE loop (512 + 88 otherwise): the regime of the 300-plan shards
    ref_a, ref_m = ag.sample(cfgs.synth_latent_batch(data, n, 1, 42), 7)
    ref_a, ref_p = np.array(ref_a), np.array(ref_m["plan"])
    for rank, seen, a, plan, again in res:
        assert seen == world and a.shape == ref_a.shape and plan.shape == ref_p.shape
        assert np.array_equal(a, again)                          # a rank's own result is bit-stable call after call
        # (until round 5 the 300-plan shards and the 600-plan loop happened to run the same tiles -- bitwise equal; the 257..512-plan
        #  regime now has its own eight-wave tiles: a shard equals the single-GPU run to round-off unless both sit in one regime)
        assert_close(plan, ref_p, 1e-5 if n == 600 else 1e-4, f"rank {rank} plans")
        assert_close(a, ref_a, 1e-4, f"rank {rank} actions")
    ag._engine.close()


@pytest.mark.parametrize("config,batch", [(1, 64), (2, 48), (3, 32), (4, 64)])
def test_bench_multi_rank_path_on_one_gpu(config, batch):
    """`python bench.py --gpus 2 --same-gpu --config C`: the N > 1 bench path of every BASELINE.json configuration end to end on the
    one-GPU box -- self-launch through torch.distributed.run, one engine per rank, row offsets, barrier + max-over-ranks timing, the
    all-gather (plans; plans + actions for the aloha configuration) inside the timed region, ONE JSON line from rank 0 that says what it
    is (INVALID: the ranks time-share a GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--same-gpu", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--batch", str(batch), "--config", str(config)], cwd=root, capture_output=True, text=True,
                       timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen_by_backend"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and "INVALID" in d["data"] and "same-gpu" in d["data"]
    assert d["config"]["plans_per_gpu"] == batch and d["steps"] == 2 and d["config"]["baseline_config"] == config
    assert f"configs[{config}]" in d["config"]["workload"] and 0 < d["roofline"]["frac"] < 1
    if config == 4:
        assert d["config"]["shared_cond_plans_per_s"] > 0 and d["config"]["candidates_total"] == 2 * batch
    if config == 3:
        assert "plans and actions" in d["config"]["workload"] and "StableVAE" in d["metric"]
    if config == 2:
        assert "plans and actions" in d["config"]["workload"] and "horizon=17" in d["metric"]
