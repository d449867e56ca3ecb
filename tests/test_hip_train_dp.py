"""Data-parallel training on real hardware: dist.update_sharded with the real LDPAgent.  Two ranks share cuda:0 (the -m gpu box has one GPU)
and exchange the gradient arenas through backend gloo (which stages device tensors); with >= 2 GPUs the same worker runs one rank per GPU
over backend "nccl" (RCCL).  Everything but the wire is the product path: shard rows keyed by the global row index, B / n weighted losses,
ONE all-reduce per module in place on the engine's gradient arena (ldp_train_arena), replicated global norm + Adam."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_STEPS = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches(data, n, n_mixed, step):
    from tests import cfgs
    b = cfgs.synth_latent_batch(data, n, 9, 300 + step, with_actions=True)
    mb = cfgs.synth_latent_batch(data, n_mixed, 9, 400 + step, with_actions=True) if n_mixed else None
    return b, mb


def _run(ag, data, n, n_mixed, step_fn):
    from latent_diffusion_planning_amd import weights as W
    from tests.util import tree_digest
    out = dict(metrics=[])
    for s in range(N_STEPS):
        b, mb = _batches(data, n, n_mixed, s)
        ag, m = step_fn(ag, b, mb, 100 + s, s)
        out["metrics"].append({k: float(m[k]) for k in ("plan_loss", "idm_loss", "loss", "g_norm", "planner_lr", "idm_lr")})
        if s == 0:
            e = ag._engine
            out["grads_planner"] = tree_digest(e.train_read("planner", e.TRAIN_GRADS, W.planner_shapes(ag._planner_spec)), 21)
            out["grads_idm"] = tree_digest(e.train_read("idm", e.TRAIN_GRADS, W.idm_shapes(ag._idm_spec)), 22)
    pp, ip = ag.planner_state.params, ag.idm_state.params
    out["planner"], out["idm"] = tree_digest(pp, 23), tree_digest(ip, 24)
    out["leaf"] = np.array(ip["MLPResNet_0/Dense_0/kernel"])
    out["steps"] = (ag.planner_state.step, ag.idm_state.step)
    return ag, out


def _worker(rank, world, port, backend, n, n_mixed, q):
    import torch.distributed as dist
    from latent_diffusion_planning_amd.dist import update_sharded
    from tests.util import idm_params, make_agent, planner_params
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    kw = dict(device_id=torch.device("cuda", dev)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        ag, data = make_agent("rm", planner_params(), idm_params())
        ag, out = _run(ag, data, n, n_mixed, lambda a, b, mb, rng, s: update_sharded(a, b, rng, s, mixed_batch=mb))
        g = ag._engine.train_arena("idm", ag._engine.TRAIN_GRADS)
        out["arena"] = (int(g.numel()), bool(g.is_cuda), str(g.dtype))
        sb = __import__("tests.cfgs", fromlist=["x"]).synth_latent_batch(data, 3, 1, 5)
        out["action"] = np.array(ag.sample(sb, 9)[0])            # the trained replica samples (publish) -- identical on every rank
        ag._engine.check_fault()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,n_mixed", [(16, 0), (7, 5)])
def test_two_ranks_train_the_real_agent_like_one(n, n_mixed):
    import torch.multiprocessing as mp
    from tests.util import idm_params, make_agent, planner_params
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, n, n_mixed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ag, data = make_agent("rm", planner_params(), idm_params())
    step = (lambda a, b, mb, rng, s: a.update(b, rng, s)) if not n_mixed else (lambda a, b, mb, rng, s: a.update_mixed(b, mb, rng, s))
    ag, ref = _run(ag, data, n, n_mixed, step)
    moved = float(np.abs(ref["leaf"] - np.array(idm_params()["MLPResNet_0/Dense_0/kernel"])).mean())
    for rank, out in res:
        assert out["steps"] == ref["steps"] == (N_STEPS, N_STEPS)
        assert out["arena"][1:] == (True, "torch.float32") and out["arena"][0] >= ref["leaf"].size
        for s, (m, r) in enumerate(zip(out["metrics"], ref["metrics"])):
            for k in m:
                assert abs(m[k] - r[k]) <= 2e-5 * max(1.0, abs(r[k])), (rank, s, k, m[k], r[k])
        for k in ("grads_planner", "grads_idm"):                 # the summed shard gradients == the whole batch's (digest columns on the leaf's scale)
            scale = np.maximum(ref[k][:, 1:2], 1e-30)
            err = (np.abs(out[k] - ref[k]) / scale)[:, 3:].max()
            print(f"n={n} rank {rank}: {k}: worst digest entry off by {err:.2e} of its leaf's max")
            assert err <= 1e-4, (rank, k, err)
        for k in ("planner", "idm"):
            assert np.abs(out[k][:, 3:] - ref[k][:, 3:]).max() <= 1e-5, (rank, k)
        d = np.abs(out["leaf"] - ref["leaf"])
        print(f"n={n} rank {rank}: IDM Dense_0 after {N_STEPS} steps: mean |dp - one process| {d.mean():.2e}, max {d.max():.2e}; mean movement {moved:.2e}")
        assert d.mean() <= 0.02 * moved
    # the replicas never diverge: same gradients in, same Adam out -- bitwise
    for k in ("planner", "idm", "leaf", "action", "grads_planner", "grads_idm"):
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    ag._engine.close()


def test_the_gradient_arena_aliases_engine_memory():
    """train_arena hands torch a view (no copy): what torch writes, the engine's leaf reader sees; padding of the gradient arena is zero."""
    from latent_diffusion_planning_amd import weights as W
    from latent_diffusion_planning_amd.engine import HipEngine
    from tests.util import idm_params, planner_params
    D, A, T = 25, 7, 8
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    ip = idm_params(D=D, A=A)
    e.load_params(planner=planner_params(D=D), idm=ip)
    e.train_init(["idm"])
    g = np.random.Generator(np.random.PCG64(4))
    R = 24
    e.train_idm_grad(torch.tensor(g.uniform(-1, 1, (R, 2 * D)).astype(np.float32)), torch.tensor(g.uniform(-1, 1, (R, A)).astype(np.float32)),
                     torch.tensor(g.standard_normal((R, A)).astype(np.float32)), g.integers(0, 100, R))
    arena = e.train_arena("idm", e.TRAIN_GRADS)
    shapes = W.idm_shapes(W.IDMSpec(D, A))
    leaves = e.train_read("idm", e.TRAIN_GRADS, shapes)
    tot = sum(float(np.abs(v.astype(np.float64)).sum()) for v in leaves.values())
    assert abs(float(arena.double().abs().sum()) - tot) <= 1e-6 * tot          # nothing but the leaves is non-zero
    n1 = float(e.train_grad_norm(["idm"]))
    arena.mul_(0.5)                                                             # torch writes ...
    assert abs(float(e.train_grad_norm(["idm"])) - 0.5 * n1) <= 1e-6 * n1       # ... the engine reads
    half = e.train_read("idm", e.TRAIN_GRADS, shapes)
    assert all(np.array_equal(half[k], 0.5 * leaves[k]) for k in leaves)
    p = e.train_arena("idm", e.TRAIN_PARAMS)
    assert p.numel() == arena.numel() and p.data_ptr() != arena.data_ptr()
    with pytest.raises(Exception):
        e.train_arena("planner", e.TRAIN_GRADS)                                 # not initialised for training
    e.close()


def _hier_agent():
    from latent_diffusion_planning_amd.hier_agent import LDPHierAgent
    from tests import cfgs
    from tests.cases import hier_idm_params
    from tests.util import planner_params
    data = cfgs.RM_LIFT
    ag = LDPHierAgent.create(0, None, data["shape_meta"], **cfgs.hier_kwargs(data))
    return ag.replace(planner_state=ag.planner_state.replace(params=planner_params()),
                      idm_state=ag.idm_state.replace(params=hier_idm_params())), data


def _hier_run(ag, data, n, step_fn):
    from latent_diffusion_planning_amd import weights as W
    from tests import cfgs
    from tests.util import tree_digest
    out = dict(metrics=[])
    for s in range(2):
        b = cfgs.synth_latent_batch(data, n, 33, 500 + s, with_actions=True)
        ag, m = step_fn(ag, b, 100 + s, s)
        out["metrics"].append({k: float(m[k]) for k in ("plan_loss", "idm_loss", "g_norm")})
        if s == 0:
            out["grads_planner"] = tree_digest(ag._engine.train_read("planner", ag._engine.TRAIN_GRADS, W.planner_shapes(ag._planner_spec)), 31)
            out["grads_idm"] = tree_digest(ag._idm_engine.train_read("planner", ag._idm_engine.TRAIN_GRADS, W.planner_shapes(ag._idm_unet_spec)), 32)
    out["planner"], out["idm"] = tree_digest(ag.planner_state.params, 33), tree_digest(ag.idm_state.params, 34)
    return ag, out


def _hier_worker(rank, world, port, n, q):
    import torch.distributed as dist
    from latent_diffusion_planning_amd.dist import update_sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag, data = _hier_agent()
        ag, out = _hier_run(ag, data, n, lambda a, b, rng, s: update_sharded(a, b, rng, s))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_train_the_hierarchical_agent_like_one():
    """dist.update_sharded on LDPHierAgent: both U-Nets' gradient arenas (one per engine handle) summed over the ranks; 5 rows split 3 + 2."""
    import torch.multiprocessing as mp
    n, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hier_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ag, data = _hier_agent()
    ag, ref = _hier_run(ag, data, n, lambda a, b, rng, s: a.update(b, rng, s))
    for rank, out in res:
        for s, (m, r) in enumerate(zip(out["metrics"], ref["metrics"])):
            for k in m:
                assert abs(m[k] - r[k]) <= 2e-5 * max(1.0, abs(r[k])), (rank, s, k, m[k], r[k])
        for k in ("grads_planner", "grads_idm"):
            scale = np.maximum(ref[k][:, 1:2], 1e-30)
            err = (np.abs(out[k] - ref[k]) / scale)[:, 3:].max()
            print(f"hier rank {rank}: {k}: worst digest entry off by {err:.2e} of its leaf's max")
            assert err <= 1e-4, (rank, k, err)
        for k in ("planner", "idm"):
            assert np.abs(out[k][:, 3:] - ref[k][:, 3:]).max() <= 1e-5, (rank, k)
    for k in ("planner", "idm", "grads_planner", "grads_idm"):
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
    ag._engine.close(); ag._idm_engine.close()
