#!/usr/bin/env python3
"""bench.py -- latent plans/sec of the LDP denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,3,4}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: with WORLD_SIZE unset and --gpus N > 1 this script launches the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`),
one process per GPU, backend "nccl" (= RCCL over xGMI); rank 0 prints the ONE JSON line.

One "step" = one full pass of the hot path over one batch.  `--config` picks the BASELINE.json configuration:
  1 (default, the driver's line) configs[1]: rm_lift planner ConditionalUnet1D (D=25, T=8), 100-step DDIM, 256 synthetic
    latents per GPU, the whole loop replayed from one hipGraph;
  3 configs[3]: aloha sim_transfer_cube -- raw 64x64 wrist frames -> StableVAE encode -> DDPM-100 planner -> DDPM-100 IDM
    through LDPAgent.sample, 512 frames per GPU (`--gpus 4` = the 2048-frame configuration BASELINE.json names);
  4 configs[4]: rm_can best-of-N candidates -- 50-step DDIM planner, 1024 candidates per GPU (`--gpus 8` = 8192), timed
    once with N independent observations and once with ONE observation broadcast to all candidates (SURVEY 8d asks for both).
Weak scaling in every case: each rank samples its own rows (independent Philox rows keyed by the global plan index) and the
sampled trajectories (+ actions) are all-gathered over RCCL inside the timed region -- a synchronous collective on the launch
stream: it orders the ranks by itself, there is no barrier inside the timed region (the two fences bracket it).  Inputs are
resident in HBM before the timed region starts.

`--dry-run` exercises only the launcher and the collective plumbing on CPU (backend gloo, no
GPU work, the line is marked INVALID): it is what the CPU test-suite runs.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[i]: 1 = rm_lift planner DDIM-100, 256 plans / GPU (the driver's line); 2 = rm_square T = 16 planner + IDM "
                         "as one graph, 1024 plans / GPU; 3 = aloha raw "
                         "frames -> StableVAE encode -> planner + IDM, 512 / GPU; 4 = rm_can DDIM-50 candidates, 1024 / GPU")
    ap.add_argument("--batch", type=int, default=None, help="plans per GPU (default: 256 / 1024 / 512 / 1024 for --config 1 / 2 / 3 / 4)")
    ap.add_argument("--sampler", default=None, choices=["ddim", "ddpm"])
    ap.add_argument("--n-steps", type=int, default=None, help="denoising steps per plan")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="ldp_set_option before the run (tools/: work-split switches, timing ablations)")
    ap.add_argument("--lib", default=None, metavar="PATH",
                    help="tools/: load this build of libldp_hip instead of the in-tree one (same-box A/B of two builds; "
                         "the line is marked INVALID)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="tests/: all N ranks share GPU 0 and gather over gloo -- exercises the N > 1 code path (self-launch, "
                         "row offsets, barrier, max over ranks, all-gather inside the timed region) on a one-GPU box; the "
                         "line is marked INVALID (the ranks time-share one GPU)")
    ap.add_argument("--configs0", action="store_true",
                    help="NOT the driver line: BASELINE.json configs[0] (the reference's own CPU-runnable case: rm_lift, DDPM-100 "
                         "planner + DDPM-100 IDM, B in {1, 16, 256}) timed on the host cores through this file's cpu_baseline leg, "
                         "next to LDPAgent.sample at the same B on the GPU; prints one JSON object")
    ap.add_argument("--train", action="store_true",
                    help="NOT the driver line: the training step (LDPAgent.update: losses, gradients, global norm, Adam for planner + IDM, "
                         "agent/ldp_agent.py:223-272) at train_bc.yaml's batch size of 256 on synthetic latent batches; prints one JSON object")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher/collective check on CPU (gloo); no GPU work, the line is INVALID")
    a = ap.parse_args(argv)
    dflt = {1: (256, "ddim", 100), 2: (1024, "ddpm", 100), 3: (512, "ddpm", 100), 4: (1024, "ddim", 50)}[a.config]
    a.batch = dflt[0] if a.batch is None else a.batch
    a.sampler = dflt[1] if a.sampler is None else a.sampler
    a.n_steps = dflt[2] if a.n_steps is None else a.n_steps
    return a


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` -> N ranks under torch.distributed.run
# ------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if args.same_gpu and have >= 1:
            have = args.gpus
        if have < args.gpus:
            sys.stderr.write(f"bench.py --gpus {args.gpus} needs {args.gpus} MI355X GPUs on this node, found {have}\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY 8d "config 1"): the torch-CPU restatement of the reference path
# ------------------------------------------------------------------------------------------------
def cpu_baseline(pp, D, T, sampler, n_steps, budget_s=45.0):
    """The SAME workload as the GPU metric of this line -- the planner alone, `n_steps`-step `sampler`, synthetic latents -- on the
    torch-CPU restatement (oracle/torch32.py, fp32).  `value` is B = 256 (the batch the GPU line runs; the best of the three): FULL loops,
    every denoising step executed, median of 3 runs while the budget lasts (about 10 s per 100-step loop on 32 threads; at least one full
    run).  B = 1 and B = 16 are reported beside it from 20 of the n_steps steps, scaled (every step costs the same: same network, same
    shapes) -- they are context, not the value."""
    import numpy as np
    import torch
    from oracle import torch32
    # torch's intra-op pool stops scaling (and then collapses) long before the 100+ hardware
    # threads of a GPU host on these small convolutions: use at most 32 and report that count.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    PP = torch32.TorchParams(pp)
    g = np.random.Generator(np.random.PCG64(1))
    rows = {}
    t_start = time.perf_counter()
    for B, s, runs in ((256, n_steps, 3), (1, 20, 1), (16, 20, 1)):
        s = min(s, n_steps)
        cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32)
        x0 = torch.tensor(g.standard_normal((B, T, D)), dtype=torch.float32)
        xn = torch.tensor(g.standard_normal((n_steps, B, T, D)), dtype=torch.float32) if sampler == "ddpm" else None

        def run(n):
            torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=n_steps, sampler=sampler, stop_after=n)
        run(1)                                                           # warm-up
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            run(s)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s:
                break
        per_call = statistics.median(ts) * n_steps / s
        rows[B] = dict(plans_per_s=round(B / per_call, 4), s_per_call=round(per_call, 3), steps_timed=s, runs=len(ts))
    best = rows[256]
    return {"value": best["plans_per_s"], "unit": "plans/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"the GPU line's own workload (planner ConditionalUnet1D alone, {n_steps}-step {sampler.upper()}, synthetic latents, B = 256) on "
                      f"oracle/torch32.py, fp32 torch-CPU: median of {best['runs']} FULL {n_steps}-step loops ({best['s_per_call']} s each); beside it, from 20 "
                      "steps scaled: " + "; ".join(f"B={b}: {r['plans_per_s']} plans/s" for b, r in rows.items() if b != 256)
                      + "; proxy for the JAX-CPU reference (JAX is not installable here)",
            "per_batch": {str(b): r for b, r in rows.items()}}


def configs0():
    """BASELINE.json configs[0] kept in the evidence set: the reference path as BASELINE.json words it --
    rm_lift latent_img ldp_agent, 100-step DDPM planner (horizon 9 = obs + 8) AND 100-step DDPM IDM -- on the torch-CPU
    restatement (oracle/torch32.py; the JAX-CPU original cannot run here), B in {1, 16, 256}, 20 / 20 / 10 of the 100
    steps of each loop timed and scaled, median of 3; next to it LDPAgent.sample at the same B on the GPU, host arrays in
    and host action out (agent/ldp_agent.py:452-506)."""
    import numpy as np
    import torch
    from latent_diffusion_planning_amd import weights as W
    from latent_diffusion_planning_amd.agent import LDPAgent
    from oracle import torch32
    from tests import cfgs
    D, A, T = 25, 7, 8
    pp, ip = W.init_planner_params(W.PlannerSpec(D, D), 0), W.init_idm_params(W.IDMSpec(D, A), 1)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    PP, PI = torch32.TorchParams(pp), torch32.TorchParams(ip)
    data = cfgs.RM_LIFT
    ag = LDPAgent.create(0, None, data["shape_meta"], **cfgs.agent_kwargs(data))
    g = np.random.Generator(np.random.PCG64(7))
    rows = {}
    for B, s in ((1, 20), (16, 20), (256, 10)):
        cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32)
        x0 = torch.tensor(g.standard_normal((B, T, D)), dtype=torch.float32)
        xn = torch.tensor(g.standard_normal((100, B, T, D)), dtype=torch.float32)
        tr = torch.tensor(g.uniform(-1, 1, (B * 4, 2 * D)), dtype=torch.float32)
        a0 = torch.tensor(g.standard_normal((B * 4, A)), dtype=torch.float32)
        an = torch.tensor(g.standard_normal((100, B * 4, A)), dtype=torch.float32)

        def timed(fn):
            fn(1)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn(s)
                ts.append(time.perf_counter() - t0)
            return statistics.median(ts) * 100 / s
        t_pl = timed(lambda n: torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=100, sampler="ddpm", stop_after=n))
        t_id = timed(lambda n: torch32.idm_sample(PI, tr, a0, an, n_train=100, n_steps=100, sampler="ddpm", stop_after=n))
        batch = cfgs.synth_latent_batch(data, B, 1, 3)
        for _ in range(3):
            np.array(ag.sample(batch, 1)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            np.array(ag.sample(batch, 2 + i)[0])
        t_gpu = (time.perf_counter() - t0) / 5
        rows[f"B{B}"] = dict(cpu_planner_s=round(t_pl, 3), cpu_idm_s=round(t_id, 3), cpu_plans_per_s=round(B / (t_pl + t_id), 3),
                             steps_timed=s, gpu_agent_sample_ms=round(t_gpu * 1e3, 2), gpu_plans_per_s=round(B / t_gpu, 1),
                             gpu_over_cpu=round((t_pl + t_id) / t_gpu, 1))
    ag._engine.close()
    return {"workload": "BASELINE.json configs[0]: rm_lift latent_img ldp_agent, horizon 9 (obs + 8), 100-step DDPM planner + 100-step DDPM IDM, fp32",
            "cpu": {"kind": "port", "what": "oracle/torch32.py on torch-CPU: proxy for the JAX-CPU reference path (jax is not installable here)",
                    "cores": int(torch.get_num_threads()), "host_cores": os.cpu_count(),
                    "sample": "20 / 20 / 10 of the 100 steps of each loop timed at B = 1 / 16 / 256 and scaled, median of 3"},
            "gpu": "LDPAgent.sample, host arrays in / host action out, one hipGraph per call, 1 x MI355X",
            "NOT_THE_DRIVER_LINE": True, **rows}


def pmc_traffic(B, args):
    """(HBM bytes per conv launch, source file).  PMC counters cannot be read from inside the process: the number
    comes from the committed rocprofv3 --pmc passes of this very command (tools/pmc_passes.sh ->
    profiles/rNN_pmc_b256_ddim100.json, newest round first) and is only reported for the
    configuration it was measured on.  The source is named in the line (`roofline.traffic_source`)."""
    if B != 256 or args.sampler != "ddim" or args.n_steps != 100:
        return None, None
    for name in ("r06_pmc_b256_ddim100.json", "r05_pmc_b256_ddim100.json", "r04_pmc_b256_ddim100.json", "r03_pmc_b256_ddim100.json", "r02_pmc_b256_ddim100.json", "r01_pmc_b256_ddim100.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                return round(json.load(f)["hbm_bytes_per_launch"]), "profiles/" + name + " (separate rocprofv3 --pmc passes of this command; not re-measured in this run)"
        except Exception:
            continue
    return None, None


def pmc_call_traffic(config, args, default_batch, default_steps, default_sampler):
    """(fabric bytes of ONE call of configs[2..4], source): from the committed counter passes of `bench.py --config C` (tools/r6/pmc_configs.sh ->
    profiles/r06_pmc_config<C>.json: FETCH_SIZE with the gfx950 2x correction + WRITE_SIZE over every kernel of a call), reported only for the
    configuration it was measured on.  These lines' `roofline.achieved` is per call too."""
    if args.batch != default_batch or args.n_steps != default_steps or args.sampler != default_sampler:
        return None, None
    name = f"r06_pmc_config{config}.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return round(json.load(f)["fabric_bytes_per_call"]), "profiles/" + name + " (separate rocprofv3 --pmc passes of this command, bytes per call; not re-measured in this run)"
    except Exception:
        return None, None


def train_line(args, rank=0, world=1, local=0):
    """One `agent, metrics = agent.update(batch, rng, step)` per step (train_bc.py:107) on rm_lift latent batches of 256 x 9 frames PER GPU,
    device-resident inputs.  Work: forward + data gradient + weight gradient of every GEMM-shaped layer = 3 x the forward FLOPs (flops.py), on the
    exact-fp32 MFMA.  --gpus N: dist.update_sharded, global batch 256 N (weak scaling), one RCCL all-reduce per module's gradient arena per step."""
    import time
    import numpy as np
    import torch
    if args.lib:
        from latent_diffusion_planning_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from latent_diffusion_planning_amd import flops, weights as W
    from latent_diffusion_planning_amd.dist import update_sharded
    from tests import cfgs
    from tests.util import idm_params, make_agent, planner_params
    dist = None
    if args.same_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    D, A, T = 25, 7, 8
    per_gpu = 256 if args.batch in (None, 256) or args.config != 1 else args.batch
    B = per_gpu * world
    steps = min(args.steps, 50)
    ag, data = make_agent("rm", planner_params(D=D), idm_params(D=D, A=A))
    for kv in args.opt:
        name, _, val = kv.partition("=")
        ag._engine.set_option(name, int(val))
    dev = [{"obs": {k: torch.tensor(v).cuda() for k, v in b["obs"].items()}, "actions": torch.tensor(b["actions"]).cuda()}
           for b in (cfgs.synth_latent_batch(data, B, T + 1, 40 + i, with_actions=True) for i in range(4))]
    step_fn = (lambda a, b, r, s: a.update(b, r, s)) if world == 1 else (lambda a, b, r, s: update_sharded(a, b, r, s))
    for i in range(max(args.warmup, 2)):
        ag, m = step_fn(ag, dev[i % 4], i, i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for i in range(steps):
        ag, m = step_fn(ag, dev[i % 4], 100 + i, 2 + i)
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = (time.time() - t0) / steps * 1e3
    ms = e0.elapsed_time(e1) / steps
    if dist is not None:
        tm = torch.tensor([wall, ms], dtype=torch.float64, device="cpu" if args.same_gpu else "cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall, ms = float(tm[0]), float(tm[1])
    loss, g_norm = float(m["loss"]), float(m["g_norm"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return None
    work = 3.0 * (flops.planner_forward_flops(W.PlannerSpec(D, D), T) * B + flops.idm_forward_flops(W.IDMSpec(D, A)) * B * T)
    t_step = max(ms, wall) if world > 1 else ms
    traffic = source = None                            # fabric bytes per step from the committed counter passes of the same step (tools/r6/pmc_train.sh)
    if per_gpu == 256 and world == 1 and not args.opt:
        try:
            with open(os.path.join(ROOT, "profiles", "r06_pmc_update.json")) as f:
                traffic = round(json.load(f)["fabric_bytes_per_step"])
            source = "profiles/r06_pmc_update.json (separate rocprofv3 --pmc passes of tools/r6/train_bench.py, bytes per step: reads = 2 x FETCH_SIZE KiB + WRITE_SIZE; not re-measured in this run)"
        except Exception:
            traffic = source = None
    return {"metric": f"training samples/sec (LDPAgent.update: planner + IDM, batch {per_gpu} per GPU, horizon 9)", "NOT_THE_DRIVER_LINE": True,
            "value": round(B / t_step * 1e3, 1), "unit": "samples/s", "n_gpus": world, "steps": steps, "ms_per_step": round(t_step, 3),
            "ms_per_step_events": round(ms, 3), "ms_per_step_wall": round(wall, 3), "scaling": "weak", "dtype": "f32",
            "data": "synthetic rm_lift latent batches (B, 9, 25) + actions (B, 9, 7), seeded init weights, explicit Philox noise",
            "config": {"workload": "train_bc.yaml:9 batch_size 256; agent/ldp_agent.py:223-272 update_step: jax.grad(loss) + global_norm + optax.adam for "
                                   "ConditionalUnet1D (65.6 M parameters) and MLPDiffusion (1.8 M)", "gflop_per_step": round(work / 1e9, 2),
                       "global_batch": B, "parallelism": f"dp{world}" + (" (gradient arenas: one all-reduce per module per step)" if world > 1 else "")},
            "roofline": {"bound": "mfma", "achieved": round(work / t_step / 1e9 / world, 2), "peak": flops.FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(work / t_step / 1e9 / world / flops.FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": source,
                         "kernel": "ldp::seg_gemm (forward / dgrad / wgrad of every Dense and convolution) + element-wise GroupNorm / LayerNorm / Adam kernels; "
                                   "3 x forward FLOPs over the step time, per GPU"},
            "loss": loss, "g_norm": g_norm}


def dry_run(args, rank, world):
    """Launcher + collective plumbing only (CPU, gloo)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from latent_diffusion_planning_amd.dist import all_gather_rows
    n = world * 4
    # what the configuration gathers: plans (configs 1, 4) or plans and actions (config 3)
    shapes = {1: [(4, 8, 25)], 2: [(4, 5, 25), (4, 4, 7)], 3: [(4, 5, 30), (4, 4, 14)], 4: [(4, 8, 25)]}[args.config]
    ok = True
    for shp in shapes:
        mine = torch.full(shp, float(rank))
        full = all_gather_rows(mine, n) if world > 1 else mine
        ok = ok and all(bool((full[r * 4:(r + 1) * 4] == float(r)).all()) for r in range(world))
    seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": f"latent plans/sec (horizon=9, {args.n_steps} {args.sampler.upper()} steps)", "value": 0.0, "unit": "plans/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "INVALID: --dry-run (launcher and collective check on CPU/gloo, no GPU work)",
                          "config": {"workload": "dry run", "baseline_config": args.config, "plans_per_gpu": args.batch,
                                     "denoise_steps": args.n_steps, "sampler": args.sampler, "ranks_seen_by_backend": seen, "gather_ok": ok,
                                     "backend": "gloo", "parallelism": f"dp{world}"}}), flush=True)
    return 0 if ok and seen == world else 1


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N does it by itself)")
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))
    if args.configs0:
        print(json.dumps(configs0(), indent=1), flush=True)
        return
    if args.train:
        line = train_line(args, rank, world, local)
        if line is not None:
            print(json.dumps(line), flush=True)
        return

    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.same_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.lib:
        from latent_diffusion_planning_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from latent_diffusion_planning_amd import flops

    wl = {1: PlannerWorkload, 2: JointT16Workload, 3: AlohaWorkload, 4: CandidatesWorkload}[args.config](args, rank, world, dev)
    eng = wl.eng
    if args.same_gpu and world > 1:
        # ranks time-sharing ONE GPU cannot keep each other's split work-groups co-resident: the exchange-free plans from the start
        # instead of a peer time-out and a recompute per call (191 s for 3 calls of configs[2] at 48 plans before this)
        eng.set_option("safe_mode", 1)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        eng.set_option(name, int(val or 1))
    stream = torch.cuda.Stream(device=dev)
    gathered = {}

    def one_step(i):
        # the i-th batch of plans: new seed, rows keyed by global plan index.  The all-gather is a synchronous torch
        # collective on the launch stream: it orders the ranks by itself (no barrier inside the timed region), and the next
        # planner graph (whose split work-groups need the whole chip at <= 256 plans) never overlaps the RCCL kernel.
        outs = wl.step(i)
        if world > 1:
            for j, t in enumerate(outs):
                buf = gathered.get(j)
                if buf is None or buf.shape[1:] != t.shape[1:]:
                    buf = gathered[j] = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                dist.all_gather_into_tensor(buf, t.contiguous())
            return [gathered[j] for j in range(len(outs))]
        return outs

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(n_steps_, first):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(n_steps_):
            last = one_step(first + i)
        ev1.record(stream)
        fence()
        dt = time.perf_counter() - t0
        dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        return float(dt_t.item()), ev0.elapsed_time(ev1), last

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            one_step(i)
        fence()
        dt_max, ev_ms, last = timed(args.steps, args.warmup)
        extra = wl.extra_timed(timed, args) if hasattr(wl, "extra_timed") else {}
    assert all(bool(torch.isfinite(t).all()) for t in last)
    eng.check_fault()                 # a fault of either kind (exchange time-out, fp16-plane range) would show here
    conv_launches, all_launches = eng.launch_counts()

    ablation = eng.active_debug_options()
    if args.lib:
        ablation = (ablation + " " if ablation else "") + f"--lib {args.lib}"
    if args.same_gpu:
        ablation = (ablation + " " if ablation else "") + "--same-gpu (ranks time-share one GPU, gloo, exchange-free plans: safe_mode)"
    if rank == 0:
        B = args.batch
        plans = world * B * args.steps
        info = wl.describe(conv_launches, ev_ms, args)
        line = {
            "metric": info["metric"],
            "value": round(plans / dt_max, 2),
            "unit": "plans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": info["dtype"],
            "data": "synthetic" if not ablation else "INVALID: ablation switch " + ablation + " set (timings only, results wrong)",
            "config": {"workload": info["workload"] + (", RCCL all-gather of " + info["gathered"] if world > 1 else ""),
                       "baseline_config": args.config,
                       "plans_per_gpu": B, "denoise_steps": args.n_steps, "sampler": args.sampler,
                       "graph": not args.no_graph, "parallelism": f"dp{world}",
                       "ranks_seen_by_backend": dist.get_world_size() if world > 1 else 1,
                       "backend": ("gloo (--same-gpu)" if args.same_gpu else "nccl (RCCL)") if world > 1 else "none",
                       "range_fallback": eng.get_option("range_fallback"), "fp16_plane_launches": eng.get_option("stat_f16_launches"),
                       **info.get("config", {}), **extra},
            "roofline": info["roofline"],
        }
        if not args.no_cpu_baseline and world == 1:          # CPU leg: rank 0 at N=1 only
            line["cpu_baseline"] = wl.cpu_baseline(args)
        print(json.dumps(line), flush=True)
    wl.close()
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# the four BASELINE.json workloads
# ------------------------------------------------------------------------------------------------
SPLIT_DTYPE = ("f32: above 256 plans the k=5 / stride-2 / transposed convs of the 256/512/1024-channel levels (and the StableVAE's 64/32/16/8-pixel "
               "3x3 convs, stride 2 included) run on 2xfp16 split operands, 3 exact products, f32 accumulate (x = h + l' / 2^11: 22 significand bits, range-guarded: "
               "|x| >= 65504 falls back to 3xbf16 planes / exact fp32), and so do the IDM's MLPResNet blocks; first conv and 1x1 convs: "
               "exact-fp32 MFMA; up to 256 plans everything is exact fp32")


class PlannerWorkload:
    """configs[1]: the rm_lift planner loop alone (the driver's line).  Exact-fp32 MFMA at <= 256 plans."""
    D, A, T, ah = 25, 7, 8, 4
    name = "rm_lift"

    def __init__(self, args, rank, world, dev):
        import numpy as np
        import torch
        from latent_diffusion_planning_amd import weights as W
        from latent_diffusion_planning_amd.engine import HipEngine
        self.args, self.rank, self.dev = args, rank, dev
        self.spec = W.PlannerSpec(self.D, self.D)
        self.pp = W.init_planner_params(self.spec, 0)                       # random-init weights of the named architecture
        self.eng = HipEngine(obs_dim=self.D, action_dim=self.A, global_cond_dim=self.D, pred_horizon=self.T,
                             action_horizon=self.ah, device=dev)
        self.eng.load_params(planner=self.pp)
        g = np.random.Generator(np.random.PCG64(1234 + rank))
        self.cond = torch.tensor(g.uniform(-1, 1, (args.batch, self.D)), dtype=torch.float32, device=dev)

    def step(self, i, cond=None):
        a = self.args
        return [self.eng.plan_sample(self.cond if cond is None else cond, seed=1000 + i, row_offset=self.rank * a.batch,
                                     sampler=a.sampler, n_steps=a.n_steps, use_graph=not a.no_graph)]

    def describe(self, conv_launches, ev_ms, args):
        from latent_diffusion_planning_amd import flops
        B = args.batch
        fwd = flops.planner_forward_flops(self.spec, self.T)                # per plan per denoising step
        # dominant kernel: tconv_kernel (30 fused conv launches per U-Net evaluation).  Per launch: algorithmic FLOPs of one
        # evaluation of the batch / 30, over the HIP-event time of the timed region on the launch stream divided by the
        # number of conv launches (gaps included).
        launches = conv_launches * args.steps
        avg_ms = ev_ms / max(launches, 1)
        per_launch = fwd * B * args.n_steps / max(conv_launches, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e12
        split = self.eng.get_option("stat_f16_launches") > 0
        peak = 2500.0 / 3 if split else flops.FP32_MFMA_PEAK_TFLOPS
        traffic, traffic_src = pmc_traffic(B, args) if args.config == 1 else (None, None)
        if args.config == 4:                  # per conv launch like `achieved`: the call's fabric bytes over its conv launches (99.7 % of them are tconv's)
            per_call, traffic_src = pmc_call_traffic(4, args, 1024, 50, "ddim")
            traffic = None if per_call is None else round(per_call / max(conv_launches, 1))
        return {
            "metric": f"latent plans/sec (horizon=9, {args.n_steps} {args.sampler.upper()} steps)",
            "dtype": SPLIT_DTYPE if split else "f32",
            "workload": f"configs[{args.config}]: {self.name} planner ConditionalUnet1D (D={self.D}, T={self.T}, down_dims [256,512,1024]), "
                        f"{args.n_steps}-step {args.sampler.upper()}, batch {B} synthetic latents per GPU, random-init weights, Philox noise, "
                        "hipGraph-captured loop",
            "gathered": "plans",
            "config": {"algorithmic_gflop_per_forward": round(fwd / 1e9, 5), "survey_gflop_per_forward": 0.16349,
                       # timestep-only work (time MLP, FiLM Dense) is hoisted into tables at finalize: FLOPs the
                       # loop actually executes per plan per step (SURVEY 8d asks for this disclosure)
                       "executed_gflop_per_forward": round(flops.planner_forward_flops(self.spec, self.T, hoisted=True) / 1e9, 5)},
            "roofline": {"bound": "mfma-f16x3" if split else "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "frac_of_fp32_mfma_peak": round(achieved / flops.FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "ldp::tconv_kernel",
                         "launches_per_step": conv_launches, "avg_launch_us": round(avg_ms * 1e3, 3),
                         "gflop_per_launch": round(per_launch / 1e9, 4)},
        }

    def cpu_baseline(self, args):
        return cpu_baseline(self.pp, self.D, self.T, args.sampler, args.n_steps)

    def close(self):
        self.eng.close()


class CandidatesWorkload(PlannerWorkload):
    """configs[4]: rm_can best-of-N candidate generation -- the planner loop, 50-step DDIM, 1024 candidates per GPU.  The reference has no
    scoring step: the configuration is "generate N candidates + gather" (SURVEY 8d).  `value` is timed with N independent observations;
    the same steps with ONE observation broadcast to every candidate are timed right behind (config.shared_cond_*)."""
    name = "rm_can"

    def extra_timed(self, timed, args):
        shared = self.cond[:1].expand(args.batch, self.D).contiguous()
        keep, self.cond = self.cond, shared
        try:
            self.step(0)
            dt, _, _ = timed(args.steps, 100000)
        finally:
            self.cond = keep
        world = int(os.environ.get("WORLD_SIZE", "1"))
        return {"candidates_total": world * args.batch, "cond": "N independent observations (value); one observation broadcast (shared_cond_*)",
                "shared_cond_plans_per_s": round(world * args.batch * args.steps / dt, 2), "shared_cond_ms_per_step": round(dt / args.steps * 1e3, 3)}


class JointT16Workload:
    """configs[2]: rm_square, pred_horizon 16 (the literal 15 of the yaml is ill-formed for the reference's U-Net: SURVEY 8d), planner + plan assembly +
    IDM as ONE captured graph (ldp_agent_sample), DDPM-100, 1024 plans per GPU, device-resident observation embeddings."""

    def __init__(self, args, rank, world, dev):
        import numpy as np
        import torch
        from latent_diffusion_planning_amd import weights as W
        from latent_diffusion_planning_amd.engine import HipEngine
        self.args, self.rank, self.dev = args, rank, dev
        self.D, self.A, self.T = 25, 7, 16
        self.pspec, self.ispec = W.PlannerSpec(self.D, self.D), W.IDMSpec(self.D, self.A)
        self.eng = HipEngine(obs_dim=self.D, action_dim=self.A, global_cond_dim=self.D, pred_horizon=self.T, action_horizon=4, device=dev)
        self.eng.load_params(planner=W.init_planner_params(self.pspec, 0), idm=W.init_idm_params(self.ispec, 1))
        g = np.random.Generator(np.random.PCG64(2222 + rank))
        self.obs = torch.tensor(g.uniform(-1, 1, (args.batch, 1, self.D)).astype(np.float32), device=dev)

    def step(self, i):
        a = self.args
        x, plan, act = self.eng.agent_sample(self.obs, 1, seed=1000 + i, row_offset=self.rank * a.batch, sampler=a.sampler,
                                             planner_steps=a.n_steps, idm_steps=a.n_steps, use_graph=not a.no_graph)
        return [plan, act]

    def describe(self, conv_launches, ev_ms, args):
        from latent_diffusion_planning_amd import flops
        B = args.batch
        per_plan = flops.planner_forward_flops(self.pspec, self.T) * args.n_steps + flops.idm_forward_flops(self.ispec) * 4 * args.n_steps
        achieved = per_plan * B * args.steps / (ev_ms * 1e-3) / 1e12
        peak = 2500.0 / 3
        return {
            "metric": f"latent plans/sec (horizon=17 planner + IDM, one graph, {args.n_steps} {args.sampler.upper()} steps)",
            "dtype": SPLIT_DTYPE if B > 256 else "f32",
            "workload": f"configs[2]: rm_square planner ConditionalUnet1D (D={self.D}, T={self.T}) + plan assembly + MLPDiffusion IDM (A={self.A}, 4 rows per plan), "
                        f"{B} plans per GPU, {args.n_steps}-step {args.sampler.upper()} each, ONE hipGraph per call (ldp_agent_sample), random-init weights, Philox noise",
            "gathered": "plans and actions",
            "config": {"algorithmic_gflop_per_plan": round(per_plan / 1e9, 3)},
            "roofline": {"bound": "mfma-f16x3" if B > 256 else "mfma", "achieved": round(achieved, 2),
                         "peak": round(peak, 1) if B > 256 else flops.FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / (peak if B > 256 else flops.FP32_MFMA_PEAK_TFLOPS), 4),
                         "frac_of_fp32_mfma_peak": round(achieved / flops.FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_call_traffic(2, args, 1024, 100, "ddpm")[0], "traffic_source": pmc_call_traffic(2, args, 1024, 100, "ddpm")[1],
                         "kernel": "whole call: ldp::tconv_kernel (planner) + ldp::idm_block_h16_kernel / idm_block_kernel; algorithmic fp32 FLOPs of the call over its HIP-event time",
                         "planner_conv_launches_per_step": conv_launches},
        }

    def cpu_baseline(self, args, budget_s=30.0):
        """The same per-plan work on the torch-CPU restatement: 5 of the 100 steps of each loop at B = 64, scaled."""
        import numpy as np
        import torch
        from latent_diffusion_planning_amd import weights as W
        from oracle import torch32
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        g = np.random.Generator(np.random.PCG64(6))
        PP = torch32.TorchParams(W.init_planner_params(self.pspec, 0))
        PI = torch32.TorchParams(W.init_idm_params(self.ispec, 1))
        B, s = 64, 5
        cond = torch.tensor(g.uniform(-1, 1, (B, self.D)), dtype=torch.float32)
        x0 = torch.tensor(g.standard_normal((B, self.T, self.D)), dtype=torch.float32)
        xn = torch.tensor(g.standard_normal((100, B, self.T, self.D)), dtype=torch.float32)
        tr = torch.tensor(g.uniform(-1, 1, (B * 4, 2 * self.D)), dtype=torch.float32)
        a0 = torch.tensor(g.standard_normal((B * 4, self.A)), dtype=torch.float32)
        an = torch.tensor(g.standard_normal((100, B * 4, self.A)), dtype=torch.float32)
        torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=100, sampler="ddpm", stop_after=1)
        t0 = time.perf_counter(); torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=100, sampler="ddpm", stop_after=s)
        t_pl = (time.perf_counter() - t0) * 100 / s / B
        t0 = time.perf_counter(); torch32.idm_sample(PI, tr, a0, an, n_train=100, n_steps=100, sampler="ddpm", stop_after=s)
        t_id = (time.perf_counter() - t0) * 100 / s / B
        return {"value": round(1.0 / (t_pl + t_id), 4), "unit": "plans/s", "cores": int(torch.get_num_threads()), "kind": "port",
                "sample": f"oracle/torch32.py (fp32 torch-CPU): {s} of the 100 DDPM steps of the T = 16 planner ({t_pl * 1e3:.1f} ms / plan scaled) and of the IDM "
                          f"({t_id * 1e3:.2f} ms / plan scaled) at B = {B}; proxy for the JAX-CPU reference (JAX is not installable here)"}

    def close(self):
        self.eng.close()


class AlohaWorkload:
    """configs[3]: aloha sim_transfer_cube through LDPAgent.sample -- raw 64x64 wrist frames [0, 255] -> normalise -> StableVAE encode ->
    latent normalise -> DDPM-100 planner -> plan assembly -> DDPM-100 IDM -> action un-normalisation, device-resident inputs, 512 frames per GPU."""

    def __init__(self, args, rank, world, dev):
        import numpy as np
        import torch
        from latent_diffusion_planning_amd import weights as W
        from latent_diffusion_planning_amd.agent import LDPAgent
        from tests import cfgs
        self.args, self.rank, self.dev = args, rank, dev
        data = cfgs.ALOHA_CUBE
        self.ag = LDPAgent.create(0, None, data["shape_meta"], vae_params=W.init_vae_params(seed=2, decoder=False), device=dev,
                                  **cfgs.agent_kwargs(data))
        self.eng = self.ag._engine
        B = args.batch
        g = np.random.Generator(np.random.PCG64(4321 + rank))
        low = cfgs.synth_latent_batch(data, B, 1, 3 + rank)["obs"]
        obs = {k: torch.tensor(v, device=dev) for k, v in low.items() if not k.startswith("latent_")}
        obs["wrist64_image"] = torch.tensor(g.integers(0, 256, (B, 1, 64, 64, 3)).astype(np.float32), device=dev)
        self.batch = {"obs": obs}
        self.pspec, self.ispec = W.PlannerSpec(30, 30), W.IDMSpec(30, 14)

    def step(self, i):
        a = self.args
        act, met = self.ag.sample(self.batch, 1000 + i, row_offset=self.rank * a.batch, sampler=a.sampler,
                                  n_steps=None if a.sampler == "ddpm" else a.n_steps)
        # the rows are about to leave the rank: the call's completion point (stream sync + fault poll; a faulted call is recomputed)
        act.complete()
        return [met["plan"].tensor, act.tensor]

    def describe(self, conv_launches, ev_ms, args):
        from latent_diffusion_planning_amd import flops
        B = args.batch
        per_plan = (flops.planner_forward_flops(self.pspec, 8) * args.n_steps + flops.idm_forward_flops(self.ispec) * 4 * args.n_steps + 10.988e9)
        achieved = per_plan * B * args.steps / (ev_ms * 1e-3) / 1e12
        peak = 2500.0 / 3
        return {
            "metric": f"latent plans/sec (aloha: StableVAE 64x64 encode + horizon=9 planner + IDM, {args.n_steps} {args.sampler.upper()} steps)",
            "dtype": SPLIT_DTYPE,
            "workload": f"configs[3]: aloha sim_transfer_cube (D=30, A=14, T=8): {B} raw 64x64 wrist frames per GPU -> StableVAE encode -> "
                        f"{args.n_steps}-step {args.sampler.upper()} planner -> plan assembly -> {args.n_steps}-step IDM (LDPAgent.sample, device-resident "
                        "inputs, one hipGraph for the two loops), random-init weights, Philox noise",
            "gathered": "plans and actions",
            "config": {"algorithmic_gflop_per_plan": round(per_plan / 1e9, 3)},
            "roofline": {"bound": "mfma-f16x3", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "frac_of_fp32_mfma_peak": round(achieved / flops.FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_call_traffic(3, args, 512, 100, "ddpm")[0], "traffic_source": pmc_call_traffic(3, args, 512, 100, "ddpm")[1],
                         "kernel": "whole call: ldp::tconv_kernel (planner and the StableVAE's stride-2 / 8-pixel convs, split tiles) + ldp::sconv3_kernel (StableVAE) + "
                                                    "ldp::idm_block_h16_kernel; algorithmic fp32 FLOPs of the call over its HIP-event time",
                         "planner_conv_launches_per_step": conv_launches},
        }

    def cpu_baseline(self, args, budget_s=30.0):
        """The same per-plan work on the torch-CPU restatement: 8 frames encoded, 10 of the 100 steps of each loop at B = 256, scaled."""
        import numpy as np
        import torch
        from latent_diffusion_planning_amd import weights as W
        from oracle import torch32
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        g = np.random.Generator(np.random.PCG64(5))
        PV = torch32.TorchParams(W.init_vae_params(seed=2, decoder=False))
        PP = torch32.TorchParams(W.init_planner_params(self.pspec, 0))
        PI = torch32.TorchParams(W.init_idm_params(self.ispec, 1))
        img = torch.tensor(g.uniform(-1, 1, (8, 64, 64, 3)), dtype=torch.float32)
        torch32.vae_encode_mean(PV, img[:1])
        t0 = time.perf_counter(); torch32.vae_encode_mean(PV, img); t_enc = (time.perf_counter() - t0) / 8
        B, s = 256, 10
        cond = torch.tensor(g.uniform(-1, 1, (B, 30)), dtype=torch.float32)
        x0 = torch.tensor(g.standard_normal((B, 8, 30)), dtype=torch.float32)
        xn = torch.tensor(g.standard_normal((100, B, 8, 30)), dtype=torch.float32)
        tr = torch.tensor(g.uniform(-1, 1, (B * 4, 60)), dtype=torch.float32)
        a0 = torch.tensor(g.standard_normal((B * 4, 14)), dtype=torch.float32)
        an = torch.tensor(g.standard_normal((100, B * 4, 14)), dtype=torch.float32)
        torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=100, sampler="ddpm", stop_after=1)
        t0 = time.perf_counter(); torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=100, sampler="ddpm", stop_after=s)
        t_pl = (time.perf_counter() - t0) * 100 / s / B
        t0 = time.perf_counter(); torch32.idm_sample(PI, tr, a0, an, n_train=100, n_steps=100, sampler="ddpm", stop_after=s)
        t_id = (time.perf_counter() - t0) * 100 / s / B
        per_plan = t_enc + t_pl + t_id
        return {"value": round(1.0 / per_plan, 4), "unit": "plans/s", "cores": int(torch.get_num_threads()), "kind": "port",
                "sample": f"oracle/torch32.py (fp32 torch-CPU): StableVAE encode of 8 frames ({t_enc * 1e3:.0f} ms / frame) + {s} of the 100 DDPM steps of the "
                          f"planner ({t_pl * 1e3:.1f} ms / plan scaled) and of the IDM ({t_id * 1e3:.2f} ms / plan scaled) at B = {B}; proxy for the JAX-CPU "
                          "reference (JAX is not installable here)"}

    def close(self):
        self.eng.close()


if __name__ == "__main__":
    main()
