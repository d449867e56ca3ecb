#!/usr/bin/env python3
"""bench.py -- latent plans/sec of the LDP planner denoising loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: with WORLD_SIZE unset and --gpus N > 1 this script launches the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`),
one process per GPU, backend "nccl" (= RCCL over xGMI); rank 0 prints the ONE JSON line.

One "step" = one full pass of the hot path over one batch: BASELINE.json configs[1], i.e. the
rm_lift planner ConditionalUnet1D (D=25, T=8), 100-step DDIM, batch 256 synthetic latents per
GPU, the whole loop replayed from one hipGraph.  Weak scaling: every rank samples its own 256
plans (independent Philox rows keyed by the global plan index) and the sampled trajectories are
all-gathered over RCCL inside the timed region.  Inputs are resident in HBM before the timed
region starts.

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
    ap.add_argument("--batch", type=int, default=256, help="plans per GPU")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--n-steps", type=int, default=100, help="denoising steps per plan")
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
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher/collective check on CPU (gloo); no GPU work, the line is INVALID")
    return ap.parse_args(argv)


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
def cpu_baseline(pp, D, T, sampler, n_steps, budget_s=40.0):
    """The SAME workload as the GPU metric of this line -- configs[1]: the rm_lift planner alone, `n_steps`-step
    `sampler` (DDIM-100 by default), synthetic latents -- on the torch-CPU restatement (oracle/torch32.py, fp32), at
    B in {1, 16, 256}, median of 3.  To keep the bench within minutes only `s` of the n_steps denoising steps are timed
    (every step costs the same: same network, same shapes) and the time is scaled by n_steps/s; `sample` says which s.
    (Rounds 1-2 timed SURVEY 8d's config 1 here -- DDPM planner + IDM -- which is not the GPU line's workload.)"""
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
    for B, s in ((1, 20), (16, 20), (256, 10)):
        s = min(s, n_steps)
        cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32)
        x0 = torch.tensor(g.standard_normal((B, T, D)), dtype=torch.float32)
        xn = torch.tensor(g.standard_normal((n_steps, B, T, D)), dtype=torch.float32) if sampler == "ddpm" else None

        def run(n):
            torch32.planner_sample(PP, cond, x0, xn, n_train=100, n_steps=n_steps, sampler=sampler, stop_after=n)
        run(1)                                                           # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(s)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and len(ts) >= 1:
                break
        per_call = statistics.median(ts) * n_steps / s
        rows[B] = dict(plans_per_s=round(B / per_call, 4), s_per_call=round(per_call, 3), steps_timed=s, runs=len(ts))
    best = max(rows.values(), key=lambda r: r["plans_per_s"])
    return {"value": best["plans_per_s"], "unit": "plans/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"the GPU line's own workload (rm_lift planner ConditionalUnet1D alone, {n_steps}-step {sampler.upper()}, "
                      "synthetic latents) on oracle/torch32.py, fp32 torch-CPU; "
                      "per B: " + "; ".join(f"B={b}: {r['plans_per_s']} plans/s ({r['s_per_call']} s per call, "
                                            f"{r['steps_timed']} of {n_steps} steps timed and scaled, median of {r['runs']})"
                                            for b, r in rows.items())
                      + "; value = best B; proxy for the JAX-CPU reference (JAX is not installable here)",
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
    for name in ("r04_pmc_b256_ddim100.json", "r03_pmc_b256_ddim100.json", "r02_pmc_b256_ddim100.json", "r01_pmc_b256_ddim100.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                return round(json.load(f)["hbm_bytes_per_launch"]), "profiles/" + name + " (separate rocprofv3 --pmc passes of this command; not re-measured in this run)"
        except Exception:
            continue
    return None, None


def dry_run(args, rank, world):
    """Launcher + collective plumbing only (CPU, gloo)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from latent_diffusion_planning_amd.dist import all_gather_rows
    n = world * 4
    mine = torch.full((4, 8, 25), float(rank))
    full = all_gather_rows(mine, n) if world > 1 else mine
    ok = all(bool((full[r * 4:(r + 1) * 4] == float(r)).all()) for r in range(world))
    seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "latent plans/sec (horizon=9, 100 DDIM steps)", "value": 0.0, "unit": "plans/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "INVALID: --dry-run (launcher and collective check on CPU/gloo, no GPU work)",
                          "config": {"workload": "dry run", "ranks_seen_by_backend": seen, "gather_ok": ok,
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
    from latent_diffusion_planning_amd import flops, weights as W
    from latent_diffusion_planning_amd.engine import HipEngine

    D, A, T, ah, B = 25, 7, 8, 4, args.batch
    spec = W.PlannerSpec(D, D)
    pp = W.init_planner_params(spec, 0)                       # random-init weights of the named architecture
    eng = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=ah, device=dev)
    eng.load_params(planner=pp)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        eng.set_option(name, int(val or 1))
    g = np.random.Generator(np.random.PCG64(1234 + rank))
    cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * B, T, D), dtype=torch.float32, device=dev) if world > 1 else None
    stream = torch.cuda.Stream(device=dev)

    def one_step(i):
        # the i-th batch of plans: new seed, rows keyed by global plan index.  The all-gather is a
        # synchronous torch collective: the launch stream waits for it, so the next planner graph
        # (whose split work-groups need the whole chip) never overlaps the RCCL kernel.
        out = eng.plan_sample(cond, seed=1000 + i, row_offset=rank * B, sampler=args.sampler,
                              n_steps=args.n_steps, use_graph=not args.no_graph)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)
            return gathered
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            one_step(i)
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(args.steps):
            last = one_step(args.warmup + i)
        ev1.record(stream)
        fence()
        dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    assert torch.isfinite(last).all()
    eng.check_fault()                 # a split work-group that timed out on its peer would show here
    conv_launches, all_launches = eng.launch_counts()

    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())

    ablation = eng.active_debug_options()
    if args.lib:
        ablation = (ablation + " " if ablation else "") + f"--lib {args.lib}"
    if args.same_gpu:
        ablation = (ablation + " " if ablation else "") + "--same-gpu (ranks time-share one GPU, gloo)"
    if rank == 0:
        plans = world * B * args.steps
        fwd_flops = flops.planner_forward_flops(spec, T)            # per plan per denoising step
        # dominant kernel: tconv_kernel (30 fused conv launches per U-Net evaluation).  Per launch:
        # algorithmic FLOPs of one evaluation of the batch / 30, over the HIP-event time of the timed
        # region on the launch stream divided by the number of conv launches (gaps included).
        traffic, traffic_src = pmc_traffic(B, args)
        launches = conv_launches * args.steps
        avg_launch_ms = ev_ms / max(launches, 1)
        flops_per_launch = fwd_flops * B * args.n_steps / max(conv_launches, 1)
        achieved = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
        line = {
            "metric": "latent plans/sec (horizon=9, 100 DDIM steps)",
            "value": round(plans / dt_max, 2),
            "unit": "plans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not ablation else "INVALID: ablation switch " + ablation + " set (timings only, results wrong)",
            "config": {"workload": "configs[1]: rm_lift planner ConditionalUnet1D (D=25, T=8, down_dims "
                                   f"[256,512,1024]), {args.n_steps}-step {args.sampler.upper()}, batch {B} synthetic "
                                   "latents per GPU, random-init weights, Philox noise, hipGraph-captured loop"
                                   + (", RCCL all-gather of plans" if world > 1 else ""),
                       "plans_per_gpu": B, "denoise_steps": args.n_steps, "sampler": args.sampler,
                       "graph": not args.no_graph, "parallelism": f"dp{world}",
                       "ranks_seen_by_backend": dist.get_world_size() if world > 1 else 1,
                       "backend": ("gloo (--same-gpu)" if args.same_gpu else "nccl (RCCL)") if world > 1 else "none",
                       "algorithmic_gflop_per_forward": round(fwd_flops / 1e9, 5),
                       "survey_gflop_per_forward": 0.16349,
                       # timestep-only work (time MLP, FiLM Dense) is hoisted into tables at finalize: FLOPs the
                       # loop actually executes per plan per step (SURVEY 8d asks for this disclosure)
                       "executed_gflop_per_forward": round(flops.planner_forward_flops(spec, T, hoisted=True) / 1e9, 5)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": flops.FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / flops.FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "ldp::tconv_kernel",
                         "launches_per_step": conv_launches,
                         "avg_launch_us": round(avg_launch_ms * 1e3, 3),
                         "gflop_per_launch": round(flops_per_launch / 1e9, 4)},
        }
        if not args.no_cpu_baseline and world == 1:          # CPU leg: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(pp, D, T, args.sampler, args.n_steps)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
