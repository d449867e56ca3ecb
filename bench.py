#!/usr/bin/env python3
"""bench.py -- latent plans/sec of the LDP planner denoising loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch: BASELINE.json configs[1], i.e. the
rm_lift planner ConditionalUnet1D (D=25, T=8), 100-step DDIM, batch 256 synthetic latents per
GPU, the whole loop replayed from one hipGraph.  Weak scaling: every rank samples its own 256
plans (independent Philox rows keyed by the global plan index) and the sampled trajectories are
all-gathered over RCCL inside the timed region.  Inputs are resident in HBM before the timed
region starts.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def cpu_baseline(pp, D, T, sample_B=256, sample_steps=100, n_steps=100):
    """The CPU port of the same math (oracle/torch32.py, fp32, all host cores) on a bounded
    sample: `sample_steps` DDIM steps at batch `sample_B`, scaled to `n_steps` steps."""
    from oracle import torch32
    # torch's intra-op pool stops scaling (and then collapses) long before the 100+ hardware
    # threads of a GPU host on these small convolutions: use at most 32 and report that count.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    P = torch32.TorchParams(pp)
    g = np.random.Generator(np.random.PCG64(1))
    cond = torch.tensor(g.uniform(-1, 1, (sample_B, D)), dtype=torch.float32)
    x0 = torch.tensor(g.standard_normal((sample_B, T, D)), dtype=torch.float32)
    torch32.planner_sample(P, cond, x0, None, n_steps=2, sampler="ddim")          # warm-up
    t0 = time.perf_counter()
    torch32.planner_sample(P, cond, x0, None, n_steps=sample_steps, sampler="ddim")
    dt = time.perf_counter() - t0
    per_plan = dt / sample_steps * n_steps / sample_B
    return {"value": round(1.0 / per_plan, 4), "unit": "plans/s", "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": f"oracle/torch32.py planner loop, fp32 torch-CPU, B={sample_B}, {sample_steps} of "
                      f"{n_steps} DDIM steps timed ({dt:.1f} s) and scaled; proxy for the JAX-CPU reference "
                      "(JAX is not installable here)"}


def pmc_traffic(B, args):
    """HBM bytes per conv launch.  PMC counters cannot be read from inside the process: the number
    comes from the committed rocprofv3 --pmc passes of this very command (tools/pmc_passes.sh ->
    profiles/r01_pmc_b256_ddim100.json) and is only reported for the configuration it was measured on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_b256_ddim100.json")
    if B != 256 or args.sampler != "ddim" or args.n_steps != 100 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="plans per GPU")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--n-steps", type=int, default=100, help="denoising steps per plan")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from latent_diffusion_planning_amd import flops, weights as W
    from latent_diffusion_planning_amd.engine import HipEngine

    D, A, T, B = 25, 7, 8, args.batch
    spec = W.PlannerSpec(D, D)
    pp = W.init_planner_params(spec, 0)                       # random-init weights of the named architecture
    eng = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4, device=dev)
    eng.load_params(planner=pp)
    g = np.random.Generator(np.random.PCG64(1234 + rank))
    cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * B, T, D), dtype=torch.float32, device=dev) if world > 1 else None
    stream = torch.cuda.Stream(device=dev)

    def one_step(i):
        # the i-th batch of plans: new seed, rows keyed by global plan index
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
    eng.check_fault()                 # a column-split work-group that timed out on its peer would show here
    conv_launches, all_launches = eng.launch_counts()

    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())

    ablation = ",".join(k for k in ("LDP_DBG", "LDP_REPEAT") if os.environ.get(k))
    if rank == 0:
        plans = world * B * args.steps
        fwd_flops = flops.planner_forward_flops(spec, T)            # per plan per denoising step
        # dominant kernel: tconv_kernel (30 fused conv launches per U-Net evaluation).  Per launch:
        # algorithmic FLOPs of one evaluation of the batch / 30, over the HIP-event time of the timed
        # region on the launch stream divided by the number of conv launches (gaps included).
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
                       "algorithmic_gflop_per_forward": round(fwd_flops / 1e9, 5),
                       "survey_gflop_per_forward": 0.16349,
                       # timestep-only work (time MLP, FiLM Dense) is hoisted into tables at finalize: FLOPs the
                       # loop actually executes per plan per step (SURVEY 8d asks for this disclosure)
                       "executed_gflop_per_forward": round(flops.planner_forward_flops(spec, T, hoisted=True) / 1e9, 5)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": flops.FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / flops.FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic(B, args), "kernel": "ldp::tconv_kernel",
                         "launches_per_step": conv_launches,
                         "avg_launch_us": round(avg_launch_ms * 1e3, 3),
                         "gflop_per_launch": round(flops_per_launch / 1e9, 4)},
        }
        if not args.no_cpu_baseline and world == 1:          # CPU leg: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(pp, D, T, sample_B=B, sample_steps=args.n_steps,
                                                n_steps=args.n_steps)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
