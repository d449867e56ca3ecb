#!/usr/bin/env python3
"""Round 6: time of one training step (LDPAgent.update: agent/ldp_agent.py:223-272) at the reference's batch size (train_bc.yaml:9: 256) on
synthetic latent batches, with the work it does.   python tools/r6/train_bench.py [--batch 256] [--steps 20] [--which both|planner|idm]
FLOPs: forward = flops.planner_flops_per_plan (0.16218 GFLOP at T = 8, D = 25) per plan and 3.572 MFLOP per IDM row; a training step is the forward,
the data gradient and the weight gradient of every GEMM-shaped layer = 3 x forward (the first layer's data gradient is not computed: -0.3 %)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd import flops              # noqa: E402
from tests import cfgs                                       # noqa: E402
from tests.util import idm_params, make_agent, planner_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--which", default="both", choices=["both", "planner", "idm"])
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE")
ap.add_argument("--lib", default=None, help="another build of libldp_hip.so (A/B)")
args = ap.parse_args()
if args.lib:
    from latent_diffusion_planning_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)

D, A, T = 25, 7, 8
ag, data = make_agent("rm", planner_params(D=D), idm_params(D=D, A=A))
for o in args.opt:
    k, v = o.split("=")
    ag._engine.set_option(k, int(v))
ag.use_planner, ag.use_idm = args.which in ("both", "planner"), args.which in ("both", "idm")
B = args.batch
batches = [cfgs.synth_latent_batch(data, B, T + 1, 40 + i, with_actions=True) for i in range(4)]
dev = [{"obs": {k: torch.tensor(v).cuda() for k, v in b["obs"].items()}, "actions": torch.tensor(b["actions"]).cuda()} for b in batches]
for i in range(args.warmup):
    ag, m = ag.update(dev[i % 4], i, i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
e0.record()
for i in range(args.steps):
    ag, m = ag.update(dev[i % 4], 100 + i, args.warmup + i)
e1.record()
host = (time.time() - t0) / args.steps * 1e3          # enqueue time per step (the host runs ahead of the GPU when this is below the step time)
torch.cuda.synchronize()
wall = (time.time() - t0) / args.steps * 1e3
ms = e0.elapsed_time(e1) / args.steps
from latent_diffusion_planning_amd import weights as W      # noqa: E402
fw_p = flops.planner_forward_flops(W.PlannerSpec(D, D), T)    # 0.16218 GFLOP per plan and evaluation at T = 8, D = 25
fw_i = flops.idm_forward_flops(W.IDMSpec(D, A))               # 3.572 MFLOP per row
work = 3.0 * ((fw_p * B if ag.use_planner else 0.0) + (fw_i * B * T if ag.use_idm else 0.0))
print(json.dumps({"what": f"LDPAgent.update, {args.which}", "batch": B, "rows_idm": B * T, "ms_per_step_gpu": round(ms, 3), "ms_per_step_wall": round(wall, 3), "ms_per_step_host_enqueue": round(host, 3),
                  "samples_per_s": round(B / ms * 1e3, 1), "gflop_per_step": round(work / 1e9, 2), "tflops": round(work / ms / 1e9, 2),
                  "frac_of_fp32_mfma_peak": round(work / ms / 1e9 / 157.3, 4), "loss": float(m["loss"]), "g_norm": float(m["g_norm"])}))
