#!/usr/bin/env python3
"""A/B of a work-split option on the bench workload: max |diff| of the plans against the default launch plan and
ms per batch with and without it.   usage: exp_opt.py NAME[=V] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
name, _, val = sys.argv[1].partition("=")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25), 0))
cond = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
def run(n=10):
    e.plan_sample(cond, seed=3, sampler="ddim", n_steps=100)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): out = e.plan_sample(cond, seed=3, sampler="ddim", n_steps=100)
    torch.cuda.synchronize(); return out, (time.perf_counter() - t) / n * 1e3
a, ta = run()
e.set_option(name, int(val or 1))
b, tb = run()
e.check_fault()
print(f"{name}: default {ta:.2f} ms, with option {tb:.2f} ms ({(ta/tb-1)*100:+.1f} %), max|diff| {float((a-b).abs().max()):.2e}, finite {bool(torch.isfinite(b).all())}")
