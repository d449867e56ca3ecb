#!/usr/bin/env python3
"""Which tconv instantiations do the DEFAULT regimes (and the bf16-plane fallback the range guard switches to) launch?  Runs the planner for
pred_horizon 8 / 16 and the hierarchical agent's two-level U-Net over the batch regimes, on fp16 planes, bf16 planes and exact fp32, and prints
the union (option dump_plans).  What is not in the list is not needed by any default path: profiles/r05_plans_used.txt decided the trim of
csrc/tconv_split*.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine

g = np.random.Generator(np.random.PCG64(0))
for T, D, down in ((8, 25, (256, 512, 1024)), (16, 25, (256, 512, 1024)), (4, 7, (256, 512))):
    G = D if len(down) == 3 else 50
    pp = W.init_planner_params(W.PlannerSpec(D, G, down_dims=down), 0)
    for opts in ({}, {"planner_split_f16": 0}, {"planner_split": 0}, {"safe_mode": 1}, {"safe_mode": 1, "planner_split_f16": 0}):
        e = HipEngine(obs_dim=D, action_dim=7, global_cond_dim=G, pred_horizon=T, action_horizon=min(4, T), down_dims=down)
        e.load_params(planner=pp)
        for k, v in opts.items():
            e.set_option(k, v)
        for B in (1, 5, 16, 64, 128, 200, 256, 300, 352, 400, 512, 600, 992, 1024, 2048, 4096):
            x = torch.tensor(g.standard_normal((B, T, D)), dtype=torch.float32)
            c = torch.tensor(g.uniform(-1, 1, (B, G)), dtype=torch.float32)
            e.unet_forward(x, 3, c)
        torch.cuda.synchronize()
        e.check_fault()
        e.set_option("dump_plans", 1)
        e.close()
    # tests: the split tiles at the goldens' own batch (planner_split = 2: wherever the plan has no column / K split)
    for f16 in (1, 0):
        e = HipEngine(obs_dim=D, action_dim=7, global_cond_dim=G, pred_horizon=T, action_horizon=min(4, T), down_dims=down)
        for k, v in (("planner_split", 2), ("planner_split_f16", f16), ("no_csplit", 1), ("no_kw", 1)):
            e.set_option(k, v)
        e.load_params(planner=pp)
        for B in (3, 40, 300):
            e.unet_forward(torch.tensor(g.standard_normal((B, T, D)), dtype=torch.float32), 3, torch.tensor(g.uniform(-1, 1, (B, G)), dtype=torch.float32))
        torch.cuda.synchronize()
        e.check_fault()
        e.set_option("dump_plans", 1)
        e.close()
