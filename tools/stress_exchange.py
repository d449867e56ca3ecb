#!/usr/bin/env python3
"""Stress of the in-launch exchanges (column split granules, K split partial tiles): the same planner call repeated
with alternating seeds must be bit-identical to its first result every time (fixed summation orders).  A consumer that
ever read a peer's data before it was there would show up as a mismatch (the slabs hold the other seed's values).
usage: [T=16] stress_exchange.py [calls per batch size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=int(os.environ.get("T", "8")), action_horizon=4)
e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25), 0))
total_bad = 0
# 384 / 512: round 4's column-split 32-row split tiles (the T = 2 layers from 353 to 512 plans exchange their statistics in-launch too)
for B in (1, 5, 16, 48, 64, 128, 256, 384, 512):
    cond = torch.tensor(np.random.default_rng(B).uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
    for graph in (True, False):
        steps = 100 if B <= 256 else 25
        refs = [e.plan_sample(cond, seed=s, sampler="ddim", n_steps=steps, use_graph=graph).clone() for s in (11, 12, 13)]
        bad = 0
        for i in range(n if graph else max(n // 5, 4)):
            out = e.plan_sample(cond, seed=11 + i % 3, sampler="ddim", n_steps=steps, use_graph=graph)
            bad += 0 if torch.equal(out, refs[i % 3]) else 1
        total_bad += bad
        print(f"B={B:4d} graph={int(graph)} mismatching calls {bad}", flush=True)
e.check_fault()
print("TOTAL_MISMATCHES", total_bad)
