#!/usr/bin/env python3
"""A/B of the first conv's virtual input chunk (option first_k, read at finalize): first_k.py [--lib P] K [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
args = sys.argv[1:]
if "--lib" in args:
    from latent_diffusion_planning_amd import _lib
    i = args.index("--lib"); _lib.LIB_PATH = os.path.abspath(args[i + 1]); del args[i:i + 2]
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
g = np.random.Generator(np.random.PCG64(1))
cond_h, x0_h = g.uniform(-1, 1, (256, 25)), g.standard_normal((256, 8, 25))
res = {}
for rep in range(2):
    for k in [int(a) for a in args]:
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
        e.set_option("first_k", k)
        e.load_params(planner=pp)
        cond = torch.tensor(cond_h, dtype=torch.float32, device="cuda")
        x0 = torch.tensor(x0_h, dtype=torch.float32, device="cuda")
        out = e.plan_sample(cond, x_init=x0, sampler="ddim", n_steps=100)
        for i in range(3): e.plan_sample(cond, seed=i, sampler="ddim", n_steps=100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): e.plan_sample(cond, seed=10 + i, sampler="ddim", n_steps=100)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        res.setdefault(k, []).append((dt, out.cpu().numpy()))
        print(f"first_k={k}: {dt * 1e3:.3f} ms per 256 plans = {256 / dt:.0f} plans/s", flush=True)
        e.close()
ks = list(res)
for k in ks[1:]:
    print(f"max |plan(first_k={k}) - plan(first_k={ks[0]})| = {np.abs(res[k][0][1] - res[ks[0]][0][1]).max():.2e}")
