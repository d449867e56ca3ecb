#!/usr/bin/env python3
"""The nine-product (exact-product) form of the 16-row split tiles, and the six-product form, at configs[1]'s 256 plans (option planner_split = 3:
16-row tiles also under the 2-way column split) and at 1024 plans, against the exact-fp32 kernels; goldens tiled to 256 plans as the parity check.
nine.py LIB   (LIB built with -DLDP_SPLIT_NPROD=9, or the product library for the six-product form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import flops, weights as W, _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from latent_diffusion_planning_amd.engine import HipEngine
from tests.cases import load_case
from tests.util import planner_params
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
print(_lib.load().ldp_version().decode() if hasattr(_lib.load(), "ldp_version") else sys.argv[1])
for name, smp, n in (("planner_loop_ddpm100", "ddpm", 100), ("planner_loop_ddim50", "ddim", 50)):
    inp, exp = load_case(name)
    idx = np.arange(256) % inp["cond"].shape[0]
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
    e.set_option("planner_split", 3)
    e.load_params(planner=planner_params())          # the weights the goldens were made with
    got = e.plan_sample(f(inp["cond"][idx]), x_init=f(inp["x0"][idx]), step_noise=f(inp["nz"][:, idx]) if smp == "ddpm" else None, sampler=smp, n_steps=n).cpu().numpy()
    e.check_fault(); e.close()
    print(f"{name} tiled to 256 plans, planner_split=3: max|err| vs the float64 golden {np.abs(got - exp['plan'][idx]).max():.2e} (tolerance 1e-4)")
g = np.random.Generator(np.random.PCG64(1))
for T, B, n in ((8, 256, 100), (8, 1024, 50), (16, 1024, 50)):
    cond = f(g.uniform(-1, 1, (B, 25))).cuda(); x0 = f(g.standard_normal((B, T, 25))).cuda()
    fl = flops.planner_forward_flops(W.PlannerSpec(25, 25), T) * n * B
    outs = {}
    for rep in range(2):
        for sp in (0, 3):
            e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
            e.set_option("planner_split", sp)
            e.load_params(planner=pp)
            outs[sp] = e.plan_sample(cond, x_init=x0, sampler="ddim", n_steps=n).cpu().numpy()
            for i in range(2): e.plan_sample(cond, seed=i, sampler="ddim", n_steps=n)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(5): e.plan_sample(cond, seed=10 + i, sampler="ddim", n_steps=n)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            e.check_fault(); e.close()
            print(f"T={T} B={B} ddim-{n} planner_split={sp}: {dt * 1e3:.2f} ms = {B / dt:.0f} plans/s ({fl / dt / 157.3e12:.3f} of the fp32 MFMA peak)", flush=True)
    print(f"  max |split - exact fp32| = {np.abs(outs[3] - outs[0]).max():.2e}")
