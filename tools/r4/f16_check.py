#!/usr/bin/env python3
"""The 16-row split tiles on two fp16 planes / three products (option planner_split_f16) against the six-product bf16 form and the exact-fp32
kernels: one evaluation and a DDIM loop at 1043 / 600 plans, both horizons; and against the float64 goldens tiled to 512 / 1024 plans."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
from tests.cases import load_case
from tests.util import planner_params
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
for T in (16, 8):
    for B in (1043, 600):
        g = np.random.Generator(np.random.PCG64(T + B))
        cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32); x = torch.tensor(g.standard_normal((B, T, 25)), dtype=torch.float32)
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        e.load_params(planner=pp); e.set_option("no_batch_split", 1)
        e.set_option("planner_split", 0); ref = e.unet_forward(x, 17, cond); lref = e.plan_sample(cond, seed=5, sampler="ddim", n_steps=20)
        e.set_option("planner_split", 1)
        for v in (0, 1):
            e.set_option("planner_split_f16", v); n0 = e.get_option("stat_f16_launches")
            o = e.unet_forward(x, 17, cond); ran = e.get_option("stat_f16_launches") - n0
            l = e.plan_sample(cond, seed=5, sampler="ddim", n_steps=20)
            print(f"T={T} B={B} f16={v}: {ran} fp16 launches; one evaluation vs exact fp32 {(o-ref).abs().max().item():.2e}; DDIM-20 loop vs exact fp32 {(l-lref).abs().max().item():.2e}", flush=True)
        e.check_fault(); e.close()
f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
for name, T, smp, n in (("planner_loop_ddpm100", 8, "ddpm", 100), ("planner_loop_ddim50", 8, "ddim", 50), ("planner_loop_t16_ddpm100", 16, "ddpm", 100)):
    inp, exp = load_case(name)
    for B in (512, 1024):
        idx = np.arange(B) % inp["cond"].shape[0]
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        e.load_params(planner=planner_params())
        r = {}
        for v in (0, 1):
            e.set_option("planner_split_f16", v)
            got = e.plan_sample(f(inp["cond"][idx]), x_init=f(inp["x0"][idx]), step_noise=f(inp["nz"][:, idx]) if smp == "ddpm" else None, sampler=smp, n_steps=n).cpu().numpy()
            r[v] = np.abs(got - exp["plan"][idx]).max()
        e.check_fault(); e.close()
        print(f"{name} x{B}: max|err| against the float64 golden: bf16 x 6 {r[0]:.2e}, fp16 x 3 {r[1]:.2e}", flush=True)
