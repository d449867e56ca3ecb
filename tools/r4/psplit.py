#!/usr/bin/env python3
"""Same-box A/B of the planner's split-operand layers (option planner_split): psplit.py [T] [B] [sampler] [steps] [split value: 1 | 2 = also below 512 plans]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import flops, weights as W, _lib
if os.environ.get("PSPLIT_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["PSPLIT_LIB"])      # e.g. the -DLDP_ABLATE build
from latent_diffusion_planning_amd.engine import HipEngine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
smp = sys.argv[3] if len(sys.argv) > 3 else "ddim"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 50
sv = int(sys.argv[5]) if len(sys.argv) > 5 else 1
opts = dict(kv.split("=") for kv in sys.argv[6:])          # further engine options for the split run: name=value
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
g = np.random.Generator(np.random.PCG64(1))
cond_h, x0_h = g.uniform(-1, 1, (B, 25)), g.standard_normal((B, T, 25))
fl = flops.planner_forward_flops(W.PlannerSpec(25, 25), T) * n * B
outs = {}
for rep in range(2):
    for split in ((sv,) if os.environ.get("PSPLIT_ONLY") else (0, sv)):
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        e.set_option("planner_split", split)
        if split:
            for k_, v_ in opts.items(): e.set_option(k_, int(v_))
        e.load_params(planner=pp)
        cond = torch.tensor(cond_h, dtype=torch.float32, device="cuda"); x0 = torch.tensor(x0_h, dtype=torch.float32, device="cuda")
        outs[split] = e.plan_sample(cond, x_init=x0, sampler="ddim", n_steps=n).cpu().numpy()
        for i in range(2): e.plan_sample(cond, seed=i, sampler=smp, n_steps=n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5): e.plan_sample(cond, seed=10 + i, sampler=smp, n_steps=n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        e.check_fault()
        print(f"T={T} B={B} {smp}-{n} planner_split={split}: {dt * 1e3:.2f} ms = {B / dt:.0f} plans/s = {fl / dt / 1e12:.1f} TF/s ({fl / dt / 157.3e12:.3f} of the fp32 MFMA peak)", flush=True)
        e.close()
if 0 in outs: print(f"max |plan(split) - plan(exact fp32)| over {B} plans (DDIM-{n}, same x_T) = {np.abs(outs[sv] - outs[0]).max():.2e}")
