#!/usr/bin/env python3
"""Same-box A/B of the 16-row split tiles over two row blocks per wave (option planner_split_mb2; tconv SPLIT = 2):
mb2.py [T] [B] [sampler] [steps] [values of planner_split_mb2 to compare, default 0 1 2].  Alternates the arms twice and
checks that the plans of every arm are bit-identical to arm 0's (same planes, same K order, same epilogue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import flops, weights as W, _lib
if os.environ.get("PSPLIT_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["PSPLIT_LIB"])
from latent_diffusion_planning_amd.engine import HipEngine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
smp = sys.argv[3] if len(sys.argv) > 3 else "ddim"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
arms = [a for a in sys.argv[5:]] or ["0", "1", "2"]       # "v" or "v,name=value,..." (further engine options of the arm)
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
g = np.random.Generator(np.random.PCG64(1))
cond_h, x0_h = g.uniform(-1, 1, (B, 25)), g.standard_normal((B, T, 25))
fl = flops.planner_forward_flops(W.PlannerSpec(25, 25), T) * n * B
outs, best = {}, {}
for rep in range(2):
    for arm in arms:
        v, *more = arm.split(",")
        e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
        e.set_option("planner_split_mb2", int(v))
        for kv in more: e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        e.load_params(planner=pp)
        cond = torch.tensor(cond_h, dtype=torch.float32, device="cuda"); x0 = torch.tensor(x0_h, dtype=torch.float32, device="cuda")
        outs[arm] = e.plan_sample(cond, x_init=x0, sampler="ddim", n_steps=n).cpu().numpy()
        for i in range(2): e.plan_sample(cond, seed=i, sampler=smp, n_steps=n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5): e.plan_sample(cond, seed=10 + i, sampler=smp, n_steps=n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        e.check_fault()
        best[arm] = min(best.get(arm, 1e9), dt)
        print(f"T={T} B={B} {smp}-{n} planner_split_mb2={arm}: {dt * 1e3:.2f} ms = {B / dt:.0f} plans/s = {fl / dt / 1e12:.1f} TF/s ({fl / dt / 157.3e12:.3f} of the fp32 MFMA peak)", flush=True)
        e.close()
for arm in arms[1:]:
    d = np.abs(outs[arm] - outs[arms[0]]).max()
    print(f"arm {arm} vs arm {arms[0]}: max |diff| over {B} plans (DDIM-{n}, same x_T) = {d:.2e} ({'bit-identical' if d == 0 else 'DIFFERENT'}); best {B / best[arm]:.0f} vs {B / best[arms[0]]:.0f} plans/s ({(best[arms[0]] / best[arm] - 1) * 100:+.1f} %)")
