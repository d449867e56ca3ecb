#!/usr/bin/env python3
"""Prints the HIP path's max abs error against the committed float64 goldens for the 100-step loops
(how much of the 1e-4 budget is used).  GPU box only; reads tests/golden, never /root/reference."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests.cases import load_case
from tests.util import planner_params
from latent_diffusion_planning_amd.engine import HipEngine
eng = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
eng.load_params(planner=planner_params())
inp, exp = load_case("bench_rows_b256_ddim100")
got = eng.plan_sample(torch.tensor(inp["cond"], dtype=torch.float32), x_init=torch.tensor(inp["x0"], dtype=torch.float32), sampler="ddim", n_steps=100).cpu().numpy()
rows = exp["rows"].astype(int)
print("bench rows max abs err vs float64 oracle:", np.abs(got[rows] - exp["plan"]).max())
for name, smp, n in (("planner_loop_ddpm100", "ddpm", 100), ("planner_loop_ddim50", "ddim", 50)):
    inp, exp = load_case(name)
    got = eng.plan_sample(torch.tensor(inp["cond"], dtype=torch.float32), x_init=torch.tensor(inp["x0"], dtype=torch.float32),
                          step_noise=torch.tensor(inp["nz"], dtype=torch.float32), sampler=smp, n_steps=n).cpu().numpy()
    print(name, np.abs(got - exp["plan"]).max())
