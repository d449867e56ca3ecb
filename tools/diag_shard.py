import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent_diffusion_planning_amd.engine import HipEngine
from tests.util import planner_params, rng
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params())
g = rng(12)
cond = torch.tensor(g.uniform(-1, 1, (24, 25)), dtype=torch.float32)
for trial in range(3):
    full = e.plan_sample(cond, seed=99, sampler="ddpm")
    lo = e.plan_sample(cond[:8], seed=99, row_offset=0, sampler="ddpm")
    hi = e.plan_sample(cond[8:], seed=99, row_offset=8, sampler="ddpm")
    full2 = e.plan_sample(cond, seed=99, sampler="ddpm")
    e.check_fault()
    d = (full[8:] - hi).abs()
    print("trial", trial, "lo diff", (full[:8] - lo).abs().max().item(), "hi diff", d.max().item(),
          "rows differing", (d.amax(dim=(1, 2)) > 0).nonzero().flatten().tolist(), "full vs full2", (full - full2).abs().max().item())
# single forward
x = torch.tensor(g.standard_normal((24, 8, 25)), dtype=torch.float32)
a = e.unet_forward(x, 40, cond); b = e.unet_forward(x[8:], 40, cond[8:])
print("forward diff", (a[8:] - b).abs().max().item())
for n in (1, 2, 5, 20):
    a = e.plan_sample(cond, seed=99, sampler="ddim", n_steps=n, x_init=x); b = e.plan_sample(cond[8:], seed=99, row_offset=8, sampler="ddim", n_steps=n, x_init=x[8:])
    print("ddim", n, "diff", (a[8:] - b).abs().max().item())
