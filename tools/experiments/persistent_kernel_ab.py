import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
B = 256
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25), 0))
cond = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for kv in sys.argv[4:]:
    k, v = kv.split("="); e.set_option(k, int(v))
for mode in [int(x, 0) for x in sys.argv[1].split(",")]:
    nbad = 0; sbs = {}
    for trial in range(NT):
        e.set_option("mega", 0)
        a = e.plan_sample(cond, seed=3 + trial, sampler="ddim", n_steps=n_steps, use_graph=False).clone()
        e.set_option("mega", mode)
        b = e.plan_sample(cond, seed=3 + trial, sampler="ddim", n_steps=n_steps, use_graph=False).clone()
        torch.cuda.synchronize()
        d = (a - b).abs()
        rows = (d.amax(dim=(1, 2)) > 0).nonzero().flatten().tolist()
        if mode & 512: rows = [r for r in rows if (mode >> 16) >> (r // 16) & 1]
        if rows:
            nbad += 1
            for r in rows: sbs[r // 16] = sbs.get(r // 16, 0) + 1
    print("mode", mode, "n_steps", n_steps, "bad trials", nbad, "of", NT, "; bad rows per sb", dict(sorted(sbs.items())), "fault", e.poll_fault(), "dbgword", hex(e.get_option("mega_ctr_max") >> 32), flush=True)
    e.set_option("safe_mode", 0)
