import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
T = int(os.environ.get("T", "16"))
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25), 0))
e.set_option("no_batch_split", 1)
g = np.random.Generator(np.random.PCG64(0))
for B in [int(x) for x in os.environ.get("BS", "16,64,128,192,256,320,384,512,640,768,1024,1280,2048").split(",")]:
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)), dtype=torch.float32, device="cuda")
    for _ in range(2): e.plan_sample(cond, seed=1, sampler="ddim", n_steps=20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): e.plan_sample(cond, seed=1, sampler="ddim", n_steps=20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    print(f"T={T} B={B:5d}: {dt * 5:.1f} ms per 100 steps, {B / (dt * 5) * 1e3:.0f} plans/s", flush=True)
