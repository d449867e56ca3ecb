import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
for T, D, A in ((16, 25, 7), (8, 30, 14)):
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    e.load_params(planner=W.init_planner_params(W.PlannerSpec(D, D), 0), idm=W.init_idm_params(W.IDMSpec(D, A), 1))
    g = np.random.Generator(np.random.PCG64(0))
    for B in (576, 768, 1280, 1536, 2560):
        obs = torch.tensor(g.uniform(-1, 1, (B, 1, D)), dtype=torch.float32, device="cuda")
        r = []
        for o in (0, 1):
            e.set_option("no_batch_split", o)
            for _ in range(2): e.agent_sample(obs, 1, seed=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): e.agent_sample(obs, 1, seed=1)
            torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 3 * 1e3)
        print(f"T={T} D={D} B={B}: two loops {r[0]:.1f} ms, one loop {r[1]:.1f} ms  ({r[1]/r[0]:.2f}x)", flush=True)
    e.close()
