import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from latent_diffusion_planning_amd.engine import HipEngine
from tests.util import planner_params, idm_params, rng
for T, B in ((8, 1024), (16, 1024), (8, 512)):
    e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params(), idm=idm_params())
    g = rng(5)
    cond = torch.tensor(g.uniform(-1, 1, (B, 25)).astype(np.float32))
    outs = []
    for p in (0, 1, 0, 1):
        e.set_option("place2d", p)
        outs.append(e.plan_sample(cond, seed=3, sampler="ddim", n_steps=10).cpu().numpy())
    print(T, B, "bitwise equal:", np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[3]), "finite:", np.isfinite(outs[1]).all())
    e.close()
