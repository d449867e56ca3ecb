import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(idm=W.init_idm_params(W.IDMSpec(25, 7), 1))
g = np.random.Generator(np.random.PCG64(0))
def t(tr, n=6):
    for _ in range(2): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in [int(x) for x in os.environ.get("BS", "1024,1536,2048,3072,4096,8192").split(",")]:
    tr = torch.tensor(g.uniform(-1, 1, (B * 4, 50)), dtype=torch.float32, device="cuda")
    out = []
    for hs in (0, 1, 2, 4, 8):
        e.set_option("idm_hs", hs); out.append("hs%d %.2f" % (hs, t(tr)))
    e.set_option("idm_hs", 0)
    print("B=%5d rows=%6d  " % (B, B * 4) + "  ".join(out), flush=True)
