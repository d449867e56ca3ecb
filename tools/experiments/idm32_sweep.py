#!/usr/bin/env python3
"""IDM loop (100-step DDPM) at B plans: the 16-row kernel against the 32-row one over hidden splits / placements."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from latent_diffusion_planning_amd import weights as W, flops
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(idm=W.init_idm_params(W.IDMSpec(25, 7), 1))
g = np.random.Generator(np.random.PCG64(0))
def t(tr, n=10):
    for _ in range(3): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
dbgs = [int(x) for x in sys.argv[1:]] or [0]
for B in [int(x) for x in os.environ.get('BS', '128,192,256,384,512,768,1024').split(',')]:
    rows = B * 4
    tr = torch.tensor(g.uniform(-1, 1, (rows, 50)), dtype=torch.float32, device="cuda")
    out = []
    for dbg in dbgs:
        e.set_option("dbg", dbg)
        e.set_option("idm_rows32", 0); out.append("dbg=%d 16-row %.3f" % (dbg, t(tr)))
        e.set_option("idm_rows32", 1)
        for hs in (2, 4, 8):
            if (rows + 31) // 32 * hs > 256: continue
            e.set_option("idm_hs32", hs)
            for place in (0, 1):
                e.set_option("idm_rt_major32", place)
                out.append("hs%d/p%d %.3f" % (hs, place, t(tr)))
        e.set_option("idm_hs32", 0); e.set_option("idm_rt_major32", 0)
    e.set_option("dbg", 0)
    print("B=%4d  " % B + "  ".join(out), flush=True)
