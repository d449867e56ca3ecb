#!/usr/bin/env python3
"""Same-box A/B of the IDM loop (100-step DDPM) across builds: tools/r3/idm_ab.py lib1.so lib2.so ..."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, time, numpy as np, torch
sys.path.insert(0, %r)
from latent_diffusion_planning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(idm=W.init_idm_params(W.IDMSpec(25, 7), 1))
g = np.random.Generator(np.random.PCG64(0))
res = []
for B in (5, 256, 512, 1024):
    tr = torch.tensor(g.uniform(-1, 1, (B * 4, 50)), dtype=torch.float32, device="cuda")
    for _ in range(3): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): e.idm_sample(tr, seed=1)
    torch.cuda.synchronize(); res.append("B=%%d %%.3f ms" %% (B, (time.perf_counter() - t0) / 10 * 1e3))
print("%%-45s %%s" %% (sys.argv[1], "  ".join(res)))
''' % R
for rnd in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code, lib])
