#!/usr/bin/env python3
"""StableVAE encode latency at env-harness batch sizes (N = 1, 5, 16, 50 images); GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(vae=W.init_vae_params(seed=2))
g = np.random.Generator(np.random.PCG64(0))
for N in (1, 5, 16, 50):
    img = torch.tensor(g.uniform(-1, 1, (N, 64, 64, 3)), dtype=torch.float32, device="cuda")
    for _ in range(3): e.vae_encode(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e.vae_encode(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"vae_encode N={N}: {dt*1e3:.3f} ms")
