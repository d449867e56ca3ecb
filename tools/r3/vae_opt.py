#!/usr/bin/env python3
"""StableVAE encode/decode time under ldp_set_option variants (one process): tools/r3/vae_opt.py name=value ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(vae=W.init_vae_params(seed=2))
g = np.random.Generator(np.random.PCG64(0))
img = torch.tensor(g.uniform(-1, 1, (256, 64, 64, 3)), dtype=torch.float32, device="cuda")
z = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)), dtype=torch.float32, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ref = e.vae_encode(img).clone()
for rnd in range(2):
    for kv in [""] + sys.argv[1:]:
        if kv:
            k, v = kv.split("="); e.set_option(k, int(v))
        out = e.vae_encode(img)
        print("%-12s encode N=256 %.2f ms   decode N=64 %.2f ms   max|diff vs default| %.2e" % (kv or "default", t(lambda: e.vae_encode(img)), t(lambda: e.vae_decode(z)), float((out - ref).abs().max())))
        if kv:
            e.set_option(k, 0)
