#!/usr/bin/env python3
"""StableVAE encode of 256 frames, timed: enc.py [--lib PATH] [--opt name=value ...] [--reps N] [--decode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
args = sys.argv[1:]
if "--lib" in args:
    from latent_diffusion_planning_amd import _lib
    i = args.index("--lib"); _lib.LIB_PATH = os.path.abspath(args[i + 1]); del args[i:i + 2]
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
opts, reps, dec = {}, 5, "--decode" in args
for i, a in enumerate(args):
    if a == "--opt": k, v = args[i + 1].split("="); opts[k] = int(v)
    if a == "--reps": reps = int(args[i + 1])
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(vae=W.init_vae_params(seed=2, decoder=dec))
for k, v in opts.items(): e.set_option(k, v)
g = np.random.Generator(np.random.PCG64(0))
x = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)) if dec else g.uniform(-1, 1, (256, 64, 64, 3)), dtype=torch.float32, device="cuda")
fn = (lambda: e.vae_decode(x)) if dec else (lambda: e.vae_encode(x))
fn(); fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): fn()
torch.cuda.synchronize()
print(f"{'decode64' if dec else 'encode256'} {opts} {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
