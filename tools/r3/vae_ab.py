#!/usr/bin/env python3
"""Same-box A/B of StableVAE encode/decode across builds: tools/r3/vae_ab.py lib1.so lib2.so ... (one subprocess each)."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
from latent_diffusion_planning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
_lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k != "ldp_mean_sq_diff" or hasattr(C.CDLL(_lib.LIB_PATH), k)}
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
print("%%-40s encode N=256 %%.2f ms   decode N=64 %%.2f ms" %% (sys.argv[1], t(lambda: e.vae_encode(img)), t(lambda: e.vae_decode(z))))
''' % R
for rnd in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", code, lib])
