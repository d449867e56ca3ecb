import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(vae=W.init_vae_params(seed=2, decoder=False))
img = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (256, 64, 64, 3)), dtype=torch.float32, device="cuda")
for _ in range(4): e.vae_encode(img)
torch.cuda.synchronize()
