#!/usr/bin/env python3
"""Round 5: repeat-determinism soak of the planner / IDM paths in every batch regime and of the StableVAE (after the packed-fp32 finding in the IDM kernel, DESIGN 4.2:
a sporadic wrong lane shows up as a call that differs from its own repeat).  N calls each on fixed inputs, every one bit-equal to the first."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd import weights as W                      # noqa: E402
from latent_diffusion_planning_amd.engine import HipEngine                  # noqa: E402
from tests.util import idm_params, planner_params, rng                      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
D, A = 25, 7
e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params(D=D), idm=idm_params(D=D, A=A), vae=W.init_vae_params(seed=2))
g = rng(77)


def soak(name, fn):
    first = fn()
    first = first if isinstance(first, (list, tuple)) else [first]
    bad = 0
    for _ in range(N - 1):
        out = fn()
        out = out if isinstance(out, (list, tuple)) else [out]
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
    fin = all(bool(torch.isfinite(t).all()) for t in first)
    print(f"{name:58s} {N} calls, {bad} differ from the first; finite {fin}", flush=True)


for B in (5, 64, 256):                                     # the exact-fp32 regimes (column / K split, in-launch exchanges)
    cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32).cuda()
    soak(f"planner loop DDIM-20, {B} plans", lambda: e.plan_sample(cond, seed=3, sampler="ddim", n_steps=20))
    obs = torch.tensor(g.uniform(-1, 1, (B, 1, D)), dtype=torch.float32).cuda()
    soak(f"joint planner + IDM graph DDIM-20, {B} plans", lambda: list(e.agent_sample(obs, 1, seed=4, sampler="ddim", planner_steps=20, idm_steps=20)))
for B in (300, 512, 1024, 1500):
    cond = torch.tensor(g.uniform(-1, 1, (B, D)), dtype=torch.float32).cuda()
    soak(f"planner loop DDIM-20, {B} plans", lambda: e.plan_sample(cond, seed=3, sampler="ddim", n_steps=20))
    obs = torch.tensor(g.uniform(-1, 1, (B, 1, D)), dtype=torch.float32).cuda()
    soak(f"joint planner + IDM graph DDIM-20, {B} plans", lambda: list(e.agent_sample(obs, 1, seed=4, sampler="ddim", planner_steps=20, idm_steps=20)))
for Nimg in (64, 256):
    img = torch.tensor(g.uniform(-1, 1, (Nimg, 64, 64, 3)), dtype=torch.float32).cuda()
    soak(f"StableVAE encode, {Nimg} frames", lambda: e.vae_encode(img))
z = torch.tensor(g.uniform(-3, 3, (64, 2, 2, 4)), dtype=torch.float32).cuda()
soak("StableVAE decode, 64 latents", lambda: e.vae_decode(z))
print("fault kinds", e.poll_fault_kinds())
