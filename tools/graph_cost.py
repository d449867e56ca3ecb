#!/usr/bin/env python3
"""What a hipGraph capture costs: wall time of the first call of a batch bucket (stream capture of
the whole loop + hipGraphInstantiate) against a replay, for the planner loop (3001 nodes at DDIM-100) and the joint
planner + IDM graph of LDPAgent.sample (3303 nodes)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from latent_diffusion_planning_amd import weights as W  # noqa: E402
from latent_diffusion_planning_amd.engine import HipEngine  # noqa: E402


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    D, A, T = 25, 7, 8
    eng = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    eng.load_params(planner=W.init_planner_params(W.PlannerSpec(D, D), 0), idm=W.init_idm_params(W.IDMSpec(D, A), 1))
    out = {}
    g = np.random.Generator(np.random.PCG64(0))
    for B in (5, 256):
        cond = torch.tensor(g.uniform(-1, 1, (B, D)).astype(np.float32), device="cuda")
        eng.plan_sample(cond, seed=0, sampler="ddim", n_steps=100, use_graph=False)          # warm the kernels / workspaces
        first = timed(lambda: eng.plan_sample(cond, seed=1, sampler="ddim", n_steps=100))
        replay = min(timed(lambda: eng.plan_sample(cond, seed=2, sampler="ddim", n_steps=100)) for _ in range(3))
        emb = torch.tensor(g.uniform(-1, 1, (B, 1, D)).astype(np.float32), device="cuda")
        eng.agent_sample(emb, 1, seed=0, sampler="ddim", planner_steps=100, idm_steps=100, use_graph=False)
        first_j = timed(lambda: eng.agent_sample(emb, 1, seed=1, sampler="ddim", planner_steps=100, idm_steps=100))
        replay_j = min(timed(lambda: eng.agent_sample(emb, 1, seed=2, sampler="ddim", planner_steps=100, idm_steps=100))
                       for _ in range(3))
        out[f"B={B}"] = dict(planner_first_ms=round(first, 2), planner_replay_ms=round(replay, 2),
                             planner_capture_ms=round(first - replay, 2), joint_first_ms=round(first_j, 2),
                             joint_replay_ms=round(replay_j, 2), joint_capture_ms=round(first_j - replay_j, 2))
    out["graphs_captured"] = eng.get_option("graphs_captured")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
