#!/bin/bash
# socket power and shader clock while the planner loop runs on exact fp32 and on split operands: power_watch.sh [T] [B]
T=${1:-16}; B=${2:-1024}; cd $GRAFT_REPO_ROOT
watch() { for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done; }
rocm-smi --showmaxpower 2>/dev/null | grep -i "max graphics" | sed 's/.*GPU/GPU/'
for sp in 0 1; do
  echo "== planner_split=$sp (T=$T, $B plans, DDIM-50 loops back to back)"
  python - <<PY &
import sys, os, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=$T, action_horizon=4)
e.set_option("planner_split", $sp)
e.load_params(planner=W.init_planner_params(W.PlannerSpec(25, 25), 0))
cond = torch.tensor(np.random.default_rng(1).uniform(-1, 1, ($B, 25)), dtype=torch.float32, device="cuda")
e.plan_sample(cond, seed=1, sampler="ddim", n_steps=50); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 9.0: e.plan_sample(cond, seed=2, sampler="ddim", n_steps=50)
torch.cuda.synchronize()
PY
  sleep 4; watch | tail -10; wait
done
