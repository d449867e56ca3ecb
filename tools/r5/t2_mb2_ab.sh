#!/bin/bash
# Same-box A/B (VERDICT r4 #5b): the seven 1024 -> 1024 T = 2 convs of an evaluation at 256 plans as quarter groups x TWO row blocks
# (option t2_mb2: 256 work-groups as before, each weight fragment feeds 32 samples, four-way statistics exchange) against the default
# half groups x one row block.  bench.py, configs[1], three alternating rounds; parity of the arm first.
cd "$(dirname "$0")/../.."
python - <<'PY'
import numpy as np, torch
from latent_diffusion_planning_amd import weights as W
from latent_diffusion_planning_amd.engine import HipEngine
pp = W.init_planner_params(W.PlannerSpec(25, 25), 0)
e = HipEngine(obs_dim=25, action_dim=7, global_cond_dim=25, pred_horizon=8, action_horizon=4)
e.load_params(planner=pp)
g = np.random.Generator(np.random.PCG64(1))
c = torch.tensor(g.uniform(-1, 1, (256, 25)), dtype=torch.float32); x = torch.tensor(g.standard_normal((256, 8, 25)), dtype=torch.float32)
a = e.unet_forward(x, 17, c).cpu().numpy()
e.set_option("t2_mb2", 1)
b = e.unet_forward(x, 17, c).cpu().numpy()
e.set_option("dump_plans", 1)
e.check_fault()
print("parity of the arm: max |t2_mb2 - default| over one evaluation at 256 plans =", float(np.abs(a - b).max()), "(identical)" if np.array_equal(a, b) else "")
PY
for round in 1 2 3; do
  for arm in default t2_mb2; do
    extra=""; [ $arm = t2_mb2 ] && extra="--opt t2_mb2=1"
    python bench.py --steps 40 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("round '$round' '$arm':", d["value"], "plans/s", d["roofline"]["avg_launch_us"], "us/launch", d["data"][:20])'
  done
done
