#!/usr/bin/env python3
"""Round 6: a long run of the training step as train_bc.py drives it (one update() per iteration, new batch every step, metrics read every 100 steps,
a snapshot-style parameter fetch every 1000): losses finite and falling on a learnable synthetic task, step time and device memory flat.
python tools/r6/train_long_run.py [steps]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cfgs                                       # noqa: E402
from tests.util import idm_params, make_agent, planner_params  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
D, A, T, B = 25, 7, 8, 256
ag, data = make_agent("rm", planner_params(D=D), idm_params(D=D, A=A))
# a learnable task: 64 fixed trajectories, revisited in random groups of 256 (with repetition): both losses can fall far below a random target's
pool = [cfgs.synth_latent_batch(data, 64, T + 1, 900 + i, with_actions=True) for i in range(1)]
base = {"obs": {k: torch.tensor(v).cuda() for k, v in pool[0]["obs"].items()}, "actions": torch.tensor(pool[0]["actions"]).cuda()}
g = np.random.Generator(np.random.PCG64(1))
t0 = time.time()
mem0 = None
for step in range(N):
    idx = torch.tensor(g.integers(0, 64, B)).cuda()
    batch = {"obs": {k: v[idx] for k, v in base["obs"].items()}, "actions": base["actions"][idx]}
    ag, m = ag.update(batch, step, step)
    if step % 500 == 0 or step == N - 1:
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated() / 1e6
        free, total = torch.cuda.mem_get_info()
        mem0 = mem0 or (total - free) / 1e6
        print(f"step {step:5d}: plan_loss {float(m['plan_loss']):.5f} idm_loss {float(m['idm_loss']):.5f} g_norm {float(m['g_norm']):.4f} lr {float(m['planner_lr']):.2e} | "
              f"{(time.time() - t0) / (step + 1) * 1e3:.2f} ms per step (wall, incl. batch gather) | device memory in use {(total - free) / 1e6:.0f} MB (torch {mem:.0f} MB)", flush=True)
    if step % 1000 == 999:
        n = sum(v.size for v in ag.planner_state.params.values())            # the snapshot path: parameters leave the GPU
        assert n > 6e7
ok = all(np.isfinite(float(m[k])) for k in ("plan_loss", "idm_loss", "g_norm"))
print("TRAIN_LONG_RUN_OK" if ok else "TRAIN_LONG_RUN_NAN")
