#!/usr/bin/env python3
"""Round 6: how well do two independent training tapes share the chip?  Two engine handles, the planner's gradient tape (B samples each) enqueued on two
streams at once against the same two tapes back to back on one stream.  python tools/r6/concurrency_probe.py [B]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd.engine import HipEngine     # noqa: E402
from tests.util import idm_params, planner_params, rng         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D, A, T = 25, 7, 8
engs = []
for _ in range(2):
    e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    e.load_params(planner=planner_params(D=D), idm=idm_params(D=D, A=A))
    e.train_init(["planner"])
    engs.append(e)
g = rng(1)
x0 = torch.tensor(g.uniform(-1, 1, (B, T, D)).astype(np.float32)).cuda()
cond = torch.tensor(g.uniform(-1, 1, (B, D)).astype(np.float32)).cuda()
nz = torch.tensor(g.standard_normal((B, T, D)).astype(np.float32)).cuda()
t = g.integers(0, 100, B)
main = torch.cuda.current_stream()
s2 = torch.cuda.Stream()


def run(mode, n=30):
    for it in range(n + 5):
        if it == 5:
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        if mode == "one":
            engs[0].train_planner_grad(x0, nz, t, cond)
        elif mode == "serial":
            engs[0].train_planner_grad(x0, nz, t, cond)
            engs[1].train_planner_grad(x0, nz, t, cond)
        else:
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                engs[1].train_planner_grad(x0, nz, t, cond)
            engs[0].train_planner_grad(x0, nz, t, cond)
            main.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for opt in (1, 0):
    for e in engs:
        e.set_option("train_streams", opt)
    r = {m: run(m) for m in ("one", "serial", "parallel", "one", "serial", "parallel")}
    print(f"B = {B}, train_streams = {opt}: one tape {r['one']:.3f} ms, two back to back {r['serial']:.3f} ms, two on two streams {r['parallel']:.3f} ms "
          f"(overlap efficiency {(r['serial'] - r['parallel']) / r['one']:.2f} of a tape hidden)")
