#!/usr/bin/env python3
"""Round 5: the fused IDM block on fp16 planes over 32-row tiles (idm_block_h16_kernel, default from 2048 rows) against the exact-fp32 16-row
kernel (option idm_f16 = 0) on the same box: one forward against the float64 oracle, the 100-step loop against each other, ms per loop."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd.engine import HipEngine   # noqa: E402
from oracle import torch32                                   # noqa: E402
from tests.util import idm_params, planner_params, rng       # noqa: E402

D, A = 25, 7
ip = idm_params(D=D, A=A)
e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params(D=D), idm=ip)
P = torch32.TorchParams(ip, dtype=torch.float64)


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for R in (2048, 2100, 4096, 8192):
    g = rng(900 + R)
    s, a = g.uniform(-1, 1, (R, 2 * D)), g.standard_normal((R, A))
    sf, af = torch.tensor(s, dtype=torch.float32).cuda(), torch.tensor(a, dtype=torch.float32).cuda()
    ref = torch32.idm_forward(P, torch.tensor(s), torch.tensor(a), 37).numpy()
    out = {}
    for f16 in (1, 0):
        e.set_option("idm_f16", f16)
        got = e.idm_forward(sf, af, 37).cpu().numpy()
        err = np.abs(got - ref).max()
        bad = np.nonzero(np.abs(got - ref).max(axis=1) > 5e-5)[0]
        if len(bad): print('   bad rows', R, 'f16' if f16 else 'fp32', len(bad), bad[:8], bad[-4:], flush=True)
        loop = e.idm_sample(sf, seed=5).cpu().numpy()
        ms = timeit(lambda: e.idm_sample(sf, seed=5)) * 1e3
        out[f16] = (err, loop, ms, e.get_option("stat_f16_launches") if hasattr(e, "get_option") else -1)
    d = np.abs(out[1][1] - out[0][1]).max()
    print(f"R={R:5d}  forward err vs f64: f16 {out[1][0]:.2e}  fp32 {out[0][0]:.2e}   100-step loop |f16 - fp32| {d:.2e} finite={np.isfinite(out[1][1]).all()}"
          f"   loop ms: f16 {out[1][2]:.3f}  fp32 {out[0][2]:.3f}  ({out[0][2] / out[1][2]:.2f}x)  f16 launches {out[1][3]}", flush=True)
print("fault word:", e.poll_fault_kinds() if hasattr(e, "poll_fault_kinds") else "?")
