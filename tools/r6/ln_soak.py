#!/usr/bin/env python3
"""Round 6: soak of one build flavour of idm_block_h16_kernel's LayerNorm (tools/r6/ln_forms.sh) at the failing occupancy --
two work-groups per CU (4096 / 2100 rows, four hidden slices).  N forward calls on fresh inputs, each against a repeat of itself (bit equality)
and every fifth against the float64 oracle.   python tools/r6/ln_soak.py [--lib PATH] [N]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from latent_diffusion_planning_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from latent_diffusion_planning_amd import _lib                # noqa: E402
from latent_diffusion_planning_amd.engine import HipEngine   # noqa: E402
from oracle import torch32                                   # noqa: E402
from tests.util import idm_params, planner_params, rng       # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D, A = 25, 7
ip = idm_params(D=D, A=A)
e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params(D=D), idm=ip)
P = torch32.TorchParams(ip, dtype=torch.float64)
print(os.path.basename(_lib.LIB_PATH), _lib.load().ldp_version().decode())
for R in (4096, 2100, 1040):
    worst, unequal, bad_rows = 0.0, 0, 0
    for it in range(N):
        g = rng(7000 + 13 * it + R)
        s, a = g.uniform(-1, 1, (R, 2 * D)), g.standard_normal((R, A))
        sf, af = torch.tensor(s, dtype=torch.float32).cuda(), torch.tensor(a, dtype=torch.float32).cuda()
        k = int(g.integers(0, 100))
        got = e.idm_forward(sf, af, k)
        again = e.idm_forward(sf, af, k)
        ne = (got != again).any(dim=1)
        unequal += int(ne.any())
        bad_rows += int(ne.sum())
        if it % 5 == 0:
            ref = torch32.idm_forward(P, torch.tensor(s), torch.tensor(a), k).numpy()
            worst = max(worst, float(np.abs(got.cpu().numpy() - ref).max()))
    print(f"  R={R:5d}: {N} forward pairs, {unequal} not bit-equal ({bad_rows} rows differ); worst error vs float64 {worst:.2e}", flush=True)
print("  fault kinds", e.poll_fault_kinds(), "range_fallback", e.get_option("range_fallback"))
