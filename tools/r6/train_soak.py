#!/usr/bin/env python3
"""Round 6: determinism soak of the training step's gradients (in-launch split-K finish with tickets + sc1 partial blocks, weight gradients on a side stream,
the two tapes on two streams): N repetitions of the same step per batch size, every gradient arena compared bitwise with the first.
python tools/r6/train_soak.py [N]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd.engine import HipEngine     # noqa: E402
from tests.util import idm_params, planner_params, rng         # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D, A, T = 25, 7, 8
e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
e.load_params(planner=planner_params(D=D), idm=idm_params(D=D, A=A))
e.train_init(["planner", "idm"])
side, main = e.aux_streams()["idm"], torch.cuda.current_stream()
bad_total = 0
for B in (32, 64, 256, 320, 512):
    g = rng(500 + B)
    emb = g.uniform(-1, 1, (B, T + 1, D)).astype(np.float32)
    act = g.uniform(-1, 1, (B, T + 1, A)).astype(np.float32)
    x0, cond = torch.tensor(emb[:, 1:]).cuda(), torch.tensor(emb[:, 0]).cuda()
    s2 = torch.tensor(np.concatenate([emb[:, :-1], emb[:, 1:]], axis=-1).reshape(-1, 2 * D)).cuda()
    a0 = torch.tensor(act[:, :-1].reshape(-1, A).copy()).cuda()
    npl, nid = torch.tensor(g.standard_normal((B, T, D)).astype(np.float32)).cuda(), torch.tensor(g.standard_normal((B * T, A)).astype(np.float32)).cuda()
    tp, ti = g.integers(0, 100, B), g.integers(0, 100, B * T)

    def grads():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            li = e.train_idm_grad(s2, a0, nid, ti)
        lp = e.train_planner_grad(x0, npl, tp, cond)
        main.wait_stream(side)
        return torch.stack([lp, li]).clone(), e.train_arena("planner", e.TRAIN_GRADS).clone(), e.train_arena("idm", e.TRAIN_GRADS).clone()
    ref = grads()
    t0 = time.time()
    bad = 0
    for _ in range(N):
        got = grads()
        bad += int(not all(torch.equal(r, x) for r, x in zip(ref, got)))
    torch.cuda.synchronize()
    bad_total += bad
    print(f"batch {B:4d}: {N} repetitions of planner + IDM gradients, {bad} differ from the first bitwise; {(time.time() - t0) / N * 1e3:.2f} ms per repetition "
          f"(losses {float(ref[0][0]):.6f} / {float(ref[0][1]):.6f})", flush=True)
print("TRAIN_SOAK_OK" if bad_total == 0 else f"TRAIN_SOAK_MISMATCH {bad_total}")
