#!/usr/bin/env python3
"""Round 6: WHERE does the packed-LayerNorm build of idm_block_h16_kernel go wrong?  An experiment build (tools/r6/ln_forms.sh)
writes every hidden slice's LayerNorm output y and input v of every block to a debug buffer (option timeline_ptr).  All four slices of a row tile
compute the same row from the same global data, so: (1) do the slices agree on v; (2) is y == LayerNorm(v) recomputed on the host in float32;
(3) for the rows that are not: which lane (= column / 4), which element (= column % 4), which wave / row slot, and what value came out.
   python tools/r6/ln_dump.py --lib latent_diffusion_planning_amd/libldp_hip_ln1.so [R] [calls]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd import _lib               # noqa: E402
if "--lib" in sys.argv:
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from latent_diffusion_planning_amd.engine import HipEngine   # noqa: E402
from tests.util import idm_params, planner_params, rng       # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D, A, H, HS, NB = 25, 7, 256, 4, 3
ip = idm_params(D=D, A=A)
e = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=8, action_horizon=4)
e.load_params(planner=planner_params(D=D), idm=ip)
print(os.path.basename(_lib.LIB_PATH), _lib.load().ldp_version().decode(), "R =", R)
Rp = (R + 31) // 32 * 32
buf = torch.zeros((2, NB, HS, Rp, H), dtype=torch.float32, device="cuda")
g = rng(99)
s = torch.tensor(g.uniform(-1, 1, (R, 2 * D)), dtype=torch.float32).cuda()
a = torch.tensor(g.standard_normal((R, A)), dtype=torch.float32).cuda()
e.idm_forward(s, a, 5)                                   # allocates the workspace (ws_R) before the debug pointer is set
torch.cuda.synchronize()
wsR = None
for cand in (Rp, (R + 15) // 16 * 16):
    wsR = cand
tot_bad = 0
lane_hist, el_hist, slot_hist, blk_hist, slice_hist = np.zeros(64, int), np.zeros(4, int), np.zeros(32, int), np.zeros(NB, int), np.zeros(HS, int)
shown = 0
for call in range(CALLS):
    buf.zero_()
    e.set_option("timeline_ptr", buf.data_ptr())
    got = e.idm_forward(s, a, 5 + call)
    torch.cuda.synchronize()
    y1 = buf.clone()
    again = e.idm_forward(s, a, 5 + call)
    torch.cuda.synchronize()
    e.set_option("timeline_ptr", 0)
    ne = (got != again).any(dim=1)
    dif = (y1 != buf)
    print(f"  call {call}: eps rows not bit-equal between two runs (dump on): {int(ne.sum())}; dumped y elements differing between the runs: {int(dif[0].sum())}, v: {int(dif[1].sum())}"
          + (f"; first bad eps rows {ne.nonzero().flatten()[:8].tolist()}" if ne.any() else ""))
    if dif[0].any():
        idx = dif[0].nonzero()[:6].tolist()
        for b_, j_, r_, c_ in idx:
            print(f"    y differs: block {b_} slice {j_} row {r_} (tile row {r_ % 32}) col {c_} (lane {c_ // 4}, element {c_ % 4}): {float(y1[0, b_, j_, r_, c_]):.7g} vs {float(buf[0, b_, j_, r_, c_]):.7g}")
    y = buf[0].cpu().numpy()[:, :, :R]                    # (NB, HS, R, H)
    v = buf[1].cpu().numpy()[:, :, :R]
    if call == 0 and not v.any():
        print("  debug buffer empty: ws_R differs from", Rp, "or the build has no dump"); break
    for b in range(NB):
        ls = ip[f"MLPResNet_0/MLPResNetBlock_{b}/LayerNorm_0/scale"].astype(np.float32)
        lb = ip[f"MLPResNet_0/MLPResNetBlock_{b}/LayerNorm_0/bias"].astype(np.float32)
        v_dis = (v[b] != v[b][0:1]).any(axis=(0, 2))      # rows whose slices disagree on v
        vv = v[b][0].astype(np.float32)
        mean = vv.astype(np.float64).mean(1)
        var = np.maximum((vv.astype(np.float64) ** 2).mean(1) - mean ** 2, 0)
        ref = ((vv - mean[:, None]) / np.sqrt(var + 1e-6)[:, None] * ls + lb)
        for j in range(HS):
            err = np.abs(y[b, j] - ref)
            bad = err > 1e-4 * (1 + np.abs(ref))
            n = int(bad.sum())
            if not n:
                continue
            tot_bad += n
            rows, cols = np.nonzero(bad)
            np.add.at(lane_hist, cols // 4, 1); np.add.at(el_hist, cols % 4, 1); np.add.at(slot_hist, rows % 32, 1)
            blk_hist[b] += n; slice_hist[j] += n
            for r_, c_ in list(zip(rows, cols))[:3]:
                if shown < 24:
                    shown += 1
                    others = [float(y[b, jj, r_, c_]) for jj in range(HS)]
                    # which row's statistics would explain the wrong value?
                    want = float(ref[r_, c_])
                    got = float(y[b, j, r_, c_])
                    cand = ((vv[r_, c_] - mean) / np.sqrt(var + 1e-6)) * ls[c_] + lb[c_]
                    k = int(np.argmin(np.abs(cand - got)))
                    print(f"  call {call} block {b} slice {j} row {r_} (tile row {r_ % 32}: wave {r_ % 8}, slot {r_ % 32 // 8}) col {c_} (lane {c_ // 4}, element {c_ % 4}): "
                          f"got {got:.7g} want {want:.7g}; slices {others}; v {vv[r_, c_]:.7g}; best other-row statistics: row {k} ({'same tile' if k // 32 == r_ // 32 else 'other tile'}, "
                          f"tile row {k % 32}) -> {cand[k]:.7g}")
        if v_dis.any():
            print(f"  call {call} block {b}: {int(v_dis.sum())} rows whose slices DISAGREE ON v (the LayerNorm input)")
print(f"  bad y elements over {CALLS} calls: {tot_bad}")
if tot_bad:
    print("  by lane group 0-15/16-31/32-47/48-63:", [int(lane_hist[i:i + 16].sum()) for i in range(0, 64, 16)], " by element:", el_hist.tolist())
    print("  by tile row slot (wave = r % 8, q = r // 8):", slot_hist.tolist())
    print("  by block:", blk_hist.tolist(), " by slice:", slice_hist.tolist())
