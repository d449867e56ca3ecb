#!/usr/bin/env python3
"""Per-wave timeline of the planner's conv launches (round-3 tool).

Needs the instrumented library: `make -C latent_diffusion_planning_amd/csrc timeline` builds
latent_diffusion_planning_amd/libldp_hip_tl.so with -DLDP_TIMELINE (every wave stamps s_memtime at eight
points of tconv_kernel; the product library carries none of this).  Runs the bench workload (DDIM-100, B plans)
from its hipGraph, reads the stamps of the first two evaluations and prints, per launch of the second one:
duration (last wave's end - first wave's entry, per XCD: s_memtime is a per-XCD counter; median over the XCDs), the gap to
the previous launch and the median over waves of each phase.  Stamps: 0 entry, 1 first tile staged (prologue barrier), 2 main loop done, 3 accumulators in LDS,
4 K-combine + statistics published, 5 projection pass done, 6 normalised + stores issued, 7 stores landed.
The instrumentation costs ~0.3 us per launch; the numbers are for comparing phases, not for the bench line.

    python tools/timeline.py [--batch 256] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latent_diffusion_planning_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "latent_diffusion_planning_amd", "libldp_hip_tl.so")
from latent_diffusion_planning_amd import weights as W  # noqa: E402
from latent_diffusion_planning_amd.engine import HipEngine  # noqa: E402

SLOT = 131072          # u64 per launch slot
PHASES = ["prologue", "main", "acc->lds", "K+stats", "proj", "norm+store", "drain"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--period", type=int, default=30)
    ap.add_argument("--horizon", type=int, default=8)
    ap.add_argument("--ghz", type=float, default=0.0, help="s_memtime ticks per ns (0: calibrate against the event-timed call)")
    ap.add_argument("--json")
    ap.add_argument("--raw", help="save the non-empty wave records of every slot (npz)")
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    D, A, T, B = 25, 7, a.horizon, a.batch
    eng = HipEngine(obs_dim=D, action_dim=A, global_cond_dim=D, pred_horizon=T, action_horizon=4)
    eng.load_params(planner=W.init_planner_params(W.PlannerSpec(D, D), 0))
    for o in a.opt:
        k, v = o.split("=")
        eng.set_option(k, int(v))
    buf = torch.zeros(64 * SLOT, dtype=torch.int64, device="cuda")
    eng.set_option("timeline_ptr", buf.data_ptr())
    g = np.random.Generator(np.random.PCG64(0))
    cond = torch.tensor(g.uniform(-1, 1, (B, D)).astype(np.float32), device="cuda")
    for _ in range(3):
        eng.plan_sample(cond, seed=1, sampler="ddim", n_steps=100)
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.plan_sample(cond, seed=2, sampler="ddim", n_steps=100)
    e1.record()
    torch.cuda.synchronize()
    call_ms = e0.elapsed_time(e1)
    flat = buf.cpu().numpy().reshape(64, SLOT)
    polls = flat[:, 120000:120002].copy()                      # [slot]: statistics exchanges, extra poll rounds
    flat[:, 120000:120002] = 0
    if os.environ.get("TL_DEBUG"):
        print("poll counters per slot:", polls[:, 0].tolist(), polls[:, 1].tolist())
    st = flat.reshape(64, SLOT // 8, 8)                       # [slot][wave record][stamp]
    P = a.period
    if a.raw:
        np.savez_compressed(a.raw, **{f"s{L}": st[L][(st[L] != 0).any(axis=1)] for L in range(2 * P)})
    # s_memtime is not one chip-wide counter: the records of a launch fall into clock domains tens of ms apart.
    # A launch lasts tens of us, so the domains of a slot separate cleanly by value; a domain is followed from one
    # slot to the next by its nearest base.  Everything is computed per domain, reported as the median over domains.
    def domains(rec):
        rec = rec[rec[:, 0] != 0]
        if len(rec) == 0:
            return []
        rec = rec[np.argsort(rec[:, 0])]
        cut = np.nonzero(np.diff(rec[:, 0]) > 250000)[0] + 1
        return np.split(rec, cut)
    if a.ghz <= 0:          # ticks between the first entries of evaluation 0 and 1 = one evaluation = call / 100
        d0, d1 = domains(st[0]), domains(st[P])
        a.ghz = float(d1[0][0, 0] - d0[0][0, 0]) / (call_ms * 1e6 / 100)
    print(f"call {call_ms:.3f} ms for 100 evaluations; s_memtime runs at {a.ghz:.4f} ticks/ns")
    us = lambda ticks: float(ticks) / a.ghz / 1e3
    rows = []
    prev = []             # (base, end) of every domain of the previous slot
    for L in range(2 * P):
        durs, gaps, skin, skout, n = [], [], [], [], 0
        phases, cur = [], []
        for r in domains(st[L]):
            n += len(r)
            done = np.where(r[:, 7] != 0, r[:, 7], r[:, 2])        # waves that left early (K-partial parts) have no late stamps
            t0, t1 = r[:, 0].min(), done.max()
            durs.append(us(t1 - t0))
            near = [e for (b0, e) in prev if abs(int(t0) - int(b0)) < 250000]
            if near:
                gaps.append(us(int(t0) - int(near[0])))
            cur.append((t0, t1))
            skin.append(us(r[:, 0].max() - t0))
            skout.append(us(t1 - np.median(done)))
            full = r[(r[:, 1:] != 0).all(axis=1)]
            if len(full):
                phases.append(full[:, 1:] - full[:, :-1])
        prev = cur
        if not durs:
            rows.append(None)
            continue
        ph = np.concatenate(phases) if phases else np.zeros((1, 7))
        rows.append(dict(layer=L % P, waves=n, domains=len(durs), dur_us=float(np.median(durs)),
                         gap_us=float(np.median(gaps)) if gaps else 0.0,
                         entry_skew_us=float(np.median(skin)), end_skew_us=float(np.median(skout)),
                         phases_us=[us(np.median(ph[:, i])) for i in range(7)],
                         phases_p90_us=[us(np.percentile(ph[:, i], 90)) for i in range(7)]))
    print("layer waves   dur   gap  skewIn skewOut | " + " ".join(f"{p:>10s}" for p in PHASES))
    tot = 0.0
    for row in rows[P:]:
        if row is None:
            continue
        tot += row["dur_us"] + (row["gap_us"] or 0)
        L = P + row["layer"]
        extra = f"  polls/exchange {1 + polls[L, 1] / polls[L, 0]:.2f}" if polls[L, 0] else ""
        print(f"{row['layer']:5d} {row['waves']:5d} {row['dur_us']:5.2f} {row['gap_us']:5.2f} {row['entry_skew_us']:6.2f} "
              f"{row['end_skew_us']:6.2f}  | " + " ".join(f"{x:10.2f}" for x in row["phases_us"]) + extra)
    print(f"sum of (gap + duration) over the evaluation: {tot:.1f} us")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
