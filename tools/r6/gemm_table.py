#!/usr/bin/env python3
"""Round 6: per-GEMM table of one training step.  Joins the LDP_GEMM lines the library prints under LDP_TRAIN_TRACE=1 (one per launch, in launch
order) with the seg_gemm / reduce_parts rows of a rocprofv3 kernel trace of the same run (ordered by start time) and prints, for the LAST step,
each launch's shape, time, TF/s and the reduce that follows it.     python tools/r6/gemm_table.py <stderr log> <kernel_trace.csv> [steps]"""
import csv, sys, collections

log, trace = sys.argv[1], sys.argv[2]
gemms = [dict(kv.split("=") for kv in l.split()[1:]) for l in open(log) if l.startswith("LDP_GEMM")]
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "seg_gemm" in n or "reduce_parts" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "reduce" if "reduce_parts" in n else "gemm"))
rows.sort()
kern = [r for r in rows if r[2] == "gemm"]
assert len(kern) == len(gemms), (len(kern), len(gemms))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
per = len(gemms) // steps
first = len(gemms) - per
# reduce following each gemm
red_after = {}
gi = -1
for r in rows:
    if r[2] == "gemm":
        gi += 1
    else:
        red_after[gi] = (r[1] - r[0]) / 1e3
tot = collections.Counter()
print(f"{'#':>3} {'form':4} {'M':>5} {'N':>5} {'K':>5} {'nb':>3} {'steps':>6} {'ks':>3} {'tile':>7} {'GF':>7} {'us':>7} {'TF/s':>6} {'reduce us':>9} {'gap us':>7}")
prev_end = None
for i in range(first, len(gemms)):
    g, k = gemms[i], kern[i]
    us = (k[1] - k[0]) / 1e3
    gf = float(g["gflop"])
    red = red_after.get(i, 0.0)
    gap = (k[0] - prev_end) / 1e3 if prev_end else 0.0
    prev_end = k[1]
    print(f"{i - first:3d} {g['form']:4} {g['M']:>5} {g['N']:>5} {g['K']:>5} {g['nb']:>3} {g['steps']:>6} {g['ks']:>3} {g['tile']:>7} {gf:7.3f} {us:7.1f} {gf / us * 1e3:6.1f} {red:9.1f} {gap:7.1f}")
    tot["gf"] += gf; tot["us"] += us; tot["red"] += red
    tot[g["form"] + "_us"] += us; tot[g["form"] + "_gf"] += gf
print(f"total: {per} launches, {tot['gf']:.2f} GFLOP, gemm {tot['us']:.0f} us ({tot['gf'] / tot['us'] * 1e3:.1f} TF/s), reduce {tot['red']:.0f} us")
for f in ("NN", "NT", "TN"):
    print(f"  {f}: {tot[f + '_gf']:.2f} GFLOP in {tot[f + '_us']:.0f} us = {tot[f + '_gf'] / max(tot[f + '_us'], 1e-9) * 1e3:.1f} TF/s")
