#!/usr/bin/env python3
"""Floor model of one planner evaluation at 256 plans: per launch (30 per evaluation) the boundary gap,
the empty-kernel time, prologue, main loop (with its MFMA-ideal time) and epilogue, from rocprofv3 --kernel-trace runs
of bench.py on the ablation build (dbg = 0 full, 16 no epilogue, 24 no main loop and no epilogue, 64 empty kernel):

    floor_model.py DIR [B]         DIR holds dbg0/ dbg16/ dbg24/ dbg64/ (each a rocprofv3 -d directory)

Durations are kernel-trace End - Start; the gap of launch i is Start_i - End_(i-1) in the dbg = 0 trace (inside the
replayed graph).  Everything is the median over the traced evaluations.
"""
import collections, csv, glob, re, statistics, sys

root, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256
PERIOD = 30

def load(d):
    f = glob.glob(f"{root}/dbg{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "tconv_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) % PERIOD:]
    dur, gap, name = collections.defaultdict(list), collections.defaultdict(list), {}
    for i, r in enumerate(rows):
        L = i % PERIOD
        dur[L].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if i and L:      # gaps inside an evaluation (the gap in front of launch 0 belongs to the previous evaluation's last launch)
            gap[L].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
        elif i:
            gap[0].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
        name[L] = re.search(r"tconv_kernel<(.*?)>", r["Kernel_Name"]).group(1).replace(" ", "")
    med = lambda x: statistics.median(x) if x else 0.0
    return {L: med(dur[L]) for L in range(PERIOD)}, {L: med([g for g in gap[L] if g < 50]) for L in range(PERIOD)}, name

def k5(t, cin, cout, res=False):
    return 2 * (5 * t - 6) * cin * cout + (2 * t * cin * cout if res else 0)
layers = []
dims = [256, 512, 1024]; t = 8; cin = 32
for l, d in enumerate(dims):
    layers += [k5(t, cin, d, True), k5(t, d, d), k5(t, d, d), k5(t, d, d)]
    cin = d
    if l < 2:
        layers.append(2 * 3 * (t // 2) * d * d); t //= 2
layers += [k5(t, 1024, 1024), k5(t, 1024, 1024)] * 2
for u, d in enumerate([512, 256]):
    layers += [k5(t, 2 * cin, d, True), k5(t, d, d), k5(t, d, d), k5(t, d, d)]
    layers.append(2 * 4 * t * d * d - 2 * 2 * d * d); t *= 2
    cin = d
layers += [k5(t, 256, 256), 2 * t * 256 * 32]

full, gap, name = load(0)
noepi, _, _ = load(16)
pro, _, _ = load(24)
empty, _, _ = load(64)
print(f"floor model of one U-Net evaluation, {B} plans, 30 launches (us; medians over the traced evaluations)")
print(f"{'#':>2s} {'instantiation':24s} {'gap':>6s} {'empty':>6s} {'prolog':>6s} {'main':>6s} {'ideal':>6s} {'epilog':>6s} | {'launch':>6s} {'+gap':>6s}  eff")
S = collections.Counter()
for L in range(PERIOD):
    ideal = layers[L] * B / 157.3e12 * 1e6
    p = max(pro[L] - empty[L], 0.0); m = max(noepi[L] - pro[L], 0.0); e = max(full[L] - noepi[L], 0.0)
    for k, v in (("gap", gap[L]), ("empty", empty[L]), ("prologue", p), ("main", m), ("ideal", ideal), ("epilogue", e), ("launch", full[L])):
        S[k] += v
    print(f"{L:2d} {name[L]:24s} {gap[L]:6.2f} {empty[L]:6.2f} {p:6.2f} {m:6.2f} {ideal:6.2f} {e:6.2f} | {full[L]:6.2f} {full[L] + gap[L]:6.2f}  {ideal / full[L]:4.2f}")
tot = S["launch"] + S["gap"]
print(f"sum{'':23s} {S['gap']:6.1f} {S['empty']:6.1f} {S['prologue']:6.1f} {S['main']:6.1f} {S['ideal']:6.1f} {S['epilogue']:6.1f} | {S['launch']:6.1f} {tot:6.1f}")
print(f"evaluation = {tot:.1f} us = gaps {S['gap']:.1f} + empty launches {S['empty']:.1f} + prologues {S['prologue']:.1f} + main loops {S['main']:.1f} "
      f"(MFMA-ideal {S['ideal']:.1f}) + epilogues {S['epilogue']:.1f}")
print(f"MFMA-ideal / evaluation = {S['ideal'] / tot:.3f} (the bench line's roofline.frac credits 0.16218 GF per plan-step: x{0.16218e9 * B / 157.3e12 * 1e6 / S['ideal']:.3f} of this table's ideal)")
floor = S["gap"] + S["empty"] + S["prologue"] + S["ideal"] + S["epilogue"]
print(f"floor if every main loop ran at the MFMA rate: {floor:.1f} us -> frac {0.16218e9 * B / 157.3e12 * 1e6 / floor:.3f}")
