#!/usr/bin/env python3
"""Floor model of one planner evaluation at a SMALL batch (the 256-plan model is tools/r4/floor_model.py).  Per launch (30 per
evaluation): the empty-kernel time, prologue, main loop and epilogue from the ablation traces (tools/r6/small_floor.sh), next to two ideals of
the main loop:
  * HBM-ideal     = the launch's LIVE weight bytes / 6.3 TB/s (at <= 16 plans a launch is a weight stream: 16 samples share one tile)
  * chain-ideal   = the longest dependent MFMA chain a wave issues = (live (position, tap) pairs) x cin / (4 KS kw) instructions x 32 cycles / 2.4 GHz
    (KS = K slices over the work-group's waves -- the kernel's 4th template parameter --, kw = K split over work-groups, engine.hip's rule)
    floor = sum over launches of  gap + empty + prologue + max(HBM-ideal, chain-ideal) + epilogue.
    small_floor.py DIR B"""
import collections, csv, glob, re, statistics, sys

root, B = sys.argv[1], int(sys.argv[2])
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
        if i:
            gap[L].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
        name[L] = re.search(r"tconv_kernel<(.*?)>", r["Kernel_Name"]).group(1).replace(" ", "")
    med = lambda x: statistics.median(x) if x else 0.0
    return {L: med(dur[L]) for L in range(PERIOD)}, {L: med([g for g in gap[L] if g < 50]) for L in range(PERIOD)}, name


def taps(mode, to, j):
    if mode == 0: return to + j - 2
    if mode == 1: return 2 * to + j
    if mode == 2:
        q = to >> 1
        if to % 2 == 0: return q - 1 if j == 0 else q if j == 2 else -1
        return q if j == 1 else q + 1 if j == 3 else -1
    return to


def layer(mode, tin, tout, cin, cout, res=False):
    nj = {0: 5, 1: 3, 2: 4, 3: 1}[mode]
    pairs = [(to, j) for to in range(tout) for j in range(nj) if 0 <= taps(mode, to, j) < tin]
    live = len({j for _, j in pairs})
    return dict(pairs=len(pairs) + (tout if res else 0), wbytes=(live + (1 if res else 0)) * cin * cout * 4, cin=cin, cout=cout)


layers = []
dims = [256, 512, 1024]; t = 8; cin = 32
for l, d in enumerate(dims):
    layers += [layer(0, t, t, cin, d, True), layer(0, t, t, d, d), layer(0, t, t, d, d), layer(0, t, t, d, d)]
    cin = d
    if l < 2:
        layers.append(layer(1, t, t // 2, d, d)); t //= 2
layers += [layer(0, t, t, 1024, 1024)] * 4
for u, d in enumerate([512, 256]):
    layers += [layer(0, t, t, 2 * cin, d, True), layer(0, t, t, d, d), layer(0, t, t, d, d), layer(0, t, t, d, d)]
    layers.append(layer(2, t, 2 * t, d, d)); t *= 2
    cin = d
layers += [layer(0, t, t, 256, 256), layer(3, t, t, 256, 32)]

full, gap, name = load(0)
noepi, _, _ = load(16)
pro, _, _ = load(24)
empty, _, _ = load(64)
print(f"floor model of one U-Net evaluation, {B} plans, 30 launches (us; medians over the traced evaluations; ablation build)")
print(f"{'#':>2s} {'instantiation':26s} {'kw':>3s} {'gap':>5s} {'empty':>6s} {'prolog':>6s} {'main':>6s} {'hbm':>6s} {'chain':>6s} {'epilog':>6s} | {'launch':>6s}")
S = collections.Counter()
for L in range(PERIOD):
    tp = [x for x in name[L].split(",")]
    nwn, ks, cpi = int(tp[2]), int(tp[3]), int(tp[4])
    ly = layers[L]
    bn = 16 * nwn
    wgs = ((B + 15) // 16) * max(ly["cout"] // bn, 1)
    nit = max(ly["cin"] // (16 * cpi * ks), 1)
    kw = 1
    kws = tp[7] == "true"
    if kws:
        while kw < 8 and wgs * kw * 2 <= 256 and nit % (kw * 2) == 0 and nit // (kw * 2) >= 1: kw *= 2
    hbm = ly["wbytes"] / 6.3e12 * 1e6
    chain = ly["pairs"] * ly["cin"] / (4 * ks * kw) * 32 / 2.4e9 * 1e6
    p = max(pro[L] - empty[L], 0.0); m = max(noepi[L] - pro[L], 0.0); e = max(full[L] - noepi[L], 0.0)
    for k, v in (("gap", gap[L]), ("empty", empty[L]), ("prologue", p), ("main", m), ("hbm", hbm), ("chain", chain), ("ideal", max(hbm, chain)), ("epilogue", e), ("launch", full[L])):
        S[k] += v
    print(f"{L:2d} {name[L]:26s} {kw:3d} {gap[L]:5.2f} {empty[L]:6.2f} {p:6.2f} {m:6.2f} {hbm:6.2f} {chain:6.2f} {e:6.2f} | {full[L]:6.2f}")
tot = S["launch"] + S["gap"]
print(f"sum{'':29s} {S['gap']:5.1f} {S['empty']:6.1f} {S['prologue']:6.1f} {S['main']:6.1f} {S['hbm']:6.1f} {S['chain']:6.1f} {S['epilogue']:6.1f} | {S['launch']:6.1f}")
print(f"evaluation = {tot:.1f} us = gaps {S['gap']:.1f} + empty launches {S['empty']:.1f} + prologues {S['prologue']:.1f} + main loops {S['main']:.1f} "
      f"(HBM-ideal {S['hbm']:.1f}, MFMA-chain-ideal {S['chain']:.1f}) + epilogues {S['epilogue']:.1f}   -> 100-step loop {tot / 10:.2f} ms")
floor = S["gap"] + S["empty"] + S["prologue"] + S["ideal"] + S["epilogue"]
print(f"floor with every main loop at max(HBM-ideal, chain-ideal): {floor:.1f} us per evaluation -> {floor / 10:.2f} ms per 100-step loop")
print(f"floor with the main loops at the ideal AND no kernel boundary cost at all (gaps + empty launches removed): {(floor - S['gap'] - S['empty']):.1f} us -> {(floor - S['gap'] - S['empty']) / 10:.2f} ms")
