#!/usr/bin/env python3
"""Per-layer launch durations of the planner U-Net from a rocprofv3 --kernel-trace CSV of bench.py:
the conv launches of an evaluation repeat with period 30 (pred_horizon 8).  Prints, per layer, the
instantiation, the average duration and the MFMA-ideal time (algorithmic FLOPs / fp32 MFMA peak)."""
import csv, re, sys, collections

path, period = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rows = [r for r in csv.DictReader(open(path)) if "tconv_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) % period:]
dur = collections.defaultdict(list); name = {}
for i, r in enumerate(rows):
    L = i % period
    dur[L].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    name[L] = re.search(r"tconv_kernel<(.*?)>", r["Kernel_Name"]).group(1).replace(" ", "")
# planner layer list (D=25 -> 32 padded first-layer chunk 128, dims 256/512/1024, T=8)
def k5(t, cin, cout, res=False):
    pairs = 5 * t - 6
    return 2 * pairs * cin * cout + (2 * t * cin * cout if res else 0)
layers = []
dims = [256, 512, 1024]; t = 8; cin = 32
for l, d in enumerate(dims):
    layers += [k5(t, cin, d, True), k5(t, d, d), k5(t, d, d), k5(t, d, d)]
    cin = d
    if l < 2:
        layers.append(2 * 3 * (t // 2) * d * d - 2 * d * d * 0); t //= 2
layers += [k5(t, 1024, 1024), k5(t, 1024, 1024)] * 2
for u, d in enumerate([512, 256]):
    layers += [k5(t, 2 * cin, d, True), k5(t, d, d), k5(t, d, d), k5(t, d, d)]
    layers.append(2 * 4 * t * d * d - 2 * 2 * d * d); t *= 2   # transposed conv k4 s2, 2 edge taps lost
    cin = d
layers += [k5(t, 256, 256), 2 * t * 256 * 32]
tot = 0
for L in range(period):
    d = sum(dur[L]) / len(dur[L])
    fl = layers[L] * B if L < len(layers) else 0
    ideal = fl / 157.3e12 * 1e6
    tot += d
    print(f"{L:2d} {name[L]:22s} {d:7.2f} us  ideal {ideal:6.2f}  eff {ideal / d:5.2f}  overhead {d - ideal:6.2f}")
print("sum", round(tot, 1))
