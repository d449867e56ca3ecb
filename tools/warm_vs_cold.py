#!/usr/bin/env python3
"""Splits a rocprofv3 kernel trace of a `bench.py --opt repeat=2` run into first (weights cold in L2) and second
(L2-warm) launch of every conv: the difference is what a next-layer weight prefetch could save."""
import csv, re, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tconv_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cold, warm = collections.defaultdict(list), collections.defaultdict(list)
i = 0
while i < len(rows):
    k = re.search(r"tconv_kernel<(.*?)>", rows[i]["Kernel_Name"]).group(1).replace(" ", "")
    d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
    if i + 1 < len(rows) and rows[i + 1]["Kernel_Name"] == rows[i]["Kernel_Name"] and "3,8,1,8,1" not in k:
        d2 = int(rows[i + 1]["End_Timestamp"]) - int(rows[i + 1]["Start_Timestamp"])
        cold[k].append(d); warm[k].append(d2); i += 2
    else:
        i += 1
for k in sorted(cold, key=lambda k: -sum(cold[k])):
    c, w = sum(cold[k]) / len(cold[k]) / 1e3, sum(warm[k]) / len(warm[k]) / 1e3
    print(f"{k:22s} n={len(cold[k]):5d} cold={c:7.2f}us warm={w:7.2f}us  delta={c - w:6.2f}")
