#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: per-kernel calls/avg/total, tconv instantiations shortened."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls_per_eval = float(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    m = re.search(r"tconv_kernel<(.*?)>", r["Name"])
    n = "tconv<" + m.group(1).replace(" ", "") + ">" if m else r["Name"][:44]
    print(f"{n:36s} n={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:8.2f}us total={float(r['TotalDurationNs'])/1e6:9.2f}ms {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print(f"total {tot/1e6:.2f} ms")
