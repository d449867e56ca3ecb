#!/bin/bash
# per-GEMM table of the training step (tools/r6/gemm_table.py): trace 4 steps (1 warm-up) and tabulate the last
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/gemm_table
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
LDP_TRAIN_TRACE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python tools/r6/train_bench.py --steps 3 --warmup 1 "$@" > $OUT/run.json 2> $OUT/run.err
grep -c LDP_GEMM $OUT/run.err
CSV=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
python tools/r6/gemm_table.py $OUT/run.err $CSV 4 > $OUT/table.txt
python - "$CSV" > $OUT/other.txt <<'PY'
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# last quarter = the last step (4 steps traced)
n = len(rows) // 4
last = rows[-n:]
span = (last[-1][1] - last[0][0]) / 1e3
busy = sum(e - s for s, e, _ in last) / 1e3
c = collections.Counter(); k = collections.Counter()
for s, e, nme in last:
    nm = nme.split("(")[0].replace("void ", "").replace("ldp::(anonymous namespace)::", "")[:60]
    c[nm] += (e - s) / 1e3; k[nm] += 1
print(f"last step: {n} launches, span {span:.0f} us, sum of kernel durations {busy:.0f} us, idle {span - busy:.0f} us")
for nm, us in c.most_common(30):
    print(f"{nm:62} {k[nm]:4d} {us:8.1f} us")
PY
tail -8 $OUT/table.txt; head -30 $OUT/other.txt
rm -rf $OUT/t
