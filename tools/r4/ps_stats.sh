#!/bin/bash
# per-kernel times of one planner evaluation with the split-operand layers on: ps_stats.sh [B] (T = 8 and 16, DDIM-10 under rocprofv3)
B=${1:-1024}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r4; cd /tmp; export TMPDIR=/tmp
for T in 8 16; do
PSPLIT_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/ks_psplit_t${T}_b$B -o k -- python $GRAFT_REPO_ROOT/tools/r4/psplit.py $T $B ddim 10 > /dev/null 2>&1
python - <<PY
import csv,re
rows=[]; tot=0
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4/ks_psplit_t${T}_b$B/k_kernel_stats.csv")):
    m=re.search(r'tconv_kernel<(.*?)>',r['Name'])
    if m: rows.append((m.group(1).replace(' ',''),int(r['Calls']),float(r['AverageNs'])/1e3,float(r['Percentage']))); tot+=float(r['TotalDurationNs'])
ev=160   # (1 + 2 + 5) calls x 10 steps x 2 repetitions
print("T=$T B=$B: %.1f us per evaluation"%(tot/1e3/ev))
for x in rows[:22]: print('  %-40s %2d/eval avg %7.1f us %5.1f%%'%(x[0],x[1]//ev,x[2],x[3]))
PY
done
