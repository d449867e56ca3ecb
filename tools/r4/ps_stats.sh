cd $GRAFT_REPO_ROOT; python -m pytest tests/test_hip_planner.py -x -q -m gpu -k split_operands -s 2>&1 | grep -E "max\|err|passed|failed|Error" | head
mkdir -p gpurun_out/r4; cd /tmp; export TMPDIR=/tmp
for T in 8 16; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4/ks_psplit_t$T -o k -- python $GRAFT_REPO_ROOT/tools/r4/psplit.py $T 1024 ddim 10 > /dev/null 2>&1
python - <<PY
import csv,re
rows=[]
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4/ks_psplit_t$T/k_kernel_stats.csv")):
    m=re.search(r'tconv_kernel<(.*?)>',r['Name'])
    if m: rows.append((m.group(1).replace(' ',''),int(r['Calls']),float(r['AverageNs'])/1e3,float(r['Percentage'])))
print("T=$T")
for x in rows[:16]: print('  %-40s calls %5d avg %7.1f us %5.1f%%'%x)
PY
done
