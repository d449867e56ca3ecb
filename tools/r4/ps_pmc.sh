#!/bin/bash
# matrix-pipe busy fraction and effective clock of the planner's split-operand kernels: ps_pmc.sh [T] [B]
T=${1:-16}; B=${2:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4/pmc_psplit_t${T}_b$B; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PSPLIT_ONLY=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/tools/r4/psplit.py $T $B ddim 4 > $OUT/log.txt 2>&1
python - <<PY
import csv, collections, glob, re
f = glob.glob("$OUT/*counter_collection.csv")
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f[0])):
    m = re.search(r'tconv_kernel<(.*?)>', r["Kernel_Name"])
    if not m: continue
    k = m.group(1).replace(' ', '')
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
        if "Start_Timestamp" in r: dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("T=$T B=$B (counter pass: durations are longer than in a timing run)")
for k in sorted(d, key=lambda k: -dur[k])[:14]:
    c = d[k]; ga = c["GRBM_GUI_ACTIVE"] / 8 / n[k]          # summed over the 8 XCDs
    us = dur[k] / n[k] / 1e3
    print("  %-40s n %4d  %7.1f us  clock %.2f GHz  mfma busy %.3f of the launch's SIMD cycles  issuing %.3f  waiting on an instruction %.3f of wave cycles" % (
        k, n[k], us, ga / (us * 1e3) if us else 0, c["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k] / (ga * 1024), c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
