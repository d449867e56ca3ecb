#!/bin/bash
# timing ablations of the split-operand planner kernels (-DLDP_ABLATE build): per-kernel averages with the main loop on cache-hot operands (256),
# without the main loop (8), without the epilogue (16): ps_ablate.sh [T] [B]
T=${1:-16}; B=${2:-1024}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4/abl_psplit_t${T}_b$B; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for d in ${DBGS:-0 256 8 16}; do
PSPLIT_ONLY=1 PSPLIT_LIB=$R/latent_diffusion_planning_amd/libldp_hip_abl.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -o k -- python $R/tools/r4/psplit.py $T $B ddim 4 1 dbg=$d $PSPLIT_OPTS > $OUT/d$d.log 2>&1
done
python - <<PY
import csv,re,collections,os
DB=[int(x) for x in os.environ.get("DBGS","0 256 8 16").split()]
t=collections.defaultdict(dict)
for d in DB:
    for r in csv.DictReader(open("$OUT/d%d/k_kernel_stats.csv"%d)):
        m=re.search(r'tconv_kernel<(.*?)>',r['Name'])
        if m: t[m.group(1).replace(' ','')][d]=(float(r['AverageNs'])/1e3,float(r['TotalDurationNs']))
print("T=$T B=$B  us per launch, dbg =", DB, "(0 full, 256 main loop on cache-hot operands, 8 no main loop, 16 no epilogue; split tiles: 512 no LDS writes / barrier, 1024 no weight loads, 2048 no staging loads)")
for k in sorted(t,key=lambda k:-t[k].get(0,(0,0))[1])[:14]:
    print("  %-40s "%k+" ".join("%7.1f"%t[k].get(d,(0,0))[0] for d in DB))
PY
find $OUT -name "*_kernel_trace.csv" -delete
