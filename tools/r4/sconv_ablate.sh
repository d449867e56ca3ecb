#!/bin/bash
# Where the split-operand conv's time goes: ablation build (results wrong by construction) + PMC passes.
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/r4; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ABL=$R/latent_diffusion_planning_amd/libldp_hip_abl.so
for d in 0 1 2 4 8 3 10 11 15; do python $R/tools/r4/enc.py --lib $ABL --opt dbg=$((d * 65536)); done > $OUT/sconv_ablate.txt 2>&1
python $R/tools/r4/enc.py --opt vae_split=0 >> $OUT/sconv_ablate.txt 2>&1
cat $OUT/sconv_ablate.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sconv -o p -- python $R/tools/r4/enc.py --reps 1 > $OUT/pmc_sconv.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/pmc_sconv/*counter_collection.csv")
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    if "sconv3" not in k and "planes" not in k: continue
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); n[k]+=1
for k in d:
    c = d[k]; wc = c["SQ_WAVE_CYCLES"] or 1
    print(k, "launches", n[k], {x: round(c[x]/wc, 3) for x in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_ANY")},
          "mfma_busy/launch", c["SQ_VALU_MFMA_BUSY_CYCLES"]/n[k], "gui_active/launch", c["GRBM_GUI_ACTIVE"]/n[k],
          "lds_conflict/idx_active", round(c["SQ_LDS_BANK_CONFLICT"]/(c["SQ_LDS_IDX_ACTIVE"] or 1),3), "wave_cycles/launch", wc/n[k])
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +30M -delete
