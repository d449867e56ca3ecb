#!/bin/bash
# floor model inputs (kernel traces of the ablation build), first-conv chunk A/B, upper bound of folding the final launch
R=$(cd "$(dirname "$0")/../.." && pwd); OUT=$R/gpurun_out/r4; mkdir -p $OUT/floor; export TMPDIR=/tmp; cd /tmp
ABL=$R/latent_diffusion_planning_amd/libldp_hip_abl.so
for d in 0 16 24 64; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/floor/dbg$d -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lib $ABL --opt dbg=$d > $OUT/floor/dbg$d.log 2>&1
done
python $R/tools/r4/floor_model.py $OUT/floor 256 > $OUT/floor_model.txt 2>&1; cat $OUT/floor_model.txt
python $R/tools/r4/first_k.py 128 32 64 2>&1 | grep -v amdgpu.ids | tee $OUT/first_k.txt
for rep in 1 2; do for d in 0 4096; do
  python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --lib $ABL --opt dbg=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
done; done | tee $OUT/fold_final_upper_bound.txt
find $OUT/floor -name "*.csv" -size +30M -delete
