#!/bin/bash
# GPU batch of the two-row-block / fp16-plane split tiles (tconv SPLIT = 2 / 3 / 4): what of profiles/r04_* the changes move, regenerated on the final tree --
# the driver's bench line + its kernel stats, the other configurations (configs[2] joint graph and its kernel stats), the parity margins,
# the planner A/B by horizon and batch (exact fp32 -> default) and the per-layer times at 1024 / 512 plans.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04n; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_bench -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ks_bench.log 2>&1
python $R/tools/bench_parts.py idm vae cfg3 cfg4 cfg5 agent > $OUT/other_configs.json 2> $OUT/other_configs.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_cfg3 -o k -- python $R/tools/bench_parts.py cfg3 > $OUT/ks_cfg3.log 2>&1
python $R/tools/parity_margin.py $OUT/parity_margins.json > $OUT/parity_margins.log 2>&1
{ for tb in "8 1024 ddim 50" "16 1024 ddim 50" "8 512 ddim 50" "16 512 ddim 50" "8 2048 ddim 50" "16 2048 ddim 50" "16 320 ddim 50"; do python $R/tools/r4/psplit.py $tb; done; } 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/split_planner_ab.txt
{ python $R/tools/r4/mb2.py 16 1024 ddim 50 0 1; python $R/tools/r4/mb2.py 16 2048 ddim 50 0 1; } 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/split_mb2_final.txt
# bf16 x 6 (planner_split_f16 = 0) against the default (fp16 x 3), and the (16, 256) level on exact fp32 against the default
{ for tb in "16 1024" "16 512" "16 320" "8 1024" "8 512" "8 2048"; do python $R/tools/r4/mb2.py $tb ddim 50 1,planner_split_f16=0 1; done; python $R/tools/r4/mb2.py 16 1024 ddim 50 1,planner_split_t16=0 1; } 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/split_f16_final.txt
python $R/tools/r4/f16_check.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/split_f16_margins.txt
{ bash $R/tools/r4/ps_stats.sh 1024; bash $R/tools/r4/ps_stats.sh 512; } > $OUT/split_planner_layers.txt 2>&1
rm -f $R/gpurun_out/r4/planner_split_margins.json
( cd $R && timeout 600 python -m pytest tests/test_hip_planner.py -q -m gpu -k split_operands > $OUT/split_planner_margins_test.txt 2>&1 ); cp $R/gpurun_out/r4/planner_split_margins.json $OUT/split_planner_margins.json 2>/dev/null
python $R/tools/stress_exchange.py 20 > $OUT/stress_exchange.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
ls -la $OUT
