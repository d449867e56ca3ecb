#!/bin/bash
# Round 6: floor model of the planner loop at 5 and 16 plans (the reference's evaluation regime, eval_bc.yaml:13-14): kernel traces
# of bench.py on the ablation build (`make ablate`): dbg 0 full, 16 no epilogue, 24 no main loop and no epilogue, 64 empty kernel.
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
for B in ${@:-5 16}; do
  OUT=$R/gpurun_out/r6/floor_b$B
  for d in 0 16 24 64; do
    mkdir -p $OUT/dbg$d
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/dbg$d -o k -- python $R/bench.py --batch $B --steps 3 --warmup 1 --no-cpu-baseline \
      --lib $R/latent_diffusion_planning_amd/libldp_hip_abl.so --opt dbg=$d > $OUT/dbg$d.log 2>&1
  done
  python $R/tools/r6/small_floor.py $OUT $B > $R/gpurun_out/r6/small_batch_floor_b$B.txt 2>&1
  rm -rf $OUT
  tail -42 $R/gpurun_out/r6/small_batch_floor_b$B.txt
done
