#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); OUT=$R/gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do
for o in "vae_split_pipe=0 vae_split_dual=1" "vae_split_pipe=1 vae_split_dual=1" "vae_split_pipe=0 vae_split_dual=0" "vae_split_pipe=1 vae_split_dual=0" "vae_split=0"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python $R/tools/r4/enc.py $args 2>/dev/null
done; done | tee $OUT/ab_pipe.txt
python $R/tools/r4/enc.py --decode --opt vae_split_pipe=0 2>/dev/null | tee -a $OUT/ab_pipe.txt
python $R/tools/r4/enc.py --decode --opt vae_split_pipe=1 2>/dev/null | tee -a $OUT/ab_pipe.txt
cd $R && timeout 600 python -m pytest tests/test_hip_vae.py -x -q -m gpu 2>&1 | tail -3
