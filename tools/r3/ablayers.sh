#!/bin/bash
# same-box per-layer A/B of two builds: tools/r3/ablayers.sh libA.so libB.so
R=$(cd "$(dirname "$0")/../.." && pwd); export TMPDIR=/tmp
OUT=$R/gpurun_out/r3_abl; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for t in A B; do
  L=$1; [ $t = B ] && L=$2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$t -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lib $R/$L > $OUT/ks_$t.log 2>&1
  f=$(find $OUT/ks_$t -name "*kernel_trace.csv" | head -1)
  python $R/tools/layer_times.py $f 30 256 > $OUT/layers_$t.txt 2>&1
  rm -f $f
done
paste <(cut -c1-38 $OUT/layers_A.txt) <(cut -c25-38 $OUT/layers_B.txt)
