#!/bin/bash
# round-3 experiment 1: final conv over position pairs, 64-channel first conv, K split on the concat convs
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/r3_exp1; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_planner.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for o in "" "--opt no_fin_rows=1" "--opt kw_concat=1" ""; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee $OUT/bench.txt
cd /tmp
for tag in base kwc; do
  o=""; [ $tag = kwc ] && o="--opt kw_concat=1"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$tag -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $o > $OUT/ks_$tag.log 2>&1
  f=$(find $OUT/ks_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/layer_times.py $f 30 256 > $OUT/layers_$tag.txt 2>&1
  rm -f $f
done
cat $OUT/layers_base.txt; cat $OUT/layers_kwc.txt
