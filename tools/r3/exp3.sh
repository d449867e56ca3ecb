#!/bin/bash
# round-3 A/B helper: planner GPU tests, bench at 256 (x2) / 512 / 1024 plans, per-layer launch times.
# usage: tools/r3/exp3.sh TAG [extra bench.py flags]
R=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-x}; shift
OUT=$R/gpurun_out/r3_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_hip_planner.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for i in 1 2; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee $OUT/bench.txt
for B in 512 1024; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee -a $OUT/bench.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/ks.log 2>&1
f=$(find $OUT/ks -name "*kernel_trace.csv" | head -1)
python $R/tools/layer_times.py $f 30 256 > $OUT/layers.txt 2>&1
rm -f $f
cat $OUT/layers.txt
