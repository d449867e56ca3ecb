#!/bin/bash
# Round 6's evidence set, written into gpurun_out/prof_r06 on the GPU box (only summaries: the raw traces are deleted before the copy back);
# tools/r6/copy_profiles.sh copies the judged files into profiles/.
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_bench -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ks_bench.log 2>&1
f=$(find $OUT/ks_bench -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/layer_times.py $f 30 256 > $OUT/layer_times_b256.txt 2>&1
bash $R/tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/pmc
for c in 2 3 4; do
  python $R/bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_config$c -o k -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $OUT/ks_config$c.log 2>&1
done
python $R/bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_train -o k -- python $R/bench.py --train --steps 10 > $OUT/ks_train.log 2>&1
python $R/tools/bench_parts.py idm vae cfg3 agent small > $OUT/other_configs.json 2> $OUT/other_configs.err
python $R/tools/parity_margin.py $OUT/parity_margins.json > $OUT/parity_margins.log 2>&1
python $R/tools/stress_exchange.py 100 > $OUT/stress_exchange.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
ls $OUT
