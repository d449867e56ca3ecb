#!/bin/bash
# Round 5's evidence set, written straight into gpurun_out/prof_r05 on the GPU box (tools/r5/copy_profiles.sh copies the judged
# summaries into profiles/).  Leaner than tools/make_profiles.sh (round 4): the driver's line with its kernel trace and PMC passes,
# the two new one-command configurations (bench.py --config 3 / 4) with their kernel traces, the other pieces, parity margins.
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/prof_r05
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_bench -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ks_bench.log 2>&1
bash $R/tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
for c in 3 4; do
  python $R/bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_config$c -o k -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $OUT/ks_config$c.log 2>&1
done
python $R/tools/bench_parts.py idm vae cfg3 agent small > $OUT/other_configs.json 2> $OUT/other_configs.err
python $R/tools/parity_margin.py $OUT/parity_margins.json > $OUT/parity_margins.log 2>&1
f=$(find $OUT/ks_bench -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/layer_times.py $f 30 256 > $OUT/layer_times_b256.txt 2>&1
python $R/tools/stress_exchange.py 100 > $OUT/stress_exchange.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls -la $OUT
