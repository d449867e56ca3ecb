#!/bin/bash
# Round 6, final tree: the evidence set of the training step (profiles/r06_update_*).   bash tools/r6/update_profiles.sh   (on the GPU box)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/update_final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# 1. the bench line (50 steps) and the A/B of what the round added, same box, same process count
timeout 300 python bench.py --train --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/r06_update_bench_b256.json
{
  echo "# LDPAgent.update at 256 samples, ms per step (tools/r6/train_bench.py --steps 30 --opt ...), same box, back to back.  Defaults: train_streams=1 train_fuse_reduce=1 train_sides=1"
  for o in "" "train_streams=0" "train_fuse_reduce=0" "train_streams=0 --opt train_fuse_reduce=0" "train_streams=3" "train_sides=2" "train_sides=3" ""; do
    printf "%-50s " "${o:-defaults}"
    timeout 200 python tools/r6/train_bench.py --steps 30 ${o:+--opt $o} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_gpu'], 'ms', d['samples_per_s'], 'samples/s', d['frac_of_fp32_mfma_peak'], 'of fp32 peak')"
  done
  for w in planner idm; do
    printf "%-50s " "which=$w"
    timeout 200 python tools/r6/train_bench.py --steps 30 --which $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_gpu'], 'ms', d['tflops'], 'TF/s')"
  done
} > $OUT/r06_update_streams_ab.txt 2>&1
# 2. kernel trace of the default step (13 steps) -> per-kernel totals
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python tools/r6/train_bench.py --steps 10 --warmup 3 > $OUT/trace_run.json 2>/dev/null
ST=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
[ -n "$ST" ] && cp $ST $OUT/r06_kernel_stats_update_b256.csv
rm -rf $OUT/t
# 3. the per-GEMM table, one stream (so that durations are the kernels' own) and the default
LDP_TRAIN_TRACE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python tools/r6/train_bench.py --steps 3 --warmup 1 --opt train_streams=0 > /dev/null 2> $OUT/run.err
CSV=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
{ echo "# per-GEMM table of one training step (256 samples), train_streams=0 (one stream: every duration is the kernel's own).  tools/r6/gemm_table.py"; python tools/r6/gemm_table.py $OUT/run.err $CSV 4; } > $OUT/r06_update_gemm_table.txt 2>&1
rm -rf $OUT/t $OUT/run.err
ls -la $OUT
cat $OUT/r06_update_bench_b256.json | cut -c1-300
cat $OUT/r06_update_streams_ab.txt
tail -5 $OUT/r06_update_gemm_table.txt
