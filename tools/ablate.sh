#!/bin/bash
# Kernel-time ablation of the dominant conv kernels: LDP_DBG bit 1 = no weight reloads,
# 2 = no MFMA, 4 = no activation restaging.  Results are wrong by construction; only timings matter.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for d in 0 1 2 4 3 5 6 7; do
  LDP_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dbg$d -o a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/dbg$d.log 2>&1
  echo "== LDP_DBG=$d"; grep tconv $OUT/dbg$d/a_kernel_stats.csv | head -6 | awk -F'","|",|,"' '{print $1, $4}' | sed 's/void ldp::tconv_kernel//; s/(ldp::ConvArgs)//' 
done
