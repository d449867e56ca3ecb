#!/bin/bash
# Round 6: rocprofv3 counter passes of `bench.py --config C` for C in 2 3 4 -- the lines whose split-operand tiles, fp16-plane IDM
# kernel and StableVAE kernels had no counters.  Separate passes (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ | FETCH_SIZE | WRITE_SIZE + TCC hit/miss
# | LDS.  -> gpurun_out/r6/pmc_config<C>/{sq,fetch,write,lds} and profiles-ready summaries via tools/r6/pmc_summary.py.
#   tools/r6/pmc_configs.sh [configs...]
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
for C in ${@:-2 3 4}; do
  OUT=$R/gpurun_out/r6/pmc_config$C
  mkdir -p $OUT
  ARGS="--config $C --steps 1 --warmup 1 --no-cpu-baseline"
  timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- python $R/bench.py $ARGS > $OUT/sq.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
  timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/lds -o p -- python $R/bench.py $ARGS > $OUT/lds.log 2>&1
  python $R/tools/r6/pmc_summary.py $OUT $R/gpurun_out/r6/pmc_config$C.json > $R/gpurun_out/r6/pmc_config$C.txt 2>&1
  rm -rf $OUT/sq $OUT/fetch $OUT/write $OUT/lds          # (hundreds of MB of raw counter rows: only the summaries travel back)
  tail -25 $R/gpurun_out/r6/pmc_config$C.txt
done
