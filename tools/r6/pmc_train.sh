#!/bin/bash
# Round 6: counter passes of the training step (tools/r6/train_bench.py --steps 3): SQ | fabric | LDS per kernel -> gpurun_out/r6/pmc_train.{json,txt}
R=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/r6/pmc_train
mkdir -p $OUT
CMD="python $R/tools/r6/train_bench.py --steps 3 --warmup 1 $*"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/lds -o p -- $CMD > $OUT/lds.log 2>&1
python $R/tools/r6/pmc_summary.py $OUT $R/gpurun_out/r6/pmc_train.json > $R/gpurun_out/r6/pmc_train.txt 2>&1
rm -rf $OUT/sq $OUT/fetch $OUT/write $OUT/lds
cat $R/gpurun_out/r6/pmc_train.txt
