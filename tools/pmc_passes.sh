#!/bin/bash
# rocprofv3 PMC passes for bench.py (separate passes: SQ / TCC fetch / TCC write+hit), as
# MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.  Usage: tools/pmc_passes.sh OUTDIR [bench args]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o p -- python "$R/bench.py" $ARGS > "$OUT/sq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python "$R/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/write" -o p -- python "$R/bench.py" $ARGS > "$OUT/write.log" 2>&1
ls -la "$OUT"/*/ | head -30
