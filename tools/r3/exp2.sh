#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
bash $R/tools/ablate.sh $R/gpurun_out/r3_exp2 0 256 8 16 24 64 > $R/gpurun_out/r3_exp2.txt 2>&1
cat $R/gpurun_out/r3_exp2.txt
