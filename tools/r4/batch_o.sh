#!/bin/bash
# GPU batch of the StableVAE on fp16 planes (sconv3 NPL = 2): the full GPU suite, then the VAE pieces of profiles/r04_* on the final tree
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04o; mkdir -p $OUT $R/gpurun_out/r4; export TMPDIR=/tmp
( cd $R && timeout 1300 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $OUT/gputests.log
cd /tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $R/tools/bench_parts.py vae vae_ab cfg3 cfg4 cfg5 agent > $OUT/other_configs.json 2> $OUT/other_configs.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_vae -o k -- python $R/tools/bench_parts.py vae > $OUT/ks_vae.log 2>&1
( cd $R && timeout 600 python -m pytest tests/test_hip_vae.py -q -m gpu -k margins > $OUT/vae_margins_test.txt 2>&1 ); cp $R/gpurun_out/r4/vae_margins.json $OUT/split_vae_margins.json 2>/dev/null
python $R/tools/parity_margin.py $OUT/parity_margins.json > $OUT/parity_margins.log 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
cat $OUT/gputests.log; ls -la $OUT
