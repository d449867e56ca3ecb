#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); OUT=$R/gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gputests.txt 2>&1; tail -5 $OUT/gputests.txt
$R/tools/bin/split_bf16_probe 2>&1 | head -16 > $OUT/split_probe_rate_random.txt; grep RANDOM $OUT/split_probe_rate_random.txt
cd /tmp
python $R/tools/bench_parts.py small > $OUT/small_batch.json 2> $OUT/small_batch.err; cat $OUT/small_batch.json
python $R/tools/bench_parts.py --lib $R/latent_diffusion_planning_amd/libldp_hip_nt.so small > $OUT/small_batch_nt.json 2>> $OUT/small_batch.err
python $R/tools/bench_parts.py small > $OUT/small_batch_2.json 2>> $OUT/small_batch.err
python - <<PY
import json
a=json.load(open("$OUT/small_batch.json"))["small_batch"]; b=json.load(open("$OUT/small_batch_nt.json"))["small_batch"]; c=json.load(open("$OUT/small_batch_2.json"))["small_batch"]
for B in ("B1","B5","B16"):
    print(B, "planner loop ms: default", a[B]["planner_loop_ms"], "nt", b[B]["planner_loop_ms"], "default again", c[B]["planner_loop_ms"], "| agent", a[B]["agent_sample_ms"], b[B]["agent_sample_ms"], c[B]["agent_sample_ms"])
PY
python $R/bench.py --configs0 > $OUT/configs0.json 2> $OUT/configs0.err; cat $OUT/configs0.json
