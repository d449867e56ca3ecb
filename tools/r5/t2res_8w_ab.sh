#!/bin/bash
# Round 5 A/B at 1024 plans: the 2048 -> 512 T = 2 conv with the projection as eight waves (two K slices) -- option planner_split_8w also switches the
# 257..512-plan forms, which do not run here
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for o in 0 1; do
  python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_8w=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[4] shard (1024 candidates, DDIM-50) 8w=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
done; done
