#!/bin/bash
# Round 5 A/B: 257 .. 512 plans, the four-wave fp16-plane tiles of the T = 4 / T = 8 layers as eight waves (option planner_split_8w), same box, alternating
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for o in 0 1; do
  python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --opt planner_split_8w=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[3] shard (512 aloha frames) 8w=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_8w=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 512 plans         8w=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 320 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_8w=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 320 plans         8w=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
done; done
