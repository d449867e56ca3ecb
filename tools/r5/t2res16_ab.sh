#!/bin/bash
# Round 5 A/B at 353..512 plans: the two T = 2 convs with the projection on fp16 planes as 16-row tiles over whole groups (option planner_split_t2res16)
# against three bf16 planes / six products on the 32-row tile over half groups; same box, alternating
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for o in 0 1; do
  python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[3] shard (512 aloha frames) t2res16=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 512 plans         t2res16=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 400 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 400 plans         t2res16=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
done; done
