#!/bin/bash
# Round 5 A/B at 353..512 plans: the T = 2 convs on fp16 planes as 16-row tiles over whole groups (three products, no in-launch exchange)
#   planner_split_t2res16 = 0                      : all on the 32-row tiles over half groups (projection convs: three bf16 planes / six products)
#   planner_split_t2res16 = 1, _t2all16 = 0        : the two convs with the projection on the 16-row fp16 tiles
#   planner_split_t2res16 = 1, _t2all16 = 1 (default): all nine
# same box, alternating
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for o in "0 0" "1 0" "1 1"; do set -- $o
  python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$1 --opt planner_split_t2all16=$2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[3] shard (512 aloha frames) t2res16=$1 t2all16=$2', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$1 --opt planner_split_t2all16=$2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 512 plans         t2res16=$1 t2all16=$2', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
  python bench.py --config 1 --batch 400 --steps 20 --warmup 3 --no-cpu-baseline --opt planner_split_t2res16=$1 --opt planner_split_t2all16=$2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planner DDIM-100, 400 plans         t2res16=$1 t2all16=$2', d['value'], 'plans/s', d['ms_per_step'], 'ms')"
done; done
