#!/bin/bash
# Round 5 A/B on the driver's line (256 plans, exact fp32): the seven 1024 -> 1024 T = 2 convs as sixteen-wave work-groups (four K slices, four waves per SIMD,
# one sample per wave in the epilogue) -- option t2_w16; same box, alternating
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for o in 0 1; do
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline --opt t2_w16=$o 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[1] 256 plans DDIM-100 t2_w16=$o', d['value'], 'plans/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
done; done
