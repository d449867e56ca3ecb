#!/bin/bash
# Round 6: A/B of the two-dimensional XCD placement of the split tiles' group-major launches (option place2d) on configs[4], [2], [3] and the driver's line
cd "$(dirname "$0")/../.."
for c in 4 2 3 1; do
  for p in 0 1 0 1; do
    timeout 400 python bench.py --config $c --no-cpu-baseline --opt place2d=$p 2>/dev/null | tail -1 > /tmp/p2d_line.json
    python - $c $p <<'PY'
import json, sys
d = json.load(open("/tmp/p2d_line.json"))
print(f"config {sys.argv[1]} place2d={sys.argv[2]}: {d['value']} plans/s, {d['ms_per_step']} ms per batch")
PY
  done
done
