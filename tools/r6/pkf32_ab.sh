#!/bin/bash
# Round 6: what building WITHOUT packed-fp32 instructions (csrc/Makefile NOPKF32) costs: bench.py of every configuration on the product
# library and on `make pkf32` (packed instructions kept), alternating, same box.   tools/r6/pkf32_ab.sh [rounds]
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
for i in $(seq 1 ${1:-2}); do
  for c in 1 2 3 4; do
    for L in libldp_hip.so libldp_hip_pk.so; do
      python bench.py --config $c --no-cpu-baseline --steps 20 --lib latent_diffusion_planning_amd/$L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c  %-20s %10.1f plans/s  %8.3f ms' % ('$L', d['value'], d['ms_per_step']))"
    done
  done
done
