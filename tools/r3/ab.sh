#!/bin/bash
# same-box A/B of two builds of the library: tools/r3/ab.sh libA.so libB.so [rounds] [extra bench flags]
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline --lib $L "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
