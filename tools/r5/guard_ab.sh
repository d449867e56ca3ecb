#!/bin/bash
# Same-box A/B: what the range guard of the fp16-plane tiles costs.  `make -C csrc noguard` builds libldp_hip_ng.so
# (-DLDP_RANGE_GUARD=0: no compare in the tile epilogues, no compare in the plane producer); the two libraries alternate, three rounds.
# usage: tools/r5/guard_ab.sh > gpurun_out/r05_guard_ab.txt
cd "$(dirname "$0")/../.."
NG=latent_diffusion_planning_amd/libldp_hip_ng.so
for round in 1 2 3; do
  for cfg in 4 3; do
    for lib in default noguard; do
      extra=""; [ $lib = noguard ] && extra="--lib $NG"
      line=$(python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline $extra 2>/dev/null | tail -1)
      echo "round $round config $cfg $lib: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "plans/s", d["ms_per_step"], "ms/step", "shared-cond", d["config"].get("shared_cond_plans_per_s"))')"
    done
  done
done
