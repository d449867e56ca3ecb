#!/bin/bash
# round-4 batch M: stride-2 / transposed convs on 16-row split tiles; goldens tiled to shard sizes; abl-build sanity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "tiled or neighbours or split_operands or full_size" -s 2>&1 | grep -E "max\|err|passed|failed|Error|error" | tail -12
{
for u in 0 1; do
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_updown=$u
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_updown=$u
python tools/r4/psplit.py 8 512 ddim 20 1 planner_split_updown=$u
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | awk 'NR%5==3 || NR%5==4 || NR%5==0' | tee gpurun_out/r4/m_updown.txt
for d in 0 64 24; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lib latent_diffusion_planning_amd/libldp_hip_abl.so --opt dbg=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($d, d['roofline']['avg_launch_us'])"
done
