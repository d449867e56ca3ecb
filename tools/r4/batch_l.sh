#!/bin/bash
# round-4 batch L: 32-row split tiles with a 2-way column split around 512 plans
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "neighbours or split_operands or full_size" 2>&1 | tail -3
{
python tools/r4/psplit.py 8 512 ddim 50 1 planner_split_cs2=0
python tools/r4/psplit.py 8 512 ddim 50 1 planner_split_cs2=1
python tools/r4/psplit.py 8 448 ddim 50 1 planner_split_cs2=0
python tools/r4/psplit.py 8 448 ddim 50 1 planner_split_cs2=1
} 2>&1 | grep -v "Warning\|amdgpu.ids" | awk 'NR%5==3 || NR%5==4 || NR%5==0' | tee gpurun_out/r4/l_cs2.txt
