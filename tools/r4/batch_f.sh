#!/bin/bash
# round-4 batch F: 16-row split tiles for the T <= 4 layers where 32-row tiles under-fill (512 plans)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "split_operands" 2>&1 | tail -2
{
python tools/r4/psplit.py 8 512 ddim 20 1 planner_split_small=0
python tools/r4/psplit.py 8 512 ddim 20 1 planner_split_small=1
python tools/r4/psplit.py 16 512 ddim 20 1 planner_split_small=0
python tools/r4/psplit.py 16 512 ddim 20 1 planner_split_small=1
python tools/r4/psplit.py 8 768 ddim 20 1 planner_split_small=0
python tools/r4/psplit.py 8 384 ddim 20 1 planner_split_small=1
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r4/f_psplit.txt
