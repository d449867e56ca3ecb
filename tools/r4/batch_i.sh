#!/bin/bash
# round-4 batch I: 32-channel steps per LDS stage of the 16-row split tiles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "split_operands" -s 2>&1 | grep -E "max\|err|passed|failed|Error|error" | head -20
{
for c in 2 4 8; do
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_cpi=$c
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_cpi=$c
python tools/r4/psplit.py 16 512 ddim 20 1 planner_split_cpi=$c
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | awk 'NR%5==3 || NR%5==4 || NR%5==0' | tee gpurun_out/r4/i_psplit.txt
