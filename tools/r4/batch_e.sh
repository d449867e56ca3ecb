#!/bin/bash
# round-4 batch E: 16-row split tiles -- goldens, A/B against 32-row-only and fp32
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "split_operands" -s 2>&1 | grep -E "max\|err|passed|failed|Error|error" | head -20
{
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_tiles=1
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_ks=1
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_ks=2
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_tiles=1
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_ks=1
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_ks=2
python tools/r4/psplit.py 16 512 ddim 20 1 planner_split_ks=1
python tools/r4/psplit.py 8 512 ddim 20 1 planner_split_ks=1
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r4/e_psplit.txt
