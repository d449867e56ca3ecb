#!/bin/bash
# round-4 batch G: which split tile for the T = 4 / T = 2 layers, by batch size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
{
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_t4=16
python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_t4=32
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_t4=16
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_t4=32
python tools/r4/psplit.py 8 1024 ddim 20 1 planner_split_t2=16
python tools/r4/psplit.py 8 768 ddim 20 1
python tools/r4/psplit.py 8 768 ddim 20 1 planner_split_t2=32
python tools/r4/psplit.py 8 512 ddim 20 1
python tools/r4/psplit.py 8 2048 ddim 20 1
python tools/r4/psplit.py 8 2048 ddim 20 1 planner_split_t4=16
python tools/r4/psplit.py 16 2048 ddim 20 1
python tools/r4/psplit.py 16 2048 ddim 20 1 planner_split_t4=16
python tools/r4/psplit.py 16 320 ddim 20 1
} 2>&1 | grep -v "Warning\|amdgpu.ids" | awk 'NR%5==3 || NR%5==4 || NR%5==0' | tee gpurun_out/r4/g_psplit.txt
