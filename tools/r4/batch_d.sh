#!/bin/bash
# round-4 batch D: lazy plane packing -- full GPU suite, planner split A/B at 1024 / 512 / 256 plans
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4/d_pytest.txt; cat gpurun_out/r4/d_pytest.txt
{
python tools/r4/psplit.py 8 1024 ddim 50
python tools/r4/psplit.py 8 512 ddim 50
python tools/r4/psplit.py 8 256 ddim 50 2
python tools/r4/psplit.py 8 256 ddpm 100 2
python tools/r4/psplit.py 16 512 ddim 50
python tools/r4/psplit.py 16 256 ddim 50 2
} 2>&1 | grep -v Warning | tee gpurun_out/r4/d_psplit.txt
