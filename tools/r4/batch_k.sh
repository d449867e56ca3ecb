#!/bin/bash
# round-4 batch K: weight loads of the 16-row split tiles issued over the first 100 / 50 / 33 % of a step's matrix instructions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_hip_planner.py -x -q -m gpu -k "split_operands" 2>&1 | tail -1
{
for lib in libldp_hip.so libldp_hip_lf50.so libldp_hip_lf33.so; do
echo "== $lib"
PSPLIT_ONLY=1 PSPLIT_LIB=latent_diffusion_planning_amd/$lib python tools/r4/psplit.py 16 1024 ddim 20 1
PSPLIT_ONLY=1 PSPLIT_LIB=latent_diffusion_planning_amd/$lib python tools/r4/psplit.py 8 1024 ddim 20 1
PSPLIT_ONLY=1 PSPLIT_LIB=latent_diffusion_planning_amd/$lib python tools/r4/psplit.py 16 1024 ddim 20 1 planner_split_cpi=8
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r4/k_lf.txt
