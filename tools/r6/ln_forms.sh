#!/bin/bash
# Round 6, VERDICT r5 item 2: experiment builds of the packed-fp32 LayerNorm of idm_block_h16_kernel (csrc/idm.hip LDP_LN_FORM):
#   1 the packed natural expression (round 5's failing form)        3 packed + s_waitcnt vmcnt(0) + 16 wait states behind the h store
#   2 packed, the global store of h issued AFTER the affine         4 packed, on a copy of the row (store data registers not overwritten)
# Only idm.hip is recompiled; the other objects come from csrc/build.  -> latent_diffusion_planning_amd/libldp_hip_ln<N>.so
set -e
cd "$(dirname "$0")/../../latent_diffusion_planning_amd/csrc"
make -j16 >/dev/null
HIPLIB=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for n in ${FORMS:-1 2 3 4 5}; do
  mkdir -p build_ln$n
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -DLDP_KERNARG_PRELOAD=1 \
        -mllvm -amdgpu-kernarg-preload-count=14 -DLDP_LN_FORM=$n -Ibuild -c idm.hip -o build_ln$n/idm.o &
done
wait
for n in ${FORMS:-1 2 3 4 5}; do
  g++ -shared -o ../libldp_hip_ln$n.so $(ls build/*.o | grep -v '/idm.o') build_ln$n/idm.o -L$HIPLIB -lamdhip64 -Wl,-rpath,$HIPLIB -Wl,-rpath,/opt/rocm/lib
done
ls -la ../libldp_hip_ln*.so
