// 16-row split tiles over TWO row blocks per wave (TConvCfg SPLIT = 2, MB = 2: v_mfma_f32_16x16x32_bf16, 32 samples x 16 columns x
// T positions per wave): every weight fragment feeds twice the matrix instructions and an LDS stage + barrier covers twice the
// samples.  Taken by the T = 4 layers of 1024 channels (128-column GroupNorm groups, eight waves) once 32-sample work-groups cover the
// chip (engine.hip PlannerRun::conv); values are those of the MB = 1 tiles.  The 64-column groups of the 512-channel level measured
// slower in this form (four waves: -2.6 %, eight waves over two K slices: -1.0 % plans/s at 1024 plans) and are not built.
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 4, 8, 1, 2, 0) \
  X(MODE_K5, 4, 8, 1, 2, 1)
namespace ldp {
int tconv_launch_split2(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LIST(LDP_CASE_S2)
    default: return -100;
  }
}
int tconv_init_split2() {
  LIST(LDP_INIT_S2)
  return 0;
}
}  // namespace ldp
