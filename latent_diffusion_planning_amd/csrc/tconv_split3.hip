// 16-row split tiles on TWO fp16 planes per operand and THREE products (TConvCfg SPLIT = 3: v_mfma_f32_16x16x32_f16; tconv.hpp), one or two
// row blocks per wave: the k = 5 convs of the T = 8 and T = 4 layers and the stride-2 / transposed convs between the levels above 256 plans
// (engine.hip PlannerRun::conv; pred_horizon 16's (16, 256) transposed conv keeps the bf16 form: 388 registers there already)
#include "tconv_inst.hpp"
#define LISTH(X) \
  X(MODE_K5, 4, 8, 1, 2, 0, 1) \
  X(MODE_K5, 4, 8, 1, 2, 1, 1) \
  X(MODE_K5, 4, 8, 1, 2, 0, 2) \
  X(MODE_K5, 8, 4, 1, 2, 0, 1) \
  X(MODE_K5, 8, 4, 1, 2, 1, 1) \
  X(MODE_K5, 4, 4, 1, 2, 0, 1) \
  X(MODE_K5, 4, 4, 1, 2, 1, 1) \
  X(MODE_K5, 8, 2, 2, 2, 0, 1) \
  X(MODE_K5, 8, 2, 2, 2, 1, 1) \
  X(MODE_K5, 4, 2, 2, 2, 0, 1) \
  X(MODE_K5, 4, 2, 2, 2, 1, 1) \
  X(MODE_DOWN, 4, 2, 2, 2, 0, 1) \
  X(MODE_DOWN, 2, 4, 1, 2, 0, 1) \
  X(MODE_DOWN, 4, 4, 1, 2, 0, 1) \
  X(MODE_UP, 4, 4, 1, 2, 0, 1) \
  X(MODE_UP, 8, 2, 2, 2, 0, 1) \
  X(MODE_UP, 8, 4, 1, 2, 0, 1) \
  X(MODE_K5, 16, 2, 2, 2, 0, 1) \
  X(MODE_K5, 4, 4, 2, 2, 1, 1) \
  X(MODE_K5, 2, 8, 1, 2, 1, 2) \
  X(MODE_K5, 2, 4, 1, 2, 1, 2)
// 32-row tiles (the plain T = 2 layers) on fp16 planes (SPLIT = 4); with the projection the four 32 x 32 accumulator sets spill (193 .. 232 bytes
// per lane): those two convs of an evaluation stay on the six-product bf16 form
#define LISTH32(X) \
  X(MODE_K5, 2, 8, 2, 1, 0) \
  X(MODE_K5, 2, 4, 4, 1, 0) \
  X(MODE_K5, 2, 2, 8, 1, 0)
namespace ldp {
int tconv_launch_split3(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LISTH(LDP_CASE_SH)
    LISTH32(LDP_CASE_SH32)
    default: return -100;
  }
}
int tconv_init_split3() {
  LISTH(LDP_INIT_SH)
  LISTH32(LDP_INIT_SH32)
  return 0;
}
}  // namespace ldp
