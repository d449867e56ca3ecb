// k = 5 convolutions of the planner on split bf16 operands (TConvCfg SPLIT: v_mfma_f32_32x32x16_bf16, 32 samples x 32 columns per wave)
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 4, 8, 2, 1, 0) \
  X(MODE_K5, 2, 8, 2, 1, 0) \
  X(MODE_K5, 4, 4, 4, 1, 0) \
  X(MODE_K5, 2, 4, 4, 1, 0) \
  X(MODE_K5, 2, 8, 2, 1, 1) \
  X(MODE_K5, 2, 4, 4, 1, 1)
namespace ldp {
int tconv_launch_split(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LIST(LDP_CASE_S)
    default: return -100;
  }
}
int tconv_init_split() {
  LIST(LDP_INIT_S)
  return 0;
}
}  // namespace ldp
