// k = 5 convolutions of the planner on split bf16 operands (TConvCfg SPLIT):
//   LIST   v_mfma_f32_32x32x16_bf16, 32 samples x 32 columns per wave (MB = 2): the T <= 4 tiles
//   LIST16 v_mfma_f32_16x16x32_bf16, the fp32 kernel's 16 x 16 wave tile over 32-channel steps (MB = 1): T = 8 and T = 4 tiles, the
//          256-channel level, the stride-2 and transposed convs between the levels
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 4, 8, 2, 1, 0) \
  X(MODE_K5, 2, 8, 2, 1, 0) \
  X(MODE_K5, 4, 4, 4, 1, 0) \
  X(MODE_K5, 2, 4, 4, 1, 0) \
  X(MODE_K5, 2, 8, 2, 1, 1) \
  X(MODE_K5, 2, 4, 4, 1, 1) \
  X(MODE_K5, 2, 2, 8, 1, 0) \
  X(MODE_K5, 2, 2, 8, 1, 1)
#define LIST16(X) \
  X(MODE_K5, 8, 4, 1, 2, 0) \
  X(MODE_K5, 8, 4, 2, 2, 0) \
  X(MODE_K5, 8, 4, 1, 2, 1) \
  X(MODE_K5, 8, 4, 2, 2, 1) \
  X(MODE_K5, 4, 8, 1, 2, 1) \
  X(MODE_K5, 4, 4, 2, 2, 1) \
  X(MODE_K5, 4, 4, 1, 2, 1) \
  X(MODE_K5, 4, 8, 1, 2, 0) \
  X(MODE_K5, 4, 4, 1, 2, 0) \
  X(MODE_K5, 2, 8, 1, 4, 0) \
  X(MODE_K5, 2, 8, 1, 4, 1) \
  X(MODE_K5, 2, 4, 1, 2, 0) \
  X(MODE_K5, 2, 4, 1, 2, 1) \
  X(MODE_K5, 8, 2, 2, 2, 0) \
  X(MODE_K5, 8, 2, 2, 2, 1) \
  X(MODE_K5, 4, 2, 2, 2, 0) \
  X(MODE_K5, 4, 2, 2, 2, 1) \
  X(MODE_K5, 4, 2, 4, 2, 0) \
  X(MODE_K5, 4, 2, 4, 2, 1) \
  X(MODE_K5, 4, 8, 1, 4, 0) \
  X(MODE_K5, 4, 8, 1, 4, 1) \
  X(MODE_K5, 4, 8, 1, 8, 0) \
  X(MODE_K5, 4, 8, 1, 8, 1) \
  X(MODE_K5, 8, 4, 1, 4, 0) \
  X(MODE_K5, 8, 4, 1, 4, 1) \
  X(MODE_K5, 4, 4, 1, 4, 0) \
  X(MODE_K5, 4, 4, 1, 4, 1) \
  X(MODE_K5, 4, 4, 1, 8, 0) \
  X(MODE_K5, 4, 4, 1, 8, 1) \
  X(MODE_K5, 4, 2, 2, 4, 0) \
  X(MODE_K5, 4, 2, 2, 4, 1) \
  X(MODE_DOWN, 4, 2, 2, 2, 0) \
  X(MODE_DOWN, 2, 4, 1, 2, 0) \
  X(MODE_DOWN, 4, 4, 1, 2, 0) \
  X(MODE_UP, 4, 4, 1, 2, 0) \
  X(MODE_UP, 8, 2, 2, 2, 0) \
  X(MODE_UP, 8, 4, 1, 2, 0) \
  X(MODE_UP, 16, 2, 2, 2, 0)
namespace ldp {
int tconv_launch_split(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LIST(LDP_CASE_S)
    LIST16(LDP_CASE_S1)
    default: return -100;
  }
}
int tconv_init_split() {
  LIST(LDP_INIT_S)
  LIST16(LDP_INIT_S1)
  return 0;
}
}  // namespace ldp
