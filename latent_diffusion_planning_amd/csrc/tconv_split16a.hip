// 16-row split tiles (TConvCfg SPLIT = 1, MB = 1: v_mfma_f32_16x16x32_bf16 on the fp32 kernel's wave tile), part a of three:
// the instantiation list of tconv_split.hip's LIST16 is spread over three translation units so that `make -j` compiles them
// side by side (one unit took seven minutes)
#include "tconv_inst.hpp"
#define LIST16(X) \
  X(MODE_K5, 8, 4, 1, 2, 0) \
  X(MODE_K5, 8, 4, 2, 2, 1) \
  X(MODE_K5, 4, 4, 1, 2, 1) \
  X(MODE_K5, 2, 8, 1, 4, 0) \
  X(MODE_K5, 2, 4, 1, 2, 1) \
  X(MODE_K5, 4, 2, 2, 2, 0) \
  X(MODE_K5, 4, 2, 4, 2, 1) \
  X(MODE_K5, 4, 8, 1, 8, 0) \
  X(MODE_K5, 8, 4, 1, 4, 1) \
  X(MODE_K5, 4, 4, 1, 8, 0) \
  X(MODE_K5, 4, 2, 2, 4, 1) \
  X(MODE_DOWN, 4, 4, 1, 2, 0) \
  X(MODE_UP, 8, 4, 1, 2, 0)
namespace ldp {
int tconv_launch_split16a(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LIST16(LDP_CASE_S1)
    default: return -100;
  }
}
int tconv_init_split16a() {
  LIST16(LDP_INIT_S1)
  return 0;
}
}  // namespace ldp
