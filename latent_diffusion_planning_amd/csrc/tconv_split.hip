// k = 5 convolutions of the planner on split bf16 operands (TConvCfg SPLIT):
//   LIST   v_mfma_f32_32x32x16_bf16, 32 samples x 32 columns per wave (MB = 2): the T <= 4 tiles
//   the 16-row tiles (v_mfma_f32_16x16x32_bf16, the fp32 kernel's 16 x 16 wave tile over 32-channel steps: T = 8 and T = 4 tiles, the
//   256-channel level, the stride-2 and transposed convs between the levels) are instantiated in tconv_split16{a,b,c}.hip (MB = 1) and
//   tconv_split2.hip (MB = 2, two row blocks per wave), one translation unit each so that `make -j` builds them side by side
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 2, 8, 2, 1, 0) \
  X(MODE_K5, 2, 4, 4, 1, 0) \
  X(MODE_K5, 2, 8, 2, 1, 1) \
  X(MODE_K5, 2, 4, 4, 1, 1) \
  X(MODE_K5, 2, 2, 8, 1, 0) \
  X(MODE_K5, 2, 2, 8, 1, 1)
namespace ldp {
int tconv_launch_split16a(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split16b(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split16c(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split2(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split3(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split3b(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_init_split16a();
int tconv_init_split16b();
int tconv_init_split16c();
int tconv_init_split2();
int tconv_init_split3();
int tconv_init_split3b();
int tconv_launch_split(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  if (p.split >= 3) { const int r = tconv_launch_split3(p, a, stream); return r == -100 ? tconv_launch_split3b(p, a, stream) : r; }
  if (p.split == 2) return tconv_launch_split2(p, a, stream);
  if (p.mb == 1) {
    int r = tconv_launch_split16a(p, a, stream);
    if (r == -100) r = tconv_launch_split16b(p, a, stream);
    if (r == -100) r = tconv_launch_split16c(p, a, stream);
    return r;
  }
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LIST(LDP_CASE_S)
    default: return -100;
  }
}
int tconv_init_split() {
  LIST(LDP_INIT_S)
  int r = tconv_init_split16a();
  if (!r) r = tconv_init_split16b();
  if (!r) r = tconv_init_split16c();
  if (!r) r = tconv_init_split2();
  if (!r) r = tconv_init_split3();
  if (!r) r = tconv_init_split3b();
  return r;
}
}  // namespace ldp
