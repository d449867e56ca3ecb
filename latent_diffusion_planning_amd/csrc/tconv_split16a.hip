// 16-row split tiles on three bf16 planes (tconv SPLIT = 1, MB = 1), part a
// (the instantiations the default regimes and the bf16-plane fallback launch: profiles/r05_plans_used.txt, tools/r5/plans_used.py)
#include "tconv_inst.hpp"
#define LIST16(X) \
  X(MODE_K5, 8, 4, 1, 2, 0) \
  X(MODE_K5, 8, 4, 1, 2, 1) \
  X(MODE_K5, 4, 2, 2, 2, 0) \
  X(MODE_DOWN, 4, 4, 1, 2, 0) \
  X(MODE_UP, 8, 4, 1, 2, 0) \
  X(MODE_UP, 16, 2, 2, 2, 0)
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
