// 16-row split tiles on two fp16 planes, three products (tconv SPLIT = 3), part b
// (the instantiations the default regimes launch: profiles/r05_plans_used.txt, tools/r5/plans_used.py)
#include "tconv_inst.hpp"
#define LISTH(X) \
  X(MODE_K5, 8, 2, 2, 2, 0, 1) \
  X(MODE_K5, 8, 2, 2, 2, 1, 1) \
  X(MODE_K5, 8, 2, 4, 2, 0, 1) \
  X(MODE_K5, 4, 2, 4, 2, 1, 1) \
  X(MODE_K5, 4, 2, 2, 2, 0, 1) \
  X(MODE_K5, 4, 2, 2, 2, 1, 1) \
  X(MODE_DOWN, 4, 2, 2, 2, 0, 1) \
  X(MODE_DOWN, 4, 2, 4, 2, 0, 1) \
  X(MODE_UP, 8, 2, 4, 2, 0, 1) \
  X(MODE_DOWN, 2, 4, 1, 2, 0, 1) \
  X(MODE_UP, 4, 4, 1, 2, 0, 1) \
  X(MODE_UP, 8, 2, 2, 2, 0, 1) \
  X(MODE_K5, 16, 2, 2, 2, 0, 1) \
  X(MODE_UP, 16, 2, 2, 2, 0, 1) \
  X(MODE_K3S, 8, 4, 1, 2, 0, 1) \
  X(MODE_K5, 2, 4, 1, 2, 1, 2) \
  X(MODE_K5, 2, 4, 2, 2, 1, 2) \
  X(MODE_K5, 2, 4, 2, 2, 1, 1) \
  X(MODE_K5, 2, 4, 2, 2, 0, 1)
#define LISTH32(X)
namespace ldp {
int tconv_launch_split3b(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws, p.split)) {
    LISTH(LDP_CASE_SH)
    LISTH32(LDP_CASE_SH32)
    default: return -100;
  }
}
int tconv_init_split3b() {
  LISTH(LDP_INIT_SH)
  LISTH32(LDP_INIT_SH32)
  return 0;
}
}  // namespace ldp
