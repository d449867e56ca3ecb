// split tiles on two fp16 planes, three products (tconv SPLIT = 3: 16-row, one or two row blocks per wave; SPLIT = 4: 32-row), part a
// (the instantiations the default regimes launch: profiles/r05_plans_used.txt, tools/r5/plans_used.py)
#include "tconv_inst.hpp"
#define LISTH(X) \
  X(MODE_K5, 4, 8, 1, 2, 0, 1) \
  X(MODE_K5, 4, 8, 1, 2, 1, 1) \
  X(MODE_K5, 4, 8, 1, 2, 0, 2) \
  X(MODE_K5, 8, 4, 1, 2, 0, 1) \
  X(MODE_K5, 8, 4, 1, 2, 1, 1) \
  X(MODE_K5, 4, 4, 1, 2, 0, 1) \
  X(MODE_K5, 4, 4, 2, 2, 1, 1) \
  X(MODE_K5, 4, 4, 2, 2, 0, 1) \
  X(MODE_K5, 4, 2, 4, 2, 0, 1) \
  X(MODE_DOWN, 4, 4, 1, 2, 0, 1) \
  X(MODE_DOWN, 2, 4, 2, 2, 0, 1) \
  X(MODE_UP, 4, 4, 2, 2, 0, 1) \
  X(MODE_UP, 8, 4, 1, 2, 0, 1) \
  X(MODE_K3H, 8, 4, 1, 2, 0, 1) \
  X(MODE_K5, 2, 8, 1, 2, 1, 2) \
  X(MODE_K5, 2, 8, 1, 4, 1, 1) \
  X(MODE_K5, 2, 8, 1, 4, 0, 1)
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
