// k=5 convolutions (Conv1dBlock): (T_out, NWN, KS, CPI); pred_horizon 8 uses (T,C) = (8,256) (4,512) (2,1024) (2,512) (4,256), pred_horizon 16 adds (16,256) (8,512) (4,1024)
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 8, 2, 4, 1, 0) \
  X(MODE_K5, 8, 2, 2, 1, 0) \
  X(MODE_K5, 4, 4, 2, 2, 0) \
  X(MODE_K5, 4, 2, 4, 2, 0) \
  X(MODE_K5, 2, 8, 1, 4, 0) \
  X(MODE_K5, 2, 4, 2, 4, 0) \
  X(MODE_K5, 16, 2, 2, 1, 0) \
  X(MODE_K5, 16, 1, 4, 1, 0) \
  X(MODE_K5, 8, 4, 2, 1, 0) \
  X(MODE_K5, 4, 8, 1, 2, 0) \
  X(MODE_K5, 8, 1, 8, 1, 0) \
  X(MODE_K5, 4, 1, 8, 1, 0) \
  X(MODE_K5, 2, 2, 4, 2, 0)
// two row blocks per work-group (batches that fill the chip twice over): weight stream halved
#define LIST2(X) \
  X(MODE_K5, 4, 4, 2, 2, 0) \
  X(MODE_K5, 2, 8, 1, 4, 0) \
  X(MODE_K5, 2, 4, 2, 4, 0) \
  X(MODE_K5, 4, 2, 4, 2, 0) \
  X(MODE_K5, 2, 2, 4, 2, 0)
// small-batch plans compiled with the K-split-over-work-groups path
#define LIST3(X) \
  X(MODE_K5, 8, 1, 8, 1, 0) \
  X(MODE_K5, 2, 2, 4, 2, 0) \
  X(MODE_K5, 4, 1, 8, 1, 0) \
  X(MODE_K5, 4, 2, 4, 2, 0)
namespace ldp {
int tconv_launch_k5(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws)) {
    LIST(LDP_CASE)
    LIST2(LDP_CASE2)
    LIST3(LDP_CASE3)
    default: return -100;
  }
}
int tconv_init_k5() {
  LIST(LDP_INIT)
  LIST2(LDP_INIT2)
  LIST3(LDP_INIT3)
  return 0;
}
}  // namespace ldp
