// k=5 convolutions that also emit the block's 1x1 residual projection (shares the A fragments)
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K5, 8, 2, 4, 1, 1) \
  X(MODE_K5, 4, 4, 2, 2, 1) \
  X(MODE_K5, 2, 8, 1, 4, 1) \
  X(MODE_K5, 16, 2, 2, 1, 1) \
  X(MODE_K5, 16, 1, 4, 1, 1) \
  X(MODE_K5, 8, 4, 2, 1, 1) \
  X(MODE_K5, 4, 8, 1, 2, 1) \
  X(MODE_K5, 8, 1, 8, 1, 1) \
  X(MODE_K5, 4, 1, 8, 1, 1) \
  X(MODE_K5, 2, 4, 2, 2, 1) \
  X(MODE_K5, 4, 2, 4, 1, 1) \
  X(MODE_K5, 2, 2, 4, 2, 1) \
  X(MODE_K5, 8, 1, 4, 1, 1)
// two row blocks per work-group (batches that fill the chip twice over): weight stream halved
#define LIST2(X) \
  X(MODE_K5, 2, 8, 1, 4, 1) \
  X(MODE_K5, 2, 4, 2, 2, 1)
// small-batch plans compiled with the K-split-over-work-groups path
#define LIST3(X) \
  X(MODE_K5, 8, 1, 8, 1, 1) \
  X(MODE_K5, 2, 2, 4, 2, 1) \
  X(MODE_K5, 4, 1, 8, 1, 1) \
  X(MODE_K5, 4, 2, 4, 1, 1)
namespace ldp {
int tconv_launch_k5r(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, p.mb, p.kws)) {
    LIST(LDP_CASE)
    LIST2(LDP_CASE2)
    LIST3(LDP_CASE3)
    default: return -100;
  }
}
int tconv_init_k5r() {
  LIST(LDP_INIT)
  LIST2(LDP_INIT2)
  LIST3(LDP_INIT3)
  return 0;
}
}  // namespace ldp
