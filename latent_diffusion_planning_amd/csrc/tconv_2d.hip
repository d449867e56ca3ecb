// 3x3 convolutions over NHWC images for the StableVAE encoder/decoder (stride 1 with halo, stride 2
// with the (0,1) padding of diffusers' Downsample2D), on the same Toeplitz MFMA kernel.
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K3H, 8, 2, 4, 1, 0) \
  X(MODE_K3H, 8, 4, 1, 2, 0) \
  X(MODE_K3H, 4, 2, 4, 1, 0) \
  X(MODE_K3H, 2, 2, 4, 1, 0) \
  X(MODE_K3S, 8, 2, 4, 1, 0) \
  X(MODE_K3S, 8, 4, 1, 2, 0) \
  X(MODE_K3S, 4, 2, 4, 1, 0) \
  X(MODE_K3S, 2, 2, 4, 1, 0)
namespace ldp {
int tconv_launch_2d(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out)) {
    LIST(LDP_CASE)
    default: return -100;
  }
}
int tconv_init_2d() {
  LIST(LDP_INIT)
  return 0;
}
}  // namespace ldp
