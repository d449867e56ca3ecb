// 3x3 convolutions over NHWC images for the StableVAE encoder/decoder (stride 1 with halo, stride 2
// with the (0,1) padding of diffusers' Downsample2D), on the same Toeplitz MFMA kernel.
// (TO = 3: the 3-pixel level of 96 x 96 frames -- vae_feature_dim 36, agent/ldp_agent.py:75-77 -- 64-column tiles: 3 x 64 is a whole number of
//  64-lane epilogue rows; the stride-1 tile stages 5 halo pixels, which four waves divide evenly only with four 16-channel sub-chunks per wave)
#include "tconv_inst.hpp"
#define LIST(X) \
  X(MODE_K3H, 8, 2, 4, 1, 0) \
  X(MODE_K3H, 8, 4, 1, 2, 0) \
  X(MODE_K3H, 4, 2, 4, 1, 0) \
  X(MODE_K3H, 2, 2, 4, 1, 0) \
  X(MODE_K3S, 8, 2, 4, 1, 0) \
  X(MODE_K3S, 8, 4, 1, 2, 0) \
  X(MODE_K3S, 4, 2, 4, 1, 0) \
  X(MODE_K3S, 2, 2, 4, 1, 0) \
  X(MODE_K3H, 3, 4, 1, 4, 0) \
  X(MODE_K3S, 3, 4, 1, 2, 0)
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
