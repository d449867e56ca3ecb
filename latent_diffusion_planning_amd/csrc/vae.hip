// vae.hip -- StableVAE encoder (FlaxAutoencoderKL.encode(...).latent_dist.mean).
// Placeholder translation unit: the encoder kernels land here; until then the entry points fail
// loudly instead of falling back to anything.
#include "engine.hpp"

namespace ldp {
int vae_finalize(ldp_handle*, hipStream_t) {
  return fail(LDP_ESTATE, "the StableVAE encoder kernels are not built in this version of libldp_hip");
}
void vae_destroy(ldp_handle*) {}
}  // namespace ldp

extern "C" int ldp_vae_encode(ldp_handle*, const float*, float*, int32_t, void*) {
  return ldp::fail(LDP_ESTATE, "the StableVAE encoder kernels are not built in this version of libldp_hip");
}
