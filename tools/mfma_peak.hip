// mfma_peak.hip -- what does v_mfma_f32_16x16x4_f32 sustain on this part?  256..1024 work-groups x 8 waves (2 per SIMD),
// each wave issuing 8 independent accumulator chains back to back, nothing else in the loop.  The guide's 157.3 TFLOP/s
// is 256 CUs x 256 FLOP/clk x 2.4 GHz; this prints the rate measured over ~1 ms and ~50 ms of solid MFMA.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wgs : {256, 512}) for (int iters : {2000, 100000}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = (double)wgs * 8 * iters * 32 * (16 * 16 * 4 * 2);
      if (rep) std::printf("%4d work-groups, %6d x 32 MFMAs per wave: %8.3f ms, %6.1f TFLOP/s (%.1f %% of 157.3)\n", wgs, iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
    }
  }
  return 0;
}
