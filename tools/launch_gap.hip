// launch_gap.hip -- cost of a dependent kernel boundary on this box, by submission form.
//   hipcc -O2 --offload-arch=gfx950 tools/launch_gap.hip -o gpurun_out/launch_gap && gpurun_out/launch_gap
// Chains of N empty launches (grid x block x dynamic LDS as the conv kernels use) submitted
// (a) eagerly on one stream, (b) as a captured hipGraph; reports us per launch.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

__global__ void empty_kernel(float* p) {
  extern __shared__ float smem[];
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = smem[0];
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int N = 3000;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                         160 * 1024));
  const int grids[] = {256, 128, 512};
  const int blocks[] = {512, 256};
  const int ldss[] = {128 * 1024, 64 * 1024, 0};
  for (int grid : grids)
    for (int block : blocks)
      for (int lds : ldss) {
        // eager
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), lds, s, nullptr);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), lds, s, nullptr);
        double t_sub = now_us() - t0;
        CK(hipStreamSynchronize(s));
        double t_eager = now_us() - t0;
        // graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), lds, s, nullptr);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double t_graph = now_us() - t0;
        std::printf("grid %4d block %4d lds %6d : eager %.2f us/launch (submit %.2f)   graph %.2f us/launch\n", grid,
                    block, lds, t_eager / N, t_sub / N, t_graph / N);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
      }
  return 0;
}
