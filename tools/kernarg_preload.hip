// kernarg_preload.hip -- does preloading the first kernel arguments into SGPRs (gfx950: up to 16 dwords,
// -mllvm -amdgpu-kernarg-preload-count=N) shorten a dependent-launch chain whose kernels start with a load through a
// pointer argument?  A graph of 2000 kernel nodes, 256 work-groups x 512 threads, each: p[i] += 1 (address from args).
//   hipcc -O2 --offload-arch=gfx950 tools/kernarg_preload.hip -o tools/bin/kp0
//   hipcc -O2 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/kernarg_preload.hip -o tools/bin/kp8
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { float* q; int pad[60]; };
__global__ __launch_bounds__(512) void k(float* p, int stride, int add, Big b) {
  const size_t i = (size_t)blockIdx.x * stride + threadIdx.x;
  p[i] += (float)add + (b.q == p ? 1.0f : 0.0f);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* p; CK(hipMalloc(&p, 256 * 512 * 4)); CK(hipMemset(p, 0, 256 * 512 * 4));
  Big b{}; b.q = nullptr;
  const int R = 2000;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, s, p, 512, 1, b);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; ++rep) {
    double t0 = now_us(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    if (rep) std::printf("graph of %d dependent kernels: %.3f us per kernel\n", R, (now_us() - t0) / R);
  }
  return 0;
}
