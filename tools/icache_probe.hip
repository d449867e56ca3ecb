// icache_probe.hip -- how fast does straight-line code run the FIRST time a kernel executes it?
// Round-3 question behind the launch-gap pattern of tools/timeline.py (gaps are 1 us longer behind the layers that
// stream > 4 MB per XCD through the L2): is the instruction cache cold at every launch, and what does a cold line cost
// when the code is still in the L2 / when the L2 has been flushed by a weight stream?
//   hipcc -O3 --offload-arch=gfx950 tools/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
// Each wave times PASSES executions of the same 32 KB block of independent v_fma_f32 (8 bytes each, 4096 of them = 512
// instruction-cache lines); pass 0 is the first touch.  Scenarios: back to back with itself; behind a small kernel;
// behind a kernel that streams 256 MB (every L2 flushed).  Grid 256 x 512 threads like the planner's conv launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int PASSES = 3;

__global__ __launch_bounds__(512) void probe(unsigned long long* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  const float m = 1.0001f, c = 0.5f;
  unsigned long long t[PASSES + 1];
  t[0] = __builtin_amdgcn_s_memtime();
  for (int p = 0; p < PASSES; ++p) {
    asm volatile(
        ".rept 512\n"
        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
        ".endr\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        : "v"(m), "v"(c));
    t[p + 1] = __builtin_amdgcn_s_memtime();
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    for (int p = 0; p < PASSES; ++p) out[((size_t)blockIdx.x * 8 + wave) * PASSES + p] = t[p + 1] - t[p];
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;
}

__global__ void small_kernel(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

__global__ void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  unsigned long long* out;
  float *buf, *big0, *big1;
  const size_t nout = 256 * 8 * PASSES, nbig = 256u << 20;
  CHECK(hipMalloc(&out, nout * 8));
  CHECK(hipMalloc(&buf, 256 * 512 * 4));
  CHECK(hipMalloc(&big0, nbig));
  CHECK(hipMalloc(&big1, nbig));
  CHECK(hipMemset(buf, 0, 256 * 512 * 4));
  CHECK(hipMemset(big0, 0, nbig));
  std::vector<unsigned long long> h(nout);
  auto report = [&](const char* what) {
    hipMemcpy(h.data(), out, nout * 8, hipMemcpyDeviceToHost);
    printf("%-46s", what);
    for (int p = 0; p < PASSES; ++p) {
      std::vector<double> first, all;
      for (int b = 0; b < 256; ++b) {
        double mn = 1e30;
        for (int w = 0; w < 8; ++w) { double v = (double)h[((size_t)b * 8 + w) * PASSES + p]; all.push_back(v); mn = std::min(mn, v); }
        first.push_back(mn);
      }
      std::sort(all.begin(), all.end());
      // s_memtime ticks at 100 MHz on this part? report raw ticks and ns assuming the constant 100 MHz reference is wrong: print ticks
      printf("  pass %d: median %7.0f  p90 %7.0f ticks", p, all[all.size() / 2], all[all.size() * 9 / 10]);
    }
    printf("\n");
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    report("probe right behind probe");
    hipLaunchKernelGGL(small_kernel, dim3(256), dim3(512), 0, 0, buf);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    report("probe behind a small kernel");
    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)big0, (float4*)big1, nbig / 16);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    report("probe behind a 256 MB stream (L2s flushed)");
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    report("probe after an idle device (host sync)");
  }
  printf("4096 v_fma_f32 per pass (32 KB = 512 lines); ideal issue: 4096 x 4 cycles x 2 waves per SIMD\n");
  return 0;
}
