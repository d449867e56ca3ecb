// icache_probe2.hip -- launch gap (last wave of kernel A done -> first instruction of kernel B) per XCD clock domain,
// for B = a 32 KB straight-line kernel, behind: itself, a small kernel, a kernel that streams S MB through the L2s.
//   hipcc -O3 --offload-arch=gfx950 tools/icache_probe2.hip -o tools/bin/icache_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { unsigned long long t0, t1; unsigned int xcc, pad; };

__device__ __forceinline__ unsigned int xcc_id() {
  unsigned int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

__global__ __launch_bounds__(512) void probe(Rec* rec, float seed, int body) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  const float m = 1.0001f, c = 0.5f;
  if (body) {
    asm volatile(".rept 1024\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n .endr\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) {
    Rec r; r.t0 = t0; r.t1 = t1; r.xcc = xcc_id(); r.pad = 0;
    rec[blockIdx.x * 8 + (threadIdx.x >> 6)] = r;
  }
  if (a0 + a1 + a2 + a3 == 12345.678f) rec[0].pad = 1;
}

__global__ void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, Rec* rec) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) {
    Rec r; r.t0 = t0; r.t1 = t1; r.xcc = xcc_id(); r.pad = 0;
    rec[blockIdx.x * 8 + (threadIdx.x >> 6)] = r;
  }
}

int main() {
  Rec *ra, *rb;
  float *big0, *big1;
  const size_t nrec = 256 * 8, nbig = 64u << 20;
  CHECK(hipMalloc(&ra, nrec * sizeof(Rec)));
  CHECK(hipMalloc(&rb, nrec * sizeof(Rec)));
  CHECK(hipMalloc(&big0, nbig));
  CHECK(hipMalloc(&big1, nbig));
  CHECK(hipMemset(big0, 0, nbig));
  std::vector<Rec> ha(nrec), hb(nrec);
  auto gap = [&](const char* what) {
    hipMemcpy(ha.data(), ra, nrec * sizeof(Rec), hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), rb, nrec * sizeof(Rec), hipMemcpyDeviceToHost);
    std::vector<double> gaps;
    // s_memtime is one counter per clock domain (tens of ms apart, tools/timeline.py): cluster kernel B's entries by
    // value, pair each cluster with kernel A's records of the same domain (the nearest ones before it)
    std::vector<unsigned long long> b0;
    for (auto& r : hb) if (r.t0) b0.push_back(r.t0);
    std::sort(b0.begin(), b0.end());
    size_t i = 0;
    while (i < b0.size()) {
      size_t j = i;
      while (j + 1 < b0.size() && b0[j + 1] - b0[j] < 200000) ++j;
      const unsigned long long startb = b0[i];
      unsigned long long enda = 0;
      for (auto& r : ha) if (r.t1 && r.t1 < startb + 200000 && startb < r.t1 + 40000000ull) enda = std::max(enda, r.t1);
      if (enda) gaps.push_back((double)startb - (double)enda);
      i = j + 1;
    }
    if (gaps.empty()) { printf("%-52s no pairs\n", what); return; }
    std::sort(gaps.begin(), gaps.end());
    printf("%-52s gap ticks per XCD: min %8.0f median %8.0f max %8.0f\n", what, gaps.front(), gaps[gaps.size() / 2], gaps.back());
  };
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(ra, 0, nrec * sizeof(Rec)));
    CHECK(hipMemset(rb, 0, nrec * sizeof(Rec)));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, ra, 1.0f, 1);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, rb, 1.0f, 1);
    CHECK(hipDeviceSynchronize());
    gap("32 KB kernel behind itself");
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, ra, 1.0f, 0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, rb, 1.0f, 0);
    CHECK(hipDeviceSynchronize());
    gap("empty body behind empty body");
    for (size_t mb : {2, 8, 16, 32, 64}) {
      CHECK(hipMemset(ra, 0, nrec * sizeof(Rec)));
      hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, 0, (const float4*)big0, (float4*)big1, (mb << 20) / 16, ra);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, rb, 1.0f, 1);
      CHECK(hipDeviceSynchronize());
      char w[96];
      snprintf(w, sizeof w, "32 KB kernel behind a %zu MB copy (%zu MB per XCD L2)", mb, mb * 2 / 8);
      gap(w);
    }
  }
  return 0;
}
