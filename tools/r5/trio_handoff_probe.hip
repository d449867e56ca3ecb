// trio_handoff_probe.hip -- VERDICT r4, next-round item 5a: what does it cost to hand a sample block's activation tile from layer to layer
// INSIDE a launch, on the real shape of the two "trios" of light launches of an evaluation at 256 plans (#19-21: 512 -> 512 at T = 2; #24-26:
// 256 -> 256 at T = 4)?  A sample block = 16 samples; its tile at those levels is 16 x 2 x 512 (or 16 x 4 x 256) fp32 = 64 KB, produced as 16
// slices of 4 KB by the 16 work-groups of the sample block (all on one XCD: block id % 8 = sample block % 8, the by_sample placement) and needed
// WHOLE by each of them for the next layer.
//   mode G: the tagged-granule path the library trusts for K-partials and statistics ({value, tag ^ value} 8-byte granules, 16-byte sc1 stores,
//           all polls of a thread issued together, s_waitcnt vmcnt(0) before any is inspected, re-poll until every tag is current);
//   mode F: plain fp32 data written through (sc1 stores), drained (vmcnt(0)), then ONE flag granule per slice; consumers poll the 16 flags, then
//           read the 64 KB with sc1 loads;
//   mode K: the baseline -- one kernel per layer in a hipGraph (kernel boundary = release / acquire), plain stores and loads.
// Every consumer verifies every value it reads (bad = 0 or the probe says so).
//   hipcc -O2 --offload-arch=gfx950 tools/r5/trio_handoff_probe.hip -o /tmp/trio_probe && /tmp/trio_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;
constexpr int NSB = 16, NM = 16, SLICE = 1024, TILE = NM * SLICE;      // floats

__device__ __forceinline__ float val(int round, int sb, int m, int i) { return (float)((round * 31 + sb * 7 + m * 3 + i) & 0xffff); }

// ---- mode G -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mode_g(unsigned long long* gran, unsigned* bad, int rounds, unsigned epoch) {
  const int sb = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    unsigned long long* tile = gran + ((size_t)(r & 1) * NSB + sb) * TILE;
    const unsigned tag = epoch + (unsigned)r + 1u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7ffffff0, 0x00020000);
    {   // publish my slice: 1024 values = 512 stores of two granules
      const int i = 2 * t;
      const unsigned b0 = __float_as_uint(val(r, sb, m, i)), b1 = __float_as_uint(val(r, sb, m, i + 1));
      const u4 gv = {b0, tag ^ b0, b1, tag ^ b1};
      __builtin_amdgcn_raw_buffer_store_b128(gv, rs, (unsigned)((m * SLICE + i) * 8), 0, AUX_SC1);
    }
    // consume the whole tile: 16 loads of two granules per thread, all in flight together
    u4 g[16];
    int spin = 0;
    for (;;) {
#pragma unroll
      for (int q = 0; q < 16; ++q) g[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((q * SLICE + 2 * t) * 8), 0, AUX_SC1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 16; ++q) ok = ok && ((g[q][0] ^ g[q][1]) == tag) && ((g[q][2] ^ g[q][3]) == tag);
      if (ok) break;
      if (++spin > (1 << 20)) { atomicAdd(bad, 1u << 16); break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (__uint_as_float(g[q][0]) != val(r, sb, q, 2 * t) || __uint_as_float(g[q][2]) != val(r, sb, q, 2 * t + 1)) atomicAdd(bad, 1u);
    __syncthreads();      // the next layer's main loop would start here (every wave has its fragments)
  }
}

// ---- mode F -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mode_f(float* data, unsigned long long* flags, unsigned* bad, int rounds, unsigned epoch) {
  const int sb = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
  for (int r = 0; r < rounds; ++r) {
    float* tile = data + ((size_t)(r & 1) * NSB + sb) * TILE;
    unsigned long long* fl = flags + ((size_t)(r & 1) * NSB + sb) * NM;
    const unsigned tag = epoch + (unsigned)r + 1u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7ffffff0, 0x00020000);
    if (t < 256) {
      const int i = 4 * t;
      const u4 v = {__float_as_uint(val(r, sb, m, i)), __float_as_uint(val(r, sb, m, i + 1)), __float_as_uint(val(r, sb, m, i + 2)), __float_as_uint(val(r, sb, m, i + 3))};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((m * SLICE + i) * 4), 0, AUX_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this work-group acknowledged by the memory side
    __syncthreads();
    if (t == 0) __hip_atomic_store(fl + m, ((unsigned long long)(tag ^ 0x5a5au) << 32) | 0x5a5au, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 64) {
      int spin = 0;
      for (;;) {
        unsigned long long f = t < NM ? __hip_atomic_load(fl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool ok = t >= NM || (((unsigned)(f >> 32) ^ (unsigned)f) == tag);
        if (__all(ok)) break;
        if (++spin > (1 << 20)) { if (t == 0) atomicAdd(bad, 1u << 16); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    u4 g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((q * 512 + t) * 16), 0, AUX_SC1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = (q * 512 + t) * 4, mm = e / SLICE, i = e % SLICE;
      if (__uint_as_float(g[q][0]) != val(r, sb, mm, i) || __uint_as_float(g[q][3]) != val(r, sb, mm, i + 3)) atomicAdd(bad, 1u);
    }
    __syncthreads();
  }
}

// ---- mode K: one kernel per layer ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mode_k(const float* prev, float* next, unsigned* bad, int r) {
  const int sb = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
  const f4* tile = reinterpret_cast<const f4*>(prev + (size_t)sb * TILE);
  if (r > 0) {
    f4 g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = tile[q * 512 + t];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = (q * 512 + t) * 4, mm = e / SLICE, i = e % SLICE;
      if (g[q][0] != val(r - 1, sb, mm, i) || g[q][3] != val(r - 1, sb, mm, i + 3)) atomicAdd(bad, 1u);
    }
  }
  if (t < 256) {
    const int i = 4 * t;
    reinterpret_cast<f4*>(next + (size_t)sb * TILE + m * SLICE)[t] = f4{val(r, sb, m, i), val(r, sb, m, i + 1), val(r, sb, m, i + 2), val(r, sb, m, i + 3)};
  }
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long long *gran, *flags; float* data; unsigned* bad;
  CK(hipMalloc(&gran, (size_t)2 * NSB * TILE * 8)); CK(hipMemset(gran, 0, (size_t)2 * NSB * TILE * 8));
  CK(hipMalloc(&data, (size_t)2 * NSB * TILE * 4)); CK(hipMemset(data, 0, (size_t)2 * NSB * TILE * 4));
  CK(hipMalloc(&flags, 2 * NSB * NM * 8)); CK(hipMemset(flags, 0, 2 * NSB * NM * 8));
  CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  const int R = 400;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned epoch = 1000;
  auto report = [&](const char* what, float ms) -> int {
    unsigned b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemset(bad, 0, 4));
    std::printf("%-62s %7.2f us per hand-off   bad values %u, time-outs %u\n", what, ms * 1e3f / R, b & 0xffffu, b >> 16);
    return 0;
  };
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(mode_g, dim3(NSB, NM), dim3(512), 0, s, gran, bad, R, epoch); CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); epoch += R + 7;
    report("G: tagged granules (128 KB polled per work-group)", ms);
    CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(mode_f, dim3(NSB, NM), dim3(512), 0, s, data, flags, bad, R, epoch); CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1)); epoch += R + 7;
    report("F: write-through data + drain + 16 flags, then 64 KB of sc1 loads", ms);
    // K: a graph of R dependent kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < R; ++r)
      hipLaunchKernelGGL(mode_k, dim3(NSB, NM), dim3(512), 0, s, data + (size_t)((r + 1) & 1) * NSB * TILE, data + (size_t)(r & 1) * NSB * TILE, bad, r);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); CK(hipMemset(bad, 0, 4));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
    report("K: one kernel per layer in a hipGraph (plain stores / loads)", ms);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
