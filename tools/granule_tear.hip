// granule_tear.hip -- is an 8-byte {value, tag} granule ever observed torn?  Writers keep rewriting granules with
// (tag = i, value = hash(i)); readers poll them and check value == hash(tag).  Pairs on the same XCD (blocks b, b + 8)
// and on different XCDs (blocks b, b + 1); agent-scope (sc1, written through) and plain stores; the rest of the GPU
// either idle or streaming 1 GB in a loop.
//   hipcc -O2 --offload-arch=gfx950 tools/granule_tear.hip -o tools/bin/granule_tear && tools/bin/granule_tear
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned hsh(unsigned i) { return i * 2654435761u ^ 0x5bd1e995u; }
// grid: 16 blocks. block b < 8 writes granule set b; block 8 + b reads the set of writer (b + shift) % 8.
__global__ __launch_bounds__(64) void k(unsigned long long* g, unsigned* bad, unsigned* reads, int iters, int shift, int plain) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b < 8) {
    unsigned long long* mine = g + b * 64;                  // 64 granules = 512 B = 4 lines per writer
    for (int i = 1; i <= iters; ++i) {
      const unsigned long long v = ((unsigned long long)i << 32) | hsh(i + lane);
      if ((plain & 2) && lane != 0) { __builtin_amdgcn_s_sleep(2); continue; }
      if (plain & 1) mine[lane] = v; else __hip_atomic_store(mine + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(2);
    }
  } else {
    const unsigned long long* src = g + ((b - 8 + shift) % 8) * 64;
    unsigned nb = 0, nr = 0, last = 0;
    for (int i = 0; i < iters * 4; ++i) {
      const unsigned long long v = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned tag = (unsigned)(v >> 32), val = (unsigned)v;
      if (tag != 0 && val != hsh(tag + lane)) ++nb;
      if (tag != last) { ++nr; last = tag; }
      if (tag >= (unsigned)iters) break;
    }
    atomicAdd(bad, nb); atomicAdd(reads, nr);
  }
}
__global__ void stream(const float4* p, size_t n, float* out, int reps) {
  float s = 0;
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x; }
  if (s == 1.2345e33f) out[0] = s;
}
int main() {
  unsigned long long* g; unsigned *bad, *reads; float* big; float* out;
  CK(hipMalloc(&g, 8 * 64 * 8)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&reads, 4)); CK(hipMalloc(&out, 4));
  CK(hipMalloc(&big, (size_t)1 << 30)); CK(hipMemset(big, 0, (size_t)1 << 30));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int load : {0, 1}) for (int plain : {0, 2}) for (int shift : {0, 1}) {
    CK(hipMemset(g, 0, 8 * 64 * 8)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(reads, 0, 4));
    if (load) hipLaunchKernelGGL(stream, dim3(960), dim3(256), 0, s2, (const float4*)big, ((size_t)1 << 30) / 16, out, 6);
    // block b -> XCD b % 8: reader 8 + b shares the XCD of writer b (shift 0) or sits on the next one's (shift 1)
    hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, s1, g, bad, reads, 200000, shift, plain);
    CK(hipDeviceSynchronize());
    unsigned hb, hr; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hr, reads, 4, hipMemcpyDeviceToHost));
    std::printf("%s, %s stores, reader on %s XCD: %u torn granules in %u distinct observations\n", load ? "GPU streaming 1 GB" : "GPU otherwise idle",
                plain == 2 ? "agent-scope single-lane" : plain ? "plain" : "agent-scope", shift ? "another" : "the writer's", hb, hr);
  }
  return 0;
}
