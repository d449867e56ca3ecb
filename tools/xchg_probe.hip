// xchg_probe.hip -- the column-split statistics exchange in isolation: 256 work-groups in pairs; per launch every wave's
// lane 0 publishes two 8-byte {value, tag ^ value} granules per sample (agent-scope stores) and then polls its peer's
// granules until they validate, exactly as tconv_kernel does; here the reader also knows what the value must be
// (hash of epoch, step, slot, writer, sample) and counts validated granules with the wrong value.  "Calls" of 30 launches x
// `steps` steps reuse 28 slots; the epoch changes per call.  Run one copy, or two at once (the shared-GPU case).
//   hipcc -O2 --offload-arch=gfx950 tools/xchg_probe.hip -o tools/bin/xchg_probe && (tools/bin/xchg_probe & tools/bin/xchg_probe & wait)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned hsh(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 2654435761u ^ (b + 0x9e3779b9u) * 40503u ^ (c * 2246822519u) ^ (d * 3266489917u);
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return h & 0x3fffffffu;                                     // a positive float's bits when OR-ed with an exponent; any bits do
}
__device__ __forceinline__ unsigned long long pack(unsigned tag, unsigned b) { return ((unsigned long long)(tag ^ b) << 32) | b; }
__global__ __launch_bounds__(512) void layer(unsigned long long* slab, const unsigned long long* ctl, unsigned epoch_arg, int use_ctl, int step, int slot,
                                             unsigned* bad, unsigned* timeouts, int work) {
  const int grp = blockIdx.x, half = blockIdx.y, sb = blockIdx.z;       // 8 x 2 x 16, block id % 8 = grp as in the engine
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // some work first (the main loop): the peers of a pair reach the exchange at slightly different times
  float acc = (float)threadIdx.x;
  for (int i = 0; i < work * (1 + ((grp + half + sb) & 3)); ++i) acc = acc * 1.0000001f + 0.5f;
  const unsigned epoch = use_ctl ? (unsigned)ctl[2] : epoch_arg;
  const unsigned tag = (epoch << 12) + (unsigned)step + 1u;
  unsigned long long* xbase = slab + (size_t)slot * (16 * 8 * 4 * 32) + ((size_t)(sb * 8 + grp) * 4) * 32;
  const int me = (sb * 8 + grp) * 2 + half, peer = (sb * 8 + grp) * 2 + (half ^ 1);
  for (int si = 0; si < 2; ++si) {
    const int sr = wave + 8 * si;
    if (lane == 0) {
      unsigned long long* xme = xbase + half * 32 + sr * 2;
      __hip_atomic_store(&xme[0], pack(tag, hsh(epoch, step * 32 + slot, me * 16 + sr, 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&xme[1], pack(tag, hsh(epoch, step * 32 + slot, me * 16 + sr, 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int si = 0; si < 2; ++si) {
    const int sr = wave + 8 * si;
    const unsigned long long* xp = xbase + (half ^ 1) * 32 + sr * 2;
    unsigned long long g1 = 0, g2 = 0;
    int spin = 0;
    for (;;) {
      g1 = __hip_atomic_load(&xp[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      g2 = __hip_atomic_load(&xp[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((((unsigned)(g1 >> 32)) ^ (unsigned)g1) == tag && (((unsigned)(g2 >> 32)) ^ (unsigned)g2) == tag) break;
      if (++spin > (1 << 20)) { if (lane == 0) atomicAdd(timeouts, 1u); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      if ((unsigned)g1 != hsh(epoch, step * 32 + slot, peer * 16 + sr, 0)) atomicAdd(bad, 1u);
      if ((unsigned)g2 != hsh(epoch, step * 32 + slot, peer * 16 + sr, 1)) atomicAdd(bad, 1u);
    }
  }
  if (acc == 1.2345e33f) bad[1] = 1;
}
__global__ void set_epoch(unsigned long long* ctl, unsigned long long e) {
  __hip_atomic_store(&ctl[2], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main(int argc, char** argv) {
  const int calls = argc > 1 ? atoi(argv[1]) : 300, steps = argc > 2 ? atoi(argv[2]) : 20, use_ctl = argc > 3 ? atoi(argv[3]) : 1, work = argc > 4 ? atoi(argv[4]) : 2000;
  unsigned long long *slab, *ctl; unsigned *bad, *to;
  CK(hipMalloc(&slab, (size_t)28 * 16 * 8 * 4 * 32 * 8)); CK(hipMemset(slab, 0, (size_t)28 * 16 * 8 * 4 * 32 * 8));
  CK(hipMalloc(&ctl, 64)); CK(hipMemset(ctl, 0, 64)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8)); CK(hipMalloc(&to, 4)); CK(hipMemset(to, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned hb = 0, ht = 0;
  for (int c = 1; c <= calls; ++c) {
    hipLaunchKernelGGL(set_epoch, dim3(1), dim3(1), 0, s, ctl, (unsigned long long)c);
    for (int st = 0; st < steps; ++st)
      for (int L = 0; L < 28; ++L)
        hipLaunchKernelGGL(layer, dim3(8, 2, 16), dim3(512), 0, s, slab, ctl, (unsigned)c, use_ctl, st, L, bad, to, work);
    CK(hipStreamSynchronize(s));
  }
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht, to, 4, hipMemcpyDeviceToHost));
  std::printf("%d calls x %d steps x 28 exchanges (epoch from %s): %u validated granules with a wrong value, %u timeouts\n", calls, steps,
              use_ctl ? "device memory" : "the kernel argument", hb, ht);
  return 0;
}
