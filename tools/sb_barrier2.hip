// sb_barrier2.hip -- which store/load flavours make the XCD-local hand-off of sb_barrier.hip CORRECT when the L2 is under
// pressure (every work-group also streams `press` KB of read-only data per round, as the planner's weights do)?
//   hipcc -O2 --offload-arch=gfx950 tools/sb_barrier2.hip -o tools/bin/sb_barrier2 && tools/bin/sb_barrier2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int CHUNK = 4096;   // floats per work-group per round (16 KB)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// smode: 0 plain dword stores, 1 plain dwordx4, 2 sc1 dword (agent-scope atomic store), 3 sc0 sc1 dword
// lmode: 0 global_load sc1 (asm), 1 buffer_load sc1 (builtin), 2 buffer_load sc0 sc1, 3 plain (expected to fail: L1)
__global__ __launch_bounds__(512) void rounds(float* buf, const float* wts, size_t wts_floats, unsigned* counters, unsigned* bad,
                                              int nrounds, unsigned base, int smode, int lmode, int press_kb, int delay) {
  const int sb = blockIdx.x, m = blockIdx.y;
  const int me = sb * 16 + m, nb = sb * 16 + ((m + 5) & 15);
  unsigned* ctr = counters + sb * 32;
  float acc = 0.f;
  for (int r = 0; r < nrounds; ++r) {
    // pressure: stream press_kb of read-only data (distinct per work-group and round)
    if (!(delay & 4)) {
      const size_t per = (size_t)press_kb * 256;       // floats
      const size_t off = ((size_t)(me * 131 + r * 7919) * per) % (wts_floats - per);
      for (size_t i = threadIdx.x * 4; i < per; i += 512 * 4) { const f4 v = *reinterpret_cast<const f4*>(wts + off + i); acc += v[0]; }
    }
    float* mine = buf + ((size_t)(r & 1) * 256 + me) * CHUNK;
    const float tagv = (float)(r * 256 + me);
    if (smode == 1) {
      for (int i = threadIdx.x; i < CHUNK / 4; i += 512) reinterpret_cast<f4*>(mine)[i] = f4{tagv, tagv, tagv, tagv};
    } else {
      for (int i = threadIdx.x; i < CHUNK; i += 512) {
        if (smode == 0) mine[i] = tagv;
        else if (smode == 2) __hip_atomic_store(mine + i, tagv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(mine + i, tagv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (delay & 1) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = base + 16u * (unsigned)(r + 1);
      int spin = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (++spin > (1 << 22)) { *bad = 0xdeadu; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (delay & 2) __builtin_amdgcn_s_sleep(127);
    if (delay & 4) {        // ... or between the hand-off and the consumer's loads: the producer's dirty lines get evicted first
      const size_t per = (size_t)press_kb * 256;
      const size_t off = ((size_t)(me * 131 + r * 7919) * per) % (wts_floats - per);
      for (size_t i = threadIdx.x * 4; i < per; i += 512 * 4) { const f4 v = *reinterpret_cast<const f4*>(wts + off + i); acc += v[0]; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const float* theirs = buf + ((size_t)(r & 1) * 256 + nb) * CHUNK;
    const float want = (float)(r * 256 + nb);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(theirs), 0, 0x7fffffff, 0x00020000);
    for (int i = threadIdx.x; i < CHUNK / 4; i += 512) {
      f4 v;
      if (lmode == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const f4*>(theirs) + i) : "memory");
      else if (lmode == 1) v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16));
      else if (lmode == 2) v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 17));
      else v = reinterpret_cast<const f4*>(theirs)[i];
      if (v[0] != want || v[1] != want || v[2] != want || v[3] != want) atomicAdd(bad, 1u);
    }
  }
  if (acc == 12345.f) buf[0] = acc;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float *buf, *wts; unsigned *ctr, *bad;
  const size_t wts_floats = (size_t)96 << 20;     // 384 MB of read-only data
  CK(hipMalloc(&buf, (size_t)2 * 256 * CHUNK * 4)); CK(hipMalloc(&ctr, 16 * 32 * 4)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&wts, wts_floats * 4)); CK(hipMemset(wts, 0, wts_floats * 4));
  CK(hipMemset(ctr, 0, 16 * 32 * 4)); CK(hipMemset(bad, 0, 4));
  const int R = 2000;
  unsigned base = 0;
  for (int press : {512, 2048})
    for (int smode : {0, 2})
      for (int lmode : {0, 1, 2, 3})
        for (int delay : {4}) {
          double t0 = now_us();
          hipLaunchKernelGGL(rounds, dim3(16, 16), dim3(512), 0, s, buf, wts, wts_floats, ctr, bad, R, base, smode, lmode, press, delay);
          CK(hipStreamSynchronize(s));
          base += 16u * R;
          unsigned b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
          std::printf("press %3d KB  stores %d  loads %d  delay %d : %.2f us/round, mismatching float4s %u\n", press, smode, lmode, delay, (now_us() - t0) / R, b);
          CK(hipMemset(bad, 0, 4));
        }
  return 0;
}
