// sb_barrier.hip -- what would a layer hand-off cost inside ONE persistent launch, against the 1.55 us graph
// kernel boundary (+ ~1.4 us until the first activation tile is back)?  256 work-groups of 512 threads, 16 per
// "sample block" (block id % 8 = XCD for all 16, as with ConvArgs::by_sample).  Each round: every work-group writes
// 8 KB, the 16 of a sample block meet at a counter barrier (release: drain stores; acquire: invalidate L1), then each
// reads the 8 KB of its right-hand neighbour in the sample block.  Reports us per round and checks the data.
//   hipcc -O2 --offload-arch=gfx950 tools/sb_barrier.hip -o tools/bin/sb_barrier && tools/bin/sb_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int CHUNK = 2048;   // floats per work-group per round (8 KB)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void rounds(float* buf, unsigned* counters, unsigned* bad, int nrounds, unsigned base, int mode) {
  const int sb = blockIdx.x, m = blockIdx.y;                 // 16 sample blocks x 16 members; linear id = sb + 16 m -> XCD sb % 8
  const int me = sb * 16 + m, nb = sb * 16 + ((m + 1) & 15);
  unsigned* ctr = counters + sb * 32;                        // one counter per sample block, own cache line
  float acc = 0.f;
  for (int r = 0; r < nrounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * 256 + me) * CHUNK;
    for (int i = threadIdx.x; i < CHUNK / 4; i += blockDim.x)
      reinterpret_cast<f4*>(mine)[i] = f4{(float)(r + me), 1.f, 2.f, 3.f};
    // ---- release: this wave's stores have reached L2
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
    // ---- acquire: drop this CU's L1 copies of the neighbour's buffer (it held round r-2's data)
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (mode == 1) asm volatile("buffer_inv sc1" ::: "memory");
    else if (mode == 2) { if (threadIdx.x < 64) asm volatile("buffer_inv sc1" ::: "memory"); __syncthreads(); }
    else if (mode == 3) { if (threadIdx.x < 64) asm volatile("buffer_inv sc0" ::: "memory"); __syncthreads(); }
    const float* theirs = buf + ((size_t)(r & 1) * 256 + nb) * CHUNK;
    for (int i = threadIdx.x; i < CHUNK / 4; i += blockDim.x) {
      f4 v;
      if (mode == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const f4*>(theirs) + i) : "memory");
      else if (mode == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(reinterpret_cast<const f4*>(theirs) + i) : "memory");
      else v = reinterpret_cast<const f4*>(theirs)[i];
      if (v[0] != (float)(r + nb)) *bad = 1u + r;
      acc += v[1];
    }
  }
  if (acc == 12345.f) buf[0] = acc;
}
__global__ __launch_bounds__(512) void one_round(float* buf, unsigned* bad, int r) {        // the same work as separate launches
  const int sb = blockIdx.x, m = blockIdx.y;
  const int me = sb * 16 + m, nb = sb * 16 + ((m + 1) & 15);
  float acc = 0.f;
  if (r > 0) {
    const float* theirs = buf + ((size_t)((r - 1) & 1) * 256 + nb) * CHUNK;
    for (int i = threadIdx.x; i < CHUNK / 4; i += blockDim.x) {
      const f4 v = reinterpret_cast<const f4*>(theirs)[i];
      if (v[0] != (float)(r - 1 + nb)) *bad = 1u + r;
      acc += v[1];
    }
  }
  float* mine = buf + ((size_t)(r & 1) * 256 + me) * CHUNK;
  for (int i = threadIdx.x; i < CHUNK / 4; i += blockDim.x) reinterpret_cast<f4*>(mine)[i] = f4{(float)(r + me), 1.f, 2.f, 3.f};
  if (acc == 12345.f) buf[0] = acc;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* buf; unsigned *ctr, *bad;
  CK(hipMalloc(&buf, (size_t)2 * 256 * CHUNK * 4)); CK(hipMalloc(&ctr, 16 * 32 * 4)); CK(hipMalloc(&bad, 4));
  CK(hipMemset(ctr, 0, 16 * 32 * 4)); CK(hipMemset(bad, 0, 4));
  const int R = 3000;
  unsigned base = 0;
  const char* names[] = {"compiler agent-scope fences", "vmcnt(0) + buffer_inv sc1 (every wave)", "vmcnt(0) + buffer_inv sc1 (one wave)",
                         "vmcnt(0) + buffer_inv sc0 (one wave)", "vmcnt(0), consumer loads sc1", "vmcnt(0), consumer loads sc0 sc1"};
  for (int mode : {0, 1, 2, 3, 4, 5}) {
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now_us();
      hipLaunchKernelGGL(rounds, dim3(16, 16), dim3(512), 0, s, buf, ctr, bad, R, base, mode);
      CK(hipStreamSynchronize(s));
      base += 16u * R;
      unsigned b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
      if (rep) std::printf("persistent launch, %s: %.2f us per round, check %s (%u)\n",
                           names[mode], (now_us() - t0) / R, b ? "FAILED" : "ok", b);
      CK(hipMemset(bad, 0, 4));
    }
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(one_round, dim3(16, 16), dim3(512), 0, s, buf, bad, r);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  double t0 = now_us(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  unsigned b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
  std::printf("one graph kernel node per round: %.2f us per round, check %s\n", (now_us() - t0) / R, b ? "FAILED" : "ok");
  return 0;
}
