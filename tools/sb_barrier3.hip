// sb_barrier3.hip -- hand-off between work-groups of one XCD inside a launch: which load flavour is (a) never served
// from a stale line of the consumer CU's vector L1 and (b) still right when the producer's dirty line may be evicted
// from the L2 between the store and the load (other work-groups of the XCD stream read-only data all the time)?
// 16 "sample blocks" x 16 work-groups (block id % 8 = XCD).  Members 0..NP-1 hand off: each round a member rewrites
// ITS chunk (same addresses every round: the consumer's L1 holds last round's copy), barrier, reads the next member's
// chunk, barrier.  Members NP..15 only stream.  A consumer optionally waits `gap` x 64 cycles before reading.
//   hipcc -O2 --offload-arch=gfx950 tools/sb_barrier3.hip -o tools/bin/sb_barrier3 && tools/bin/sb_barrier3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NP = 8;

template <int AUX>
__device__ __forceinline__ f4 bload(__amdgpu_buffer_rsrc_t rs, int off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
}

__global__ __launch_bounds__(512) void rounds(float* buf, int chunk, const float* wts, size_t wts_floats, unsigned* counters,
                                              unsigned* bad, int nrounds, unsigned base, int lmode, int smode, int press_kb, int gap, int share) {
  const int sb = blockIdx.x, m = blockIdx.y;
  unsigned* ctr = counters + sb * 32;
  float acc = 0.f;
  if (m >= NP) {            // streamer: keeps the XCD's L2 turning over for about as long as the others run
    const size_t per = (size_t)press_kb * 256;
    if (per == 0) return;
    for (int r = 0; r < nrounds; ++r) {
      const size_t off = ((size_t)((sb * 16 + m) * 131 + r * 7919) * per) % (wts_floats - per);
      for (size_t i = threadIdx.x * 4; i < per; i += 512 * 4) { const f4 v = *reinterpret_cast<const f4*>(wts + off + i); acc += v[0]; }
    }
    if (acc == 12345.f) buf[0] = acc;
    return;
  }
  const int me = sb * NP + m, nb = share ? sb * NP + (m & ~1) : sb * NP + ((m + 3) % NP);
  float* mine = share ? buf + (size_t)(sb * NP + (m & ~1)) * chunk : buf + (size_t)me * chunk;
  const float* theirs = buf + (size_t)nb * chunk;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(theirs), 0, 0x7fffffff, 0x00020000);
  unsigned phase = 0;
  auto barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++phase;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = base + (unsigned)NP * phase;
      int spin = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (++spin > (1 << 22)) { *bad = 0xdeadu; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  };
  for (int r = 0; r < nrounds; ++r) {
    const float tagv = (float)(r * 256 + me), want = (float)(r * 256 + nb);
    for (int i = threadIdx.x; i < chunk; i += 512) {
      if (share && ((i >> 4) & 1) != (m & 1)) continue;          // my 64-byte half of every line only
      if (smode == 0) mine[i] = tagv;
      else __hip_atomic_store(mine + i, tagv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    barrier();
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(1);
    for (int i = threadIdx.x; i < chunk / 4; i += 512) {
      f4 v;
      switch (lmode) {
        case 0: v = reinterpret_cast<const f4*>(theirs)[i]; break;                 // plain
        case 1: v = bload<1>(rs, i * 16); break;                                   // sc0
        case 2: v = bload<2>(rs, i * 16); break;                                   // nt
        case 3: v = bload<3>(rs, i * 16); break;                                   // sc0 nt
        case 4: v = bload<16>(rs, i * 16); break;                                  // sc1
        case 5: v = bload<17>(rs, i * 16); break;                                  // sc0 sc1
        case 6: v = bload<18>(rs, i * 16); break;                                  // sc1 nt
        default: v = bload<19>(rs, i * 16); break;                                 // sc0 sc1 nt
      }
      if (share) {
        const int owner = (sb * NP + (m & ~1)) + ((i >> 2) & 1);       // float4 i covers floats 4i..4i+3: half = (4i >> 4) & 1
        const float w2 = (float)(r * 256 + owner);
        if (v[0] != w2 || v[1] != w2 || v[2] != w2 || v[3] != w2) atomicAdd(bad, 1u);
      } else if (v[0] != want || v[1] != want || v[2] != want || v[3] != want) atomicAdd(bad, 1u);
    }
    barrier();
  }
  if (acc == 12345.f) buf[0] = acc;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float *buf, *wts; unsigned *ctr, *bad;
  const size_t wts_floats = (size_t)96 << 20;
  const int max_chunk = 16384;
  CK(hipMalloc(&buf, (size_t)128 * max_chunk * 4)); CK(hipMalloc(&ctr, 16 * 32 * 4)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&wts, wts_floats * 4)); CK(hipMemset(wts, 0, wts_floats * 4));
  CK(hipMemset(ctr, 0, 16 * 32 * 4)); CK(hipMemset(bad, 0, 4));
  const int R = 1500;
  unsigned base = 0;
  const char* ln[] = {"plain", "sc0", "nt", "sc0 nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
  for (int share : {1})
  for (int chunk : {1024, 4096})
    for (int press : {0, 1024})
      for (int smode : {0, 1})
        for (int gap : {0, 40})
          for (int lmode = 0; lmode < 8; ++lmode) {
            double t0 = now_us();
            hipLaunchKernelGGL(rounds, dim3(16, 16), dim3(512), 0, s, buf, chunk, wts, wts_floats, ctr, bad, R, base, lmode, smode, press, gap, share);
            CK(hipStreamSynchronize(s));
            base += 2u * NP * R;
            unsigned b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
            std::printf("share %d chunk %2d KB press %4d KB stores %s gap %2d loads %-10s : %6.2f us/round, bad float4s %u\n", share, chunk / 256, press,
                        smode ? "sc1  " : "plain", gap, ln[lmode], (now_us() - t0) / R, b);
            CK(hipMemset(bad, 0, 4));
          }
  return 0;
}
