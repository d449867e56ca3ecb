// l2_retention.hip -- does an XCD's L2 keep what its work-groups wrote across a kernel boundary in a graph?
// Chain of (writer, reader) kernel pairs; each of 256 work-groups writes / reads a 64 KB region.  The reader of
// block b reads the region written by block b + shift: shift 0 = same XCD (block id % 8), shift 1 = the next XCD.
//   hipcc -O2 --offload-arch=gfx950 tools/l2_retention.hip -o tools/bin/l2_retention && tools/bin/l2_retention
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int REGION = 16384;   // floats = 64 KB
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void writer(float* buf, int it, int nt) {
  f4* p = reinterpret_cast<f4*>(buf + (size_t)blockIdx.x * REGION);
  const f4 v = f4{(float)it, 1.f, 2.f, 3.f};
  if (nt == 1) for (int i = threadIdx.x; i < REGION / 4; i += blockDim.x) __builtin_nontemporal_store(v, p + i);
  else if (nt == 2) for (int i = threadIdx.x; i < REGION / 4; i += blockDim.x) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p + i), "v"(v) : "memory");
  else if (nt == 3) for (int i = threadIdx.x; i < REGION / 4; i += blockDim.x) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p + i), "v"(v) : "memory");
  else for (int i = threadIdx.x; i < REGION / 4; i += blockDim.x) p[i] = v;
}
__global__ void reader(const float* buf, float* out, int shift) {
  const int src = (blockIdx.x + shift) % gridDim.x;
  const float4* p = reinterpret_cast<const float4*>(buf + (size_t)src * REGION);
  float s = 0;
  for (int i = threadIdx.x; i < REGION / 4; i += blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[blockIdx.x] = s;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float *buf, *out; CK(hipMalloc(&buf, (size_t)256 * REGION * 4)); CK(hipMalloc(&out, 4096));
  const int N = 1000;
  const char* wn[] = {"plain stores,", "non-temporal stores,", "sc1 (agent-scope write-through) stores,", "sc0 sc1 (system-scope) stores,"};
  for (int nt : {0, 1, 2, 3})
  for (int shift : {0, 1}) {
    for (int only_reader = 0; only_reader < (nt ? 1 : 2); ++only_reader) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) {
        if (!only_reader) hipLaunchKernelGGL(writer, dim3(256), dim3(512), 0, s, buf, i, nt);
        hipLaunchKernelGGL(reader, dim3(256), dim3(512), 0, s, buf, out, shift);
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      double t0 = now_us(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      std::printf("%s shift %d %s: %.2f us per %s\n", wn[nt], shift, only_reader ? "reader only (data never rewritten)" : "writer+reader pair",
                  (now_us() - t0) / N, only_reader ? "launch" : "pair");
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  // do CLEAN lines survive the kernel boundary?  A chain of readers over the same 16 MB (2 MB per XCD: fits its L2) against
  // a chain whose launches alternate between two 16 MB sets (4 MB per XCD together) and one that walks through 480 MB
  // (never in any cache).  MI355X: 2.42 / 3.63 / 5.77 us per launch.
  {
    float* big; CK(hipMalloc(&big, (size_t)512 << 20)); CK(hipMemset(big, 0, (size_t)512 << 20));
    for (int mode : {0, 1, 2}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) {
        const size_t off = mode == 0 ? 0 : mode == 1 ? (size_t)(i & 1) * (64 << 20) / 4 : (size_t)(i % 30) * (16 << 20) / 4;
        hipLaunchKernelGGL(reader, dim3(256), dim3(512), 0, s, big + off, out, 0);
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      double t0 = now_us(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      std::printf("read-only chain, %s: %.2f us per launch\n", mode == 0 ? "same 16 MB every launch" : mode == 1 ? "two 16 MB sets alternating" : "30 sets of 16 MB in turn (480 MB)",
                  (now_us() - t0) / N);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
