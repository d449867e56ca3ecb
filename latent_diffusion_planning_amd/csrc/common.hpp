// common.hpp -- error plumbing and small RAII helpers shared by the host side of libldp_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ldp_hip.h"

namespace ldp {

std::string& last_error();                       // thread-local
int fail(int code, const char* fmt, ...);        // formats into last_error(), returns code

#define LDP_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return ::ldp::fail(LDP_EHIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                         hipGetErrorString(_e));                                              \
  } while (0)

#define LDP_TRY(expr)                 \
  do {                                \
    int _r = (expr);                  \
    if (_r != LDP_OK) return _r;      \
  } while (0)

// device allocation that frees itself
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  int alloc(size_t n) {               // grows only
    if (n <= bytes && p) return LDP_OK;
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { p = nullptr; return fail(LDP_ENOMEM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
    bytes = n;
    return LDP_OK;
  }
  float* f() const { return static_cast<float*>(p); }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

int upload(DevBuf& dst, const void* host, size_t bytes, hipStream_t s);

// ---- small kernels (kernels_misc.hip) ------------------------------------------------------
// out[m][n] = act_out( sum_k act_in(in[m][k]) * W[k][n] + bias[n] ),  W in Flax (K, N) layout;
// act: 0 none, 1 mish, 2 relu.  in row stride = ldi, out row stride = ldo.
int dense_launch(const float* in, int ldi, const float* W, int ldw, const float* bias, float* out,
                 int ldo, int M, int K, int N, int act_in, int act_out, hipStream_t s);
// dst (rows, dp) <- src (rows, d), zero padded; and back
int pad_rows_launch(const float* src, float* dst, int64_t rows, int d, int dp, hipStream_t s);
int unpad_rows_launch(const float* src, float* dst, int64_t rows, int d, int dp, hipStream_t s);
// dst (rows, dp): first d columns N(0,1) from Philox (stream id 1), rest 0
int philox_init_launch(float* dst, int64_t rows_per_sample, int B, int d, int dp,
                       const uint64_t* seed_dev, hipStream_t s);
int set_seed_launch(uint64_t* seed_dev, uint64_t seed, int64_t row_offset, uint64_t epoch, hipStream_t s);
int philox_raw_launch(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, uint32_t* out, int64_t n,
                      hipStream_t s);
int philox_normal_launch(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, float* out, int64_t n,
                         hipStream_t s);
int assemble_plan_launch(const float* state, const float* obs_last, float* plan, float* trans, float* x_out,
                         int B, int T, int D, int DP, int ah, hipStream_t s);
int gather_obs_launch(const float* obs_emb, float* cond, float* obs_last, int B, int H, int D, int oh,
                      hipStream_t s);
int mean_sq_diff_launch(const float* a, const float* b, int64_t n, float* out, hipStream_t s);
int reduce_stats_launch(const float* x, int64_t n, float* out4, hipStream_t s);      // out4 = {min, max, mean, population std}
struct AbarTable { float v[256]; };                                               // float32 cumprod of (1 - beta), passed by value
int add_noise_launch(const float* x0, const float* noise, const int* t_dev, const AbarTable& tab, int n_train, float* out,
                     int64_t rows, int width, hipStream_t s);
int normalize_launch(const float* x, float* y, int64_t n, const float* lo, const float* hi, int dim,
                     int normalize, hipStream_t s);
// LayerNorm over the last axis (eps 1e-6, fast variance): y = (x-mean)*rstd*scale+bias
int layernorm_launch(const float* x, float* y, const float* scale, const float* bias, int rows,
                     int dim, hipStream_t s);
// IDM input assembly: h0[r][c] = a[r][:A] . W_a + s_part[r][c] + c_part[k][c]
// and scheduler update for (R, A) tensors (see idm.hip)

}  // namespace ldp
