// kernels_misc.hip -- small HBM-bound helper kernels around the MFMA convolutions:
// table-building dense layer (time MLP / FiLM tables, run once per weight load or once per
// sample() call), state padding, Philox initial noise, min/max normalisation, LayerNorm.
#include "common.hpp"
#include "tconv.hpp"

namespace ldp {

std::string& last_error() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

int upload(DevBuf& dst, const void* host, size_t bytes, hipStream_t s) {
  // synchronous on purpose: `host` is usually a temporary (pageable) buffer
  (void)s;
  LDP_TRY(dst.alloc(bytes));
  LDP_HIP(hipMemcpy(dst.p, host, bytes, hipMemcpyHostToDevice));
  return LDP_OK;
}

__device__ __forceinline__ float act_f(float x, int act) {
  if (act == 1) return mish_f(x);
  if (act == 2) return fmaxf(x, 0.0f);
  return x;
}

// One thread per output element; lanes run along n so W reads are coalesced and in[m][k] is a
// wave-uniform broadcast.  k-ordered fmaf chain (the same summation order as an MFMA chain).
__global__ void dense_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ W,
                             int ldw, const float* __restrict__ bias, float* __restrict__ out,
                             int ldo, int M, int K, int N, int act_in, int act_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (n >= N || m >= M) return;
  const float* row = in + (size_t)m * ldi;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) acc = fmaf(act_f(row[k], act_in), W[(size_t)k * ldw + n], acc);
  if (bias) acc += bias[n];
  out[(size_t)m * ldo + n] = act_f(acc, act_out);
}

int dense_launch(const float* in, int ldi, const float* W, int ldw, const float* bias, float* out,
                 int ldo, int M, int K, int N, int act_in, int act_out, hipStream_t s) {
  if (M <= 0 || N <= 0) return LDP_OK;
  dim3 grid((N + 255) / 256, M);
  hipLaunchKernelGGL(dense_kernel, grid, dim3(256), 0, s, in, ldi, W, ldw, bias, out, ldo, M, K, N,
                     act_in, act_out);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

__global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                int64_t rows, int d, int dp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dp) return;
  const int64_t r = i / dp;
  const int c = (int)(i - r * dp);
  dst[i] = c < d ? src[r * d + c] : 0.0f;
}

__global__ void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                  int64_t rows, int d, int dp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  const int64_t r = i / d;
  const int c = (int)(i - r * d);
  dst[i] = src[r * dp + c];
}

int pad_rows_launch(const float* src, float* dst, int64_t rows, int d, int dp, hipStream_t s) {
  const int64_t n = rows * dp;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst,
                     rows, d, dp);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

int unpad_rows_launch(const float* src, float* dst, int64_t rows, int d, int dp, hipStream_t s) {
  const int64_t n = rows * d;
  hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst,
                     rows, d, dp);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// Initial state x_T ~ N(0, I): element (global row = row_offset + b, t, c) -> Philox stream 1.
// Keyed by the *global* sample index so the draw is independent of how a batch is sharded.
__global__ void philox_init_kernel(float* __restrict__ dst, int64_t rps, int B, int d, int dp,
                                   const uint64_t* __restrict__ seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = rps * dp;
  if (i >= (int64_t)B * per) return;
  const int64_t b = i / per;
  const int64_t e = i - b * per;
  const int c = (int)(e % dp);
  float v = 0.0f;
  if (c < d) v = philox_normal(seed[0], (uint64_t)(seed[1] + b) * (uint64_t)per + (uint64_t)e, 0u, 1u);
  dst[i] = v;
}

int philox_init_launch(float* dst, int64_t rows_per_sample, int B, int d, int dp,
                       const uint64_t* seed_dev, hipStream_t s) {
  const int64_t n = (int64_t)B * rows_per_sample * dp;
  hipLaunchKernelGGL(philox_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst,
                     rows_per_sample, B, d, dp, seed_dev);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// control words: [0] seed, [1] row offset, [2] call epoch (tags of the GroupNorm statistics
// exchange; advanced once per forward / sample call, also under graph replay), [3] fault flag
__global__ void set_seed_kernel(uint64_t* p, uint64_t seed, uint64_t row_offset) {
  p[0] = seed;
  p[1] = row_offset;
  p[2] = p[2] + 1;
}

int set_seed_launch(uint64_t* seed_dev, uint64_t seed, int64_t row_offset, hipStream_t s) {
  hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, s, seed_dev, seed, (uint64_t)row_offset);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// utils/data_utils.py:9-16: same operation order as the reference expression
__global__ void normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                 const float* __restrict__ lo, const float* __restrict__ hi,
                                 int dim, int normalize) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = dim == 1 ? 0 : (int)(i % dim);
  const float l = lo[c], h = hi[c];
  float v = x[i];
  if (normalize == 2) {                       // clip_min / clip_max entries: plain clip
    v = fminf(fmaxf(v, l), h);
  } else if (normalize) {
    v = (v - l) / (h - l) * 2.0f - 1.0f;
  } else {
    v = (v + 1.0f) / 2.0f;
    v = v * (h - l) + l;
    v = fminf(fmaxf(v, l), h);
  }
  y[i] = v;
}

int normalize_launch(const float* x, float* y, int64_t n, const float* lo, const float* hi, int dim,
                     int normalize, hipStream_t s) {
  if (n <= 0) return LDP_OK;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n,
                     lo, hi, dim, normalize);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// one wave per row; dim <= 64 * 16
__global__ void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                 const float* __restrict__ scale, const float* __restrict__ bias,
                                 int rows, int dim) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < dim; c += 64) {
    const float v = xr[c];
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float mean = s1 / (float)dim;
  const float var = fmaxf(s2 / (float)dim - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  for (int c = lane; c < dim; c += 64)
    y[(size_t)row * dim + c] = (xr[c] - mean) * rstd * scale[c] + bias[c];
}

int layernorm_launch(const float* x, float* y, const float* scale, const float* bias, int rows,
                     int dim, hipStream_t s) {
  if (rows <= 0) return LDP_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, scale, bias, rows,
                     dim);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

}  // namespace ldp
