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

// One thread per output element; lanes run along n so W reads are coalesced.  The block's input row is
// activated once into LDS (not once per output: the FiLM call applies Mish to 25 inputs for 14 336 outputs)
// and read back as a broadcast.  k-ordered fmaf chain (the same summation order as an MFMA chain).
__global__ void dense_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ W,
                             int ldw, const float* __restrict__ bias, float* __restrict__ out,
                             int ldo, int M, int K, int N, int act_in, int act_out) {
  extern __shared__ float row_s[];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  const float* row = in + (size_t)m * ldi;
  for (int k = threadIdx.x; k < K; k += blockDim.x) row_s[k] = act_f(row[k], act_in);
  __syncthreads();
  if (n >= N || m >= M) return;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) acc = fmaf(row_s[k], W[(size_t)k * ldw + n], acc);
  if (bias) acc += bias[n];
  out[(size_t)m * ldo + n] = act_f(acc, act_out);
}

int dense_launch(const float* in, int ldi, const float* W, int ldw, const float* bias, float* out,
                 int ldo, int M, int K, int N, int act_in, int act_out, hipStream_t s) {
  if (M <= 0 || N <= 0) return LDP_OK;
  dim3 grid((N + 255) / 256, M);
  if (K > 8192) return fail(LDP_EINVAL, "dense_kernel: K = %d exceeds the 32 KB row buffer", K);
  hipLaunchKernelGGL(dense_kernel, grid, dim3(256), (size_t)K * 4, s, in, ldi, W, ldw, bias, out, ldo, M, K, N,
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
  // seed[1] = first global row; element index = global row * dp + channel (as in the step epilogue)
  if (c < d) v = philox_normal(seed[0], (seed[1] + (uint64_t)(b * rps + e / dp)) * (uint64_t)dp + (uint64_t)c, 0u, 1u);
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
__global__ void set_seed_kernel(uint64_t* p, uint64_t seed, uint64_t row_offset, uint64_t epoch) {
  // The call epoch comes from the host with the launch (never read back from memory: a work-group that saw a stale
  // copy would re-issue an old epoch, and the exchange tags of two calls would collide); all three words are written
  // through at agent scope.
  __hip_atomic_store(&p[0], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p[1], row_offset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p[2], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- test primitives for the in-kernel noise source (ldp_philox_raw / ldp_philox_normal) -------
__global__ void philox_raw_kernel(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream, uint32_t* out,
                                  int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[4];
  philox4x32_10(seed, elem0 + (uint64_t)i, step, stream, w);
  for (int j = 0; j < 4; ++j) out[i * 4 + j] = w[j];
}
__global__ void philox_normal_kernel(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream, float* out,
                                     int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = philox_normal(seed, elem0 + (uint64_t)i, step, stream);
}
int philox_raw_launch(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, uint32_t* out, int64_t n,
                      hipStream_t s) {
  hipLaunchKernelGGL(philox_raw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, elem0, step,
                     stream_id, out, n);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}
int philox_normal_launch(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, float* out, int64_t n,
                         hipStream_t s) {
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, elem0, step,
                     stream_id, out, n);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// ---- plan assembly (agent/ldp_agent.py:478-487) ------------------------------------------------
// plan[b] = [obs_last[b], x[b, 0..ah-1]]  (B, ah+1, D);  transition row (b, i) = [plan[b,i] | plan[b,i+1]]
// x is the padded loop state (B, T, DP); also writes the unpadded x (B, T, D) when x_out != nullptr.
__global__ void assemble_plan_kernel(const float* __restrict__ state, const float* __restrict__ obs_last,
                                     float* __restrict__ plan, float* __restrict__ trans,
                                     float* __restrict__ x_out, int B, int T, int D, int DP, int ah) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int row = (int)(i / D), c = (int)(i % D);
  const int rows = T > ah + 1 ? T : ah + 1;
  if (row >= B * rows) return;
  const int b = row / rows, t = row % rows;
  if (t < T && x_out) x_out[((size_t)b * T + t) * D + c] = state[((size_t)b * T + t) * DP + c];
  if (t <= ah) {
    const float v = t == 0 ? obs_last[(size_t)b * D + c] : state[((size_t)b * T + (t - 1)) * DP + c];
    plan[((size_t)b * (ah + 1) + t) * D + c] = v;
    if (t < ah) trans[((size_t)b * ah + t) * 2 * D + c] = v;            // first half of transition t
    if (t > 0) trans[((size_t)b * ah + (t - 1)) * 2 * D + D + c] = v;   // second half of transition t-1
  }
}
int assemble_plan_launch(const float* state, const float* obs_last, float* plan, float* trans, float* x_out,
                         int B, int T, int D, int DP, int ah, hipStream_t s) {
  const int rows = T > ah + 1 ? T : ah + 1;
  const int64_t n = (int64_t)B * rows * D;
  hipLaunchKernelGGL(assemble_plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, state, obs_last,
                     plan, trans, x_out, B, T, D, DP, ah);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// obs_emb (B, H, D) -> cond (B, oh*D) = first oh frames, obs_last (B, D) = frame oh-1
__global__ void gather_obs_kernel(const float* __restrict__ obs_emb, float* __restrict__ cond,
                                  float* __restrict__ obs_last, int B, int H, int D, int oh) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * oh * D) return;
  const int b = (int)(i / (oh * D)), e = (int)(i % (oh * D));
  const float v = obs_emb[(size_t)b * H * D + e];
  cond[i] = v;
  if (e >= (oh - 1) * D) obs_last[(size_t)b * D + (e - (oh - 1) * D)] = v;
}
int gather_obs_launch(const float* obs_emb, float* cond, float* obs_last, int B, int H, int D, int oh,
                      hipStream_t s) {
  const int64_t n = (int64_t)B * oh * D;
  hipLaunchKernelGGL(gather_obs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, obs_emb, cond,
                     obs_last, B, H, D, oh);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

int set_seed_launch(uint64_t* seed_dev, uint64_t seed, int64_t row_offset, uint64_t epoch, hipStream_t s) {
  hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, s, seed_dev, seed, (uint64_t)row_offset, epoch);
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

// mean((a - b)^2): one work-group of 1024, per-thread strided partial sums in float64, LDS tree in a fixed order
__global__ __launch_bounds__(1024) void mean_sq_diff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int64_t n, float* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double d = (double)a[i] - (double)b[i];
    acc += d * d;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)n);
}

int mean_sq_diff_launch(const float* a, const float* b, int64_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(mean_sq_diff_kernel, dim3(1), dim3(1024), 0, s, a, b, n, out);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// (min, max, mean, population std) of n floats -> out[0..3]: one work-group of 1024; sums in float64, LDS trees in a fixed order; the
// deviations in a second pass over the data (jnp.std's definition, not E[x^2] - E[x]^2)
__global__ __launch_bounds__(1024) void reduce_stats_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ double red[1024];
  __shared__ float rmin[1024], rmax[1024];
  double acc = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = x[i];
    acc += (double)v;
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  red[threadIdx.x] = acc; rmin[threadIdx.x] = lo; rmax[threadIdx.x] = hi;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] += red[threadIdx.x + w];
      rmin[threadIdx.x] = fminf(rmin[threadIdx.x], rmin[threadIdx.x + w]);
      rmax[threadIdx.x] = fmaxf(rmax[threadIdx.x], rmax[threadIdx.x + w]);
    }
    __syncthreads();
  }
  const double mean = red[0] / (double)n;
  const float mn = rmin[0], mx = rmax[0];
  __syncthreads();
  acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double d = (double)x[i] - mean;
    acc += d * d;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = mn; out[1] = mx; out[2] = (float)mean; out[3] = (float)sqrt(red[0] / (double)n);
  }
}

int reduce_stats_launch(const float* x, int64_t n, float* out4, hipStream_t s) {
  hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(1024), 0, s, x, n, out4);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// FlaxDDPMScheduler.add_noise: out[r][c] = sqrt(abar[t[r]]) * x0[r][c] + sqrt(1 - abar[t[r]]) * noise[r][c], float32 throughout
// (diffusers scheduling_utils_flax.py get_sqrt_alpha_prod / add_noise_common); abar by value (<= 256 training steps)
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int* __restrict__ t,
                                 const AbarTable tab, int n_train, float* __restrict__ out, int64_t total, int width) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int k = t[i / width];
  k = k < 0 ? 0 : (k >= n_train ? n_train - 1 : k);
  const float a = tab.v[k];
  out[i] = sqrtf(a) * x0[i] + sqrtf(1.0f - a) * noise[i];
}

int add_noise_launch(const float* x0, const float* noise, const int* t_dev, const AbarTable& tab, int n_train, float* out,
                     int64_t rows, int width, hipStream_t s) {
  const int64_t total = rows * width;
  if (total <= 0) return LDP_OK;
  hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x0, noise, t_dev, tab, n_train, out,
                     total, width);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
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
