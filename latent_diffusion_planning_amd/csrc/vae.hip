// vae.hip -- StableVAE (diffusers FlaxAutoencoderKL, model/stable_vae_model.yaml:4-16):
//   ldp_vae_encode = encode(x).latent_dist.mean   (call site agent/ldp_agent.py:55-60)
//   ldp_vae_decode = decode(z).sample             (call site agent/ldp_agent.py:66-85)
// NHWC throughout (the Flax modules are NHWC internally).  The 3x3 convolutions run on the same
// Toeplitz f32-MFMA kernel as the planner (tconv.hpp, MODE_K3H / MODE_K3S: a "sample" is a row tile
// of TO pixels, the three image rows are folded into K); GroupNorm(32, eps 1e-6)+swish is applied by
// HBM-bound element-wise kernels between convolutions (statistics: coalesced two-stage reduction,
// deterministic order); the 4-token single-head attention of the mid block is a tiny VALU kernel.
#include "engine.hpp"
#include "sconv.hpp"

#include <algorithm>

namespace ldp {

namespace {

struct GnW { DevBuf scale, bias; int c = 0; };
struct Res2dW { GnW n1, n2; ConvW c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct AttnW {
  GnW gn;
  DevBuf wq, bq, wk, bk, wv, bv, wo, bo;     // Flax layout (VALU path: odd row counts)
  ConvW qkv, proj;                            // MFMA path: [query | key | value] as one (C -> 3C) 1x1 conv, proj_attn
  int c = 0;
};
struct MidW { Res2dW r0, r1; AttnW at; };
struct DownW { Res2dW r[2]; ConvW ds; bool has_ds = false; };
struct UpW { Res2dW r[3]; ConvW us; bool has_us = false; };

struct VaeState {
  bool enc_ready = false, dec_ready = false;
  int S = 64, LC = 4, G = 32;
  std::vector<int> ch;
  // encoder
  DevBuf cin_w, cin_b;                 // conv_in (3,3,3,C0) direct kernel
  std::vector<DownW> down;
  MidW emid;
  GnW enorm;
  ConvW econv_out;                     // C -> 2*LC (padded to 32 columns)
  DevBuf quant_w, quant_b;             // (2LC, 2LC), (2LC)
  // decoder
  DevBuf pq_w, pq_b;                   // post_quant (LC, LC)
  ConvW dconv_in;                      // LC (padded to 64) -> C
  MidW dmid;
  std::vector<UpW> up;
  GnW dnorm;
  ConvW dconv_out;                     // C0 -> 3 (padded to 32 columns)
  // workspaces for `ws_n` images
  int ws_n = 0;
  DevBuf b0, b1, b2, b3, b4, part, part2, stats, small[5], qkv;
  DevBuf planes, zero;                 // split-operand convs: the normalised input as three bf16 planes; a zero page
  int eout_cols = 32;                  // padded column count of the encoder's conv_out (2 LC real)
};

VaeState* V(ldp_handle* h) { return static_cast<VaeState*>(h->vae); }

// ---------------------------------------------------------------------------------------------
// element-wise / reduction kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float swish_f(float x) { return x / (1.0f + expf(-x)); }

// conv_in: 3x3, pad 1, Cin = 3 (27 MACs per output).  Block = one image row, thread = (4 output channels, pixel phase):
// the thread's 27 x 4 weights live in registers for the whole row (round 3: the first version re-read them per pixel
// and spent its time on 64-bit index divisions -- 366 us per 256 frames against a 100 us output-write floor), the row's
// three input rows (+ zero halo) are staged in LDS once.  blockDim = (C/4) * PPB threads, PPB pixels in flight.
template <int PPB>
__global__ void conv_in3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                const float* __restrict__ b, float* __restrict__ y, float* __restrict__ part,
                                int N, int S, int C) {
  extern __shared__ float rows[];                     // [3][(S + 2) * 3]: input rows hy-1..hy+1 with a zero pixel either side; later 2 floats per thread
  const int cq = C / 4, tid = threadIdx.x;
  const int q = tid % cq, ph = tid / cq;              // channel quad, pixel phase
  const int hy = blockIdx.x % S;
  const int64_t n = blockIdx.x / S;
  const int RW = (S + 2) * 3;
  for (int i = tid; i < 3 * RW; i += blockDim.x) {
    const int dh = i / RW, j = i % RW, ww = j / 3 - 1, hh = hy + dh - 1;
    rows[i] = (hh >= 0 && hh < S && ww >= 0 && ww < S) ? x[((n * S + hh) * S + ww) * 3 + j % 3] : 0.0f;
  }
  float4 wr[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (size_t)k * C + q * 4);
  const float4 bias = *reinterpret_cast<const float4*>(b + q * 4);
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (int wx = ph; wx < S; wx += PPB) {
    float4 a = bias;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int t = 0; t < 9; ++t) {                   // t = dw * 3 + ci: nine consecutive floats of the padded row
        const float v = rows[dh * RW + wx * 3 + t];
        const float4 k = wr[dh * 9 + t];
        a.x = fmaf(v, k.x, a.x); a.y = fmaf(v, k.y, a.y); a.z = fmaf(v, k.z, a.z); a.w = fmaf(v, k.w, a.w);
      }
    *reinterpret_cast<float4*>(y + ((n * S + hy) * S + wx) * C + q * 4) = a;
    s1 += (a.x + a.y) + (a.z + a.w);
    s2 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  // the first GroupNorm's stage-1 sums on the way out (gn_part_kernel's layout with one chunk per image row):
  // part[n][hy][quad][{sum, sum of squares}], pixel phases added in order
  __syncthreads();
  rows[tid * 2] = s1; rows[tid * 2 + 1] = s2;
  __syncthreads();
  if (ph == 0 && part) {
    for (int r = 1; r < PPB; ++r) { s1 += rows[(r * cq + q) * 2]; s2 += rows[(r * cq + q) * 2 + 1]; }
    float* o = part + ((n * S + hy) * cq + q) * 2;
    o[0] = s1; o[1] = s2;
  }
}

// GroupNorm statistics, stage 1: block = (image n, chunk of PCH pixels); coalesced float4 reads of
// whole pixel rows; per (chunk, channel quad) partial (sum, sumsq) -> part[n][chunk][C/4][2]
constexpr int PCH = 256;
__global__ void gn_part_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C) {
  extern __shared__ float sh[];
  const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int cq = C / 4, tid = threadIdx.x, q = tid % cq, pr = tid / cq, npr = blockDim.x / cq;
  const int p0 = chunk * PCH, p1 = min(p0 + PCH, HW);
  float s1 = 0.f, s2 = 0.f;
  for (int p = p0 + pr; p < p1; p += npr) {
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)n * HW + p) * C + q * 4);
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  sh[tid * 2] = s1; sh[tid * 2 + 1] = s2;
  __syncthreads();
  if (pr == 0) {
    for (int r = 1; r < npr; ++r) { s1 += sh[(r * cq + q) * 2]; s2 += sh[(r * cq + q) * 2 + 1]; }
    float* o = part + (((size_t)n * nchunk + chunk) * cq + q) * 2;
    o[0] = s1; o[1] = s2;
  }
}

// stage 2 when the producing conv left per-work-group column sums (ConvArgs::stats_part): thread per (n, group)
// sums the image's `sbpi` sample blocks and the group's channels, in a fixed order -> (mean, rstd)
__global__ void gn_final_fused_kernel(const float* __restrict__ part, float* __restrict__ stats, int N, int sbpi,
                                      int C, int G, int HW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * G) return;
  const int n = i / G, g = i % G, cpg = C / G;
  float s1 = 0.f, s2 = 0.f;
  for (int sb = 0; sb < sbpi; ++sb) {
    const float* p = part + (((size_t)n * sbpi + sb) * C + g * cpg) * 2;
    for (int c = 0; c < cpg; ++c) { s1 += p[2 * c]; s2 += p[2 * c + 1]; }
  }
  const float inv = 1.0f / ((float)HW * (float)cpg);
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.0f);
  stats[i * 2] = mean;
  stats[i * 2 + 1] = 1.0f / sqrtf(var + 1e-6f);
}

// stage 2: thread per (n, group): sum chunks and the quads of the group -> (mean, rstd)
__global__ void gn_final_kernel(const float* __restrict__ part, float* __restrict__ stats, int N, int nchunk,
                                int C, int G, int HW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * G) return;
  const int n = i / G, g = i % G, cq = C / 4, qpg = (C / G) / 4;
  float s1 = 0.f, s2 = 0.f;
  for (int ch = 0; ch < nchunk; ++ch)
    for (int q = 0; q < qpg; ++q) {
      const float* p = part + (((size_t)n * nchunk + ch) * cq + g * qpg + q) * 2;
      s1 += p[0]; s2 += p[1];
    }
  const float inv = 1.0f / ((float)HW * (float)(C / G));
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.0f);
  stats[i * 2] = mean;
  stats[i * 2 + 1] = 1.0f / sqrtf(var + 1e-6f);
}

// y = act((x - mean) * rstd * scale + bias), act = swish or identity; float4 per thread
__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                const float* __restrict__ scale, const float* __restrict__ bias,
                                float* __restrict__ y, int64_t total4, int HW, int C, int G, int act) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int cq = C / 4;
  const int c = (int)(i % cq) * 4;
  const int64_t n = (i / cq) / HW;
  const float* st = stats + (n * G + c / (C / G)) * 2;
  const float mean = st[0], rstd = st[1];
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 s = *reinterpret_cast<const float4*>(scale + c);
  const float4 b = *reinterpret_cast<const float4*>(bias + c);
  float4 o;
  o.x = (v.x - mean) * rstd * s.x + b.x; o.y = (v.y - mean) * rstd * s.y + b.y;
  o.z = (v.z - mean) * rstd * s.z + b.z; o.w = (v.w - mean) * rstd * s.w + b.w;
  if (act) { o.x = swish_f(o.x); o.y = swish_f(o.y); o.z = swish_f(o.z); o.w = swish_f(o.w); }
  reinterpret_cast<float4*>(y)[i] = o;
}

// single-head attention over T tokens (T = 4): block = image, thread = channel
__global__ void attn_small_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                  const float* __restrict__ v, float* __restrict__ o, int T, int C, int ld) {
  extern __shared__ float sh[];          // scores [T][T]
  const int n = blockIdx.x, c = threadIdx.x;
  const float sc = 1.0f / sqrtf(sqrtf((float)C));
  const float* qn = q + (size_t)n * T * ld;          // row stride ld (C, or 3C when q|k|v share one tensor)
  const float* kn = k + (size_t)n * T * ld;
  const float* vn = v + (size_t)n * T * ld;
  // scores[i][j] = sum_c (q[i][c] sc) (k[j][c] sc): block-wide reduction per (i, j)
  for (int ij = 0; ij < T * T; ++ij) {
    const int i = ij / T, j = ij % T;
    float p = (qn[i * ld + c] * sc) * (kn[j * ld + c] * sc);
    p = wave_sum(p);
    __shared__ float red[16];
    if ((c & 63) == 0) red[c >> 6] = p;
    __syncthreads();
    if (c == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
      sh[ij] = t;
    }
    __syncthreads();
  }
  for (int i = 0; i < T; ++i) {
    float m = sh[i * T];
    for (int j = 1; j < T; ++j) m = fmaxf(m, sh[i * T + j]);
    float den = 0.f, acc = 0.f;
    for (int j = 0; j < T; ++j) {
      const float e = expf(sh[i * T + j] - m);
      den += e;
      acc += e * vn[j * ld + c];
    }
    o[((size_t)n * T + i) * C + c] = acc / den;
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                           int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 u = reinterpret_cast<const float4*>(a)[i], w = reinterpret_cast<const float4*>(b)[i];
  reinterpret_cast<float4*>(y)[i] = float4{u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w};
}

// out[r][j] = sum_i in[r][i] (stride ldi) W[i][j] + b[j], j < nout (tiny 1x1 convs: quant / post_quant)
__global__ void tiny_dense_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ w,
                                  const float* __restrict__ b, float* __restrict__ out, int ldo, int64_t rows,
                                  int nin, int nw, int nout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nout) return;
  const int64_t r = i / nout;
  const int j = (int)(i % nout);
  float acc = 0.f;
  for (int k = 0; k < nin; ++k) acc = fmaf(in[r * ldi + k], w[k * nw + j], acc);
  out[r * ldo + j] = acc + b[j];
}

// nearest-neighbour x2 (jax.image.resize 'nearest' on an integer factor = pixel replication)
__global__ void upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = C / 4;
  const int64_t total = (int64_t)N * 2 * H * 2 * W * cq;
  if (i >= total) return;
  const int q = (int)(i % cq);
  const int64_t p = i / cq;
  const int wo = (int)(p % (2 * W)), ho = (int)((p / (2 * W)) % (2 * H));
  const int64_t n = p / ((int64_t)4 * H * W);
  reinterpret_cast<float4*>(y)[i] =
      reinterpret_cast<const float4*>(x)[((n * H + ho / 2) * W + wo / 2) * cq + q];
}

// (N, H, W, CP) first 3 channels -> (N, 3, H, W)  (decode(...).sample is NCHW)
__global__ void nhwc_to_nchw3_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int CP) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * 3 * HW) return;
  const int p = (int)(i % HW), c = (int)((i / HW) % 3);
  const int64_t n = i / ((int64_t)3 * HW);
  y[i] = x[(n * HW + p) * CP + c];
}

inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------
int up_vec(ldp_handle* h, const std::string& path, int n, DevBuf& out) {
  const HostTensor* t = nullptr;
  LDP_TRY(get_weight(h, path, &t, {n}));
  return upload(out, t->data.data(), (size_t)n * 4, nullptr);
}

int load_gn(ldp_handle* h, const std::string& p, int c, GnW& g) {
  g.c = c;
  LDP_TRY(up_vec(h, p + "/scale", c, g.scale));
  return up_vec(h, p + "/bias", c, g.bias);
}

// 3x3 kernel (3,3,Cin,Cout) -> Toeplitz packing with the image rows folded into K:
// W'[dw][dh * Cin_p + c][co] = W[dh][dw][c][co]
int load_conv3(ldp_handle* h, const std::string& p, int cin, int cout, int cin_p, int cout_p, ConvW& out, bool planes16 = false) {
  const HostTensor *k = nullptr, *b = nullptr;
  LDP_TRY(get_weight(h, p + "/kernel", &k, {3, 3, cin, cout}));
  LDP_TRY(get_weight(h, p + "/bias", &b, {cout}));
  std::vector<float> tmp((size_t)3 * 3 * cin_p * cout, 0.f);
  for (int dh = 0; dh < 3; ++dh)
    for (int dw = 0; dw < 3; ++dw)
      for (int c = 0; c < cin; ++c)
        std::copy(k->data.begin() + (((size_t)dh * 3 + dw) * cin + c) * cout,
                  k->data.begin() + (((size_t)dh * 3 + dw) * cin + c + 1) * cout,
                  tmp.begin() + (((size_t)dw * 3 + dh) * cin_p + c) * cout);
  std::vector<float> packed = pack_conv(tmp.data(), 3, 3 * cin_p, cout, 3 * cin_p, cout_p);
  LDP_TRY(upload(out.w, packed.data(), packed.size() * 4, nullptr));
  std::vector<float> bb(cout_p, 0.f);
  std::copy(b->data.begin(), b->data.end(), bb.begin());
  LDP_TRY(upload(out.bias, bb.data(), bb.size() * 4, nullptr));
  out.nj = 3; out.cin = cin; out.cout = cout; out.cin_p = cin_p; out.cout_p = cout_p;
  out.wsplit16h.release();
  if (planes16 && cin == cin_p && cout == cout_p && cin % 64 == 0 && cout % 64 == 0) {
    // Downsample2D (stride 2) and the 8-pixel level's stride-1 convs (round 5): the same virtual-channel order (dh * C + c, taps along W) as two fp16 planes for tconv's 16-row split tile
    // (MODE_K3S / MODE_K3H, SPLIT = 3) -- unless a weight is outside their range
    out.f16_refused = !fits_f16_planes(k->data.data(), k->data.size());
    if (!out.f16_refused) {
      const std::vector<uint16_t> wp = pack_conv_split16h(tmp.data(), 3, 3 * cin_p, cout);
      LDP_TRY(out.wsplit16h.alloc(wp.size() * 2));
      LDP_HIP(hipMemcpy(out.wsplit16h.p, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
    }
  }
  if (cin == cin_p && cout == cout_p && cin % 16 == 0 && cout % 128 == 0) {
    // the same kernel as three bf16 planes in the split-operand conv's LDS-image order (sconv.hpp)
    std::vector<uint16_t> wp = pack_sconv3(k->data.data(), cin, cout);
    LDP_TRY(upload(out.wsplit, wp.data(), wp.size() * 2, nullptr));
    // ... and as two fp16 planes (option vae_split_f16) -- unless a weight is outside their range (|w| >= 65504): that conv keeps the bf16 planes
    out.wsplith.release();
    out.f16_refused = !fits_f16_planes(k->data.data(), k->data.size());
    if (!out.f16_refused) {
      wp = pack_sconv3(k->data.data(), cin, cout, 2);
      LDP_TRY(upload(out.wsplith, wp.data(), wp.size() * 2, nullptr));
    }
  }
  return LDP_OK;
}

int load_conv1(ldp_handle* h, const std::string& p, int cin, int cout, ConvW& out) {
  const HostTensor *k = nullptr, *b = nullptr;
  LDP_TRY(get_weight(h, p + "/kernel", &k, {1, 1, cin, cout}));
  LDP_TRY(get_weight(h, p + "/bias", &b, {cout}));
  std::vector<float> packed = pack_conv(k->data.data(), 1, cin, cout, cin, cout);
  LDP_TRY(upload(out.w, packed.data(), packed.size() * 4, nullptr));
  LDP_TRY(upload(out.bias, b->data.data(), (size_t)cout * 4, nullptr));
  out.nj = 1; out.cin = cin; out.cout = cout; out.cin_p = cin; out.cout_p = cout;
  return LDP_OK;
}

int load_res(ldp_handle* h, const std::string& p, int cin, int cout, Res2dW& r, bool px8 = false) {
  r.cin = cin; r.cout = cout;
  LDP_TRY(load_gn(h, p + "/norm1", cin, r.n1));
  LDP_TRY(load_conv3(h, p + "/conv1", cin, cout, cin, cout, r.c1, px8));
  LDP_TRY(load_gn(h, p + "/norm2", cout, r.n2));
  LDP_TRY(load_conv3(h, p + "/conv2", cout, cout, cout, cout, r.c2, px8));
  r.has_sc = cin != cout;
  if (r.has_sc) LDP_TRY(load_conv1(h, p + "/conv_shortcut", cin, cout, r.sc));
  return LDP_OK;
}

int load_dense(ldp_handle* h, const std::string& p, int cin, int cout, DevBuf& w, DevBuf& b) {
  const HostTensor *k = nullptr, *bb = nullptr;
  LDP_TRY(get_weight(h, p + "/kernel", &k, {cin, cout}));
  LDP_TRY(get_weight(h, p + "/bias", &bb, {cout}));
  LDP_TRY(upload(w, k->data.data(), k->data.size() * 4, nullptr));
  return upload(b, bb->data.data(), bb->data.size() * 4, nullptr);
}

int load_mid(ldp_handle* h, const std::string& p, int c, MidW& m) {
  LDP_TRY(load_res(h, p + "/resnets_0", c, c, m.r0));
  LDP_TRY(load_res(h, p + "/resnets_1", c, c, m.r1));
  m.at.c = c;
  const std::string a = p + "/attentions_0";
  LDP_TRY(load_gn(h, a + "/group_norm", c, m.at.gn));
  LDP_TRY(load_dense(h, a + "/query", c, c, m.at.wq, m.at.bq));
  LDP_TRY(load_dense(h, a + "/key", c, c, m.at.wk, m.at.bk));
  LDP_TRY(load_dense(h, a + "/value", c, c, m.at.wv, m.at.bv));
  LDP_TRY(load_dense(h, a + "/proj_attn", c, c, m.at.wo, m.at.bo));
  {
    // MFMA path: q, k, v as ONE 1x1 conv with 3C output columns, proj_attn as another
    const HostTensor *kq, *kk, *kv, *bq, *bk, *bv, *ko, *bo;
    LDP_TRY(get_weight(h, a + "/query/kernel", &kq, {c, c}));
    LDP_TRY(get_weight(h, a + "/key/kernel", &kk, {c, c}));
    LDP_TRY(get_weight(h, a + "/value/kernel", &kv, {c, c}));
    LDP_TRY(get_weight(h, a + "/query/bias", &bq, {c}));
    LDP_TRY(get_weight(h, a + "/key/bias", &bk, {c}));
    LDP_TRY(get_weight(h, a + "/value/bias", &bv, {c}));
    LDP_TRY(get_weight(h, a + "/proj_attn/kernel", &ko, {c, c}));
    LDP_TRY(get_weight(h, a + "/proj_attn/bias", &bo, {c}));
    std::vector<float> w3((size_t)c * 3 * c), b3((size_t)3 * c);
    for (int r = 0; r < c; ++r) {
      std::copy(kq->data.begin() + (size_t)r * c, kq->data.begin() + (size_t)(r + 1) * c, w3.begin() + (size_t)r * 3 * c);
      std::copy(kk->data.begin() + (size_t)r * c, kk->data.begin() + (size_t)(r + 1) * c, w3.begin() + (size_t)r * 3 * c + c);
      std::copy(kv->data.begin() + (size_t)r * c, kv->data.begin() + (size_t)(r + 1) * c, w3.begin() + (size_t)r * 3 * c + 2 * c);
    }
    std::copy(bq->data.begin(), bq->data.end(), b3.begin());
    std::copy(bk->data.begin(), bk->data.end(), b3.begin() + c);
    std::copy(bv->data.begin(), bv->data.end(), b3.begin() + 2 * c);
    auto mk = [&](const std::vector<float>& w, const float* b, int cout, ConvW& out) -> int {
      std::vector<float> packed = pack_conv(w.data(), 1, c, cout, c, cout);
      LDP_TRY(upload(out.w, packed.data(), packed.size() * 4, nullptr));
      LDP_TRY(upload(out.bias, b, (size_t)cout * 4, nullptr));
      out.nj = 1; out.cin = c; out.cout = cout; out.cin_p = c; out.cout_p = cout;
      return LDP_OK;
    };
    LDP_TRY(mk(w3, b3.data(), 3 * c, m.at.qkv));
    LDP_TRY(mk(ko->data, bo->data.data(), c, m.at.proj));
  }
  return LDP_OK;
}

// ---------------------------------------------------------------------------------------------
// layer launches
// ---------------------------------------------------------------------------------------------
struct Run {
  ldp_handle* h;
  VaeState& S;
  hipStream_t s;
  // the tensor whose column sums the last 3x3 conv left in S.part2 (nullptr: none), and their geometry
  const float* fused_for = nullptr;
  int fused_sbpi = 0, fused_c = 0;
  // the tensor whose gn_part-style sums its producer (conv_in) left in S.part
  const float* part_for = nullptr;
  int part_nchunk = 0, part_c = 0;
  void wrote(const float* p) { if (fused_for == p) fused_for = nullptr; if (part_for == p) part_for = nullptr; }      // any other writer of that buffer

  // GroupNorm statistics of x -> S.stats (N, G, {mean, rstd})
  int gn_stats(const GnW& g, const float* x, int N, int HW) {
    const int C = g.c, G = S.G;
    const int nchunk = (HW + PCH - 1) / PCH;
    if (x == fused_for && C == fused_c && !h->opt.vae_no_conv_stats) {
      // the conv that wrote x summed its columns on the way out: no second pass over x for the statistics
      hipLaunchKernelGGL(gn_final_fused_kernel, dim3(nblk((int64_t)N * G)), dim3(256), 0, s, S.part2.f(), S.stats.f(), N,
                         fused_sbpi, C, G, HW);
    } else if (x == part_for && C == part_c) {
      hipLaunchKernelGGL(gn_final_kernel, dim3(nblk((int64_t)N * G)), dim3(256), 0, s, S.part.f(), S.stats.f(), N,
                         part_nchunk, C, G, HW);
      part_for = nullptr;                                  // S.part is scratch for every other GroupNorm
    } else {
      part_for = nullptr;
      int threads = 256;
      while (threads % (C / 4) != 0) threads += 64;          // whole pixel rows per pass
      if (threads < C / 4) threads = C / 4;
      hipLaunchKernelGGL(gn_part_kernel, dim3(nchunk, N), dim3(threads), threads * 8, s, x, S.part.f(), HW, C);
      hipLaunchKernelGGL(gn_final_kernel, dim3(nblk((int64_t)N * G)), dim3(256), 0, s, S.part.f(), S.stats.f(), N,
                         nchunk, C, G, HW);
    }
    LDP_HIP(hipGetLastError());
    return LDP_OK;
  }

  int gn(const GnW& g, const float* x, float* y, int N, int HW, int act) {
    const int C = g.c, G = S.G;
    LDP_TRY(gn_stats(g, x, N, HW));
    const int64_t t4 = (int64_t)N * HW * (C / 4);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk(t4)), dim3(256), 0, s, x, S.stats.f(), g.scale.f(), g.bias.f(),
                       y, t4, HW, C, G, act);
    wrote(y);
    LDP_HIP(hipGetLastError());
    return LDP_OK;
  }

  // 3x3 conv, NHWC (N, Hin, Win, cin_p) -> (N, Hout, Wout, cout_p); stride 1 (pad 1) or 2 (pad (0,1))
  // the split-operand conv on an input that is already (or needs no) normalised: stats == nullptr -> plain split
  int split_conv3(const ConvW& w, const GnW* g, const float* x, float* y, int N, int H, int W, const float* res) {
    // two fp16 planes unless the handle fell back to the bf16 form (a range fault, engine.hpp) or this conv's weights do not fit them
    const int npl = (h->f16_vae() && w.wsplith.p) ? 2 : 3;
    int r = planes_launch(x, g ? S.stats.f() : nullptr, g ? g->scale.f() : nullptr, g ? g->bias.f() : nullptr, S.planes.p,
                          N, H * W, w.cin_p, S.G, 1, s, npl, h->range_dev());
    if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "planes launch failed (%d)", r);
    const int tpi = H * W / 256;
    const bool fuse = (size_t)N * tpi * w.cout_p * 8 <= S.part2.bytes;
    SConvArgs a{S.planes.p, npl == 2 ? w.wsplith.p : w.wsplit.p, w.bias.f(), res, y, fuse ? S.part2.f() : nullptr, S.zero.p,
                N, H, W, w.cin_p, w.cout_p, h->opt.dbg};
    a.npl = npl;
    r = sconv3_launch(a, s);
    if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "split-operand 3x3 conv launch failed (%d)", r);
    wrote(y);
    if (fuse) { fused_for = y; fused_sbpi = tpi; fused_c = w.cout_p; }
    return LDP_OK;
  }
  bool split_ok(const ConvW& w, int N, int H, int W) const {
    return h->opt.vae_split && w.wsplit.p && sconv3_supported(H, W, w.cin_p, w.cout_p) &&
           PlaneGeom{N, H, W, w.cin_p}.bytes() <= S.planes.bytes;
  }

  int conv3(const ConvW& w, const float* x, float* y, int N, int Hin, int Win, int stride, const float* res) {
    // stride-1 convs on raw inputs (the decoder's upsamplers): one extra pass splits the input into planes (4 B read,
    // 6 B written per element) and the conv runs on the bf16 matrix pipe (sconv.hpp)
    if (stride == 1 && split_ok(w, N, Hin, Win) && !h->opt.vae_split_gn_only)
      return split_conv3(w, nullptr, x, y, N, Hin, Win, res);
    const int Ho = Hin / stride, Wo = Win / stride;
    // row tiles of 8 / 4 / 2 output pixels, whichever is the widest that divides the row (64- and 128-pixel frames: 8 down to 2; 96-pixel
    // frames: 96 / 48 / 24 -> 8, 12 -> 4, 6 -> 2, and 3 -> the 3-pixel tile)
    const int to = Wo % 8 == 0 ? 8 : Wo % 4 == 0 ? 4 : Wo % 2 == 0 ? 2 : Wo == 3 ? 3 : 0;
    if (to == 0)
      return fail(LDP_EINVAL, "unsupported image width %d for the 3x3 conv tiles", Wo);
    ConvPlan p{stride == 1 ? MODE_K3H : MODE_K3S, to, 2, 4, 1, 0};
    if (to == 3) {                          // 64-column tiles only (3 x 64 = three 64-lane epilogue rows)
      if (w.cout_p % 64 != 0 || w.cin_p % 64 != 0)
        return fail(LDP_EINVAL, "3-pixel conv tile: %d -> %d channels must both be multiples of 64", w.cin_p, w.cout_p);
      p.nwn = 4; p.ks = 1; p.cpi = stride == 1 ? 4 : 2;
    }
    // 64-column tiles (4 column waves, 32-channel sub-chunks) where the shape allows: the activation tile is staged once per 64
    // instead of per 32 output channels (+21 % on the encoder; stride 2 as well: 18-pixel input tile),
    // as FOUR-wave work-groups without a K split over waves (40 KB of LDS, 154 VGPRs: three of them share a CU, one's barrier / LDS phase
    // runs under the others' MFMAs, no K combine in the epilogue; round 3: encode 27.05 -> 26.03 ms at 256 frames against the eight-wave
    // 4 x 2 tiles, which are no longer built)
    if (to == 8 && w.cout_p % 64 == 0 && w.cin_p % 64 == 0 && (stride == 1 || !h->opt.no_mb2)) { p.nwn = 4; p.ks = 1; p.cpi = 2; }
    if (w.cin_p % p.chunk() != 0 || w.cout_p % p.bn() != 0)
      return fail(LDP_EINVAL, "3x3 conv %d->%d does not tile (chunk %d, block %d)", w.cin_p, w.cout_p, p.chunk(), p.bn());
    ConvArgs a{};
    a.xa = x; a.ca = w.cin_p; a.w = w.w.f(); a.bias = w.bias.f();
    a.out = y; a.cout = w.cout_p; a.res_in = res; a.flags = res ? EP_RESIN : 0;
    a.h_out = Ho; a.w_tiles = Wo / to; a.h_in = Hin; a.w_in = Win;
    a.B = N * Ho * a.w_tiles; a.rows_valid = a.B * to;
    // 64-column tiles whose 16 row tiles lie in one image: leave the column sums for the GroupNorm that follows
    const int tpi = Ho * a.w_tiles;
    const bool fuse = stride == 1 && p.nwn == 4 && p.ks == 1 && tpi % 16 == 0 && (size_t)(a.B / 16) * w.cout_p * 8 <= S.part2.bytes;
    if (fuse) a.stats_part = S.part2.f();
    a.dbg = h->opt.dbg;                                      // timing ablations for tools/ (0 in production)
    // Downsample2D on two fp16 planes / three products (round 5): the 64-column four-wave tile's 16-row split form; the operand is the RAW residual
    // stream, so the range guard (fault word [1]) decides -- a fault reruns the call with these convs on the exact-fp32 tile
    // (also the stride-1 convs of the 8-pixel level, whose 64-pixel images are too small for sconv3's 256-pixel tiles)
    if ((stride == 2 || (Win == 8 && !fuse)) && to == 8 && p.nwn == 4 && p.ks == 1 && p.cpi == 2 && h->opt.vae_split && h->opt.vae_split_s2 && h->f16_vae() && w.wsplit16h.p) {
      p.split = 3; p.mb = 1;
      a.w = w.wsplit16h.f();
      a.fault = h->fault_dev;
      h->stat_f16_launches++;
    }
    const int r = tconv_launch(p, a, s);
    if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "3x3 conv launch failed (%d)", r);
    if (fused_for == y) fused_for = nullptr;               // y rewritten: older sums are stale
    if (fuse) { fused_for = y; fused_sbpi = tpi / 16; fused_c = w.cout_p; }
    return LDP_OK;
  }

  // norm -> swish -> 3x3 conv (ResnetBlock2D halves, conv_norm_out -> conv_out).  GroupNorm+swish stay their own
  // HBM-bound kernels: applied inside the conv's staging path instead (every input element is staged by 8-15
  // work-groups: 3 image rows x halo x output-column blocks) the transform measured 17 % SLOWER end to end.
  int gn_conv3(const GnW& g, const ConvW& w, const float* x, float* y, float* tmp, int N, int H, int W,
               const float* res) {
    if (g.c == w.cin_p && split_ok(w, N, H, W)) {
      // split-operand path (sconv.hpp): GroupNorm + swish leave the conv's input as three bf16 planes, the conv runs
      // on v_mfma_f32_32x32x16_bf16 (six plane products, fp32 accumulate) and sums its columns for the next GroupNorm
      LDP_TRY(gn_stats(g, x, N, H * W));
      return split_conv3(w, &g, x, y, N, H, W, res);
    }
    LDP_TRY(gn(g, x, tmp, N, H * W, 1));
    return conv3(w, tmp, y, N, H, W, 1, res);
  }

  // 1x1 conv over pixels (rows grouped by 8)
  int conv1(const ConvW& w, const float* x, float* y, int64_t pixels, const float* res = nullptr) {
    if (pixels % 8 != 0) return fail(LDP_EINVAL, "1x1 conv needs a multiple of 8 pixels");
    ConvPlan p{MODE_P1, 8, 2, 4, 1, 0};
    ConvArgs a{};
    a.xa = x; a.ca = w.cin_p; a.w = w.w.f(); a.bias = w.bias.f(); a.out = y; a.cout = w.cout_p;
    a.res_in = res; a.flags = res ? EP_RESIN : 0;
    wrote(y);
    a.B = (int)(pixels / 8); a.rows_valid = (int)pixels;
    const int r = tconv_launch(p, a, s);
    if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "1x1 conv launch failed (%d)", r);
    return LDP_OK;
  }

  // ResnetBlock2D: x -> out (tmp buffers t0, t1; sc buffer for the projected shortcut)
  int res(const Res2dW& r, const float* x, float* out, float* t0, float* t1, float* scb, int N, int H, int W) {
    LDP_TRY(gn_conv3(r.n1, r.c1, x, t1, t0, N, H, W, nullptr));
    const float* skip = x;
    if (r.has_sc) {
      LDP_TRY(conv1(r.sc, x, scb, (int64_t)N * H * W));
      skip = scb;
    }
    return gn_conv3(r.n2, r.c2, t1, out, t0, N, H, W, skip);
  }

  int attn(const AttnW& at, const float* x, float* out, float* t0, int N, int T) {
    const int C = at.c;
    const int R = N * T;
    float *q = S.small[0].f(), *k = S.small[1].f(), *v = S.small[2].f(), *o = S.small[3].f(), *pr = S.small[4].f();
    LDP_TRY(gn(at.gn, x, t0, N, T, 0));
    if (R % 8 == 0) {
      // the four Dense layers on the MFMA 1x1 kernel: q|k|v in one launch (small[0..2] are contiguous slices
      // of one allocation: see workspace), proj_attn with the residual add fused into its epilogue
      float* qkv = S.qkv.f();
      LDP_TRY(conv1(at.qkv, t0, qkv, R));
      hipLaunchKernelGGL(attn_small_kernel, dim3(N), dim3(C), T * T * 4, s, qkv, qkv + C, qkv + 2 * C, o, T, C, 3 * C);
      LDP_HIP(hipGetLastError());
      return conv1(at.proj, o, out, R, x);
    }
    LDP_TRY(dense_launch(t0, C, at.wq.f(), C, at.bq.f(), q, C, R, C, C, 0, 0, s));
    LDP_TRY(dense_launch(t0, C, at.wk.f(), C, at.bk.f(), k, C, R, C, C, 0, 0, s));
    LDP_TRY(dense_launch(t0, C, at.wv.f(), C, at.bv.f(), v, C, R, C, C, 0, 0, s));
    hipLaunchKernelGGL(attn_small_kernel, dim3(N), dim3(C), T * T * 4, s, q, k, v, o, T, C, C);
    LDP_TRY(dense_launch(o, C, at.wo.f(), C, at.bo.f(), pr, C, R, C, C, 0, 0, s));
    const int64_t n4 = (int64_t)R * C / 4;
    hipLaunchKernelGGL(add_kernel, dim3(nblk(n4)), dim3(256), 0, s, pr, x, out, n4);
    wrote(out);
    LDP_HIP(hipGetLastError());
    return LDP_OK;
  }

  int mid(const MidW& m, float*& cur, float*& o1, float* t0, float* t1, int N, int H, int W) {
    LDP_TRY(res(m.r0, cur, o1, t0, t1, nullptr, N, H, W));
    std::swap(cur, o1);
    LDP_TRY(attn(m.at, cur, o1, t0, N, H * W));
    std::swap(cur, o1);
    LDP_TRY(res(m.r1, cur, o1, t0, t1, nullptr, N, H, W));
    std::swap(cur, o1);
    return LDP_OK;
  }
};

int workspace(ldp_handle* h, int n) {
  VaeState& S = *V(h);
  if (n <= S.ws_n) return LDP_OK;
  // largest activation: the decoder's last upsampler works on S x S x ch[1] (64 x 64 x 256)
  const size_t big = (size_t)n * S.S * S.S * std::max(S.ch[0], S.ch[1]) * 4;
  LDP_TRY(S.b0.alloc(big)); LDP_TRY(S.b1.alloc(big)); LDP_TRY(S.b2.alloc(big)); LDP_TRY(S.b3.alloc(big));
  LDP_TRY(S.b4.alloc(big));                                       // projected shortcuts
  const int nchunk = (S.S * S.S + PCH - 1) / PCH;
  // gn_part's layout [n][chunk of 256 pixels][C/4][2] and conv_in's [n][image row][C/4][2] (C <= 256)
  LDP_TRY(S.part.alloc((size_t)n * std::max(nchunk, S.S) * (256 / 4) * 2 * 4 * 2));
  LDP_TRY(S.stats.alloc((size_t)n * S.G * 2 * 4));
  LDP_TRY(S.part2.alloc((size_t)n * (S.S * S.S / 8 / 16) * 256 * 2 * 4));      // [sample block][C <= 256][2]
  LDP_TRY(S.planes.alloc((size_t)n * S.S * S.S * std::max(S.ch[0], S.ch[1]) * 6));     // three bf16 planes of the largest conv input
  if (!S.zero.p) { LDP_TRY(S.zero.alloc(256)); LDP_HIP(hipMemset(S.zero.p, 0, 256)); }
  for (auto& b : S.small) LDP_TRY(b.alloc((size_t)n * 16 * 256 * 4));
  LDP_TRY(S.qkv.alloc((size_t)n * 16 * 3 * 256 * 4));
  S.ws_n = n;
  return LDP_OK;
}

}  // namespace

int vae_finalize(ldp_handle* h, hipStream_t s) {
  if (!h->vae) h->vae = new VaeState();
  VaeState& S = *V(h);
  S.enc_ready = S.dec_ready = false;
  S.S = h->cfg.image_size > 0 ? h->cfg.image_size : 64;
  S.LC = h->cfg.vae_latent_channels > 0 ? h->cfg.vae_latent_channels : 4;
  S.ch = {128, 256, 256, 256, 256, 256};                   // model/stable_vae_model.yaml:6
  const int NB = (int)S.ch.size(), C0 = S.ch[0], CL = S.ch.back();
  // every level's width must tile by 8 / 4 / 3 / 2 pixels: 64 (2x2 latent), 96 (3x3 latent, vae_feature_dim 36) or 128 (4x4 latent, vae_feature_dim 64)
  // 96 (3x3 latent, vae_feature_dim 36) would need 3-pixel tiles
  if (S.S != 64 && S.S != 96 && S.S != 128)
    return fail(LDP_EINVAL, "image_size %d: the 3x3 conv tiles are built for 64, 96 or 128 pixel squares", S.S);
  if (S.LC < 1 || 2 * S.LC > 32)
    return fail(LDP_EINVAL, "vae_latent_channels %d: at most 16", S.LC);
  const std::string e = "vae/encoder/";
  {
    const HostTensor *k = nullptr, *b = nullptr;
    LDP_TRY(get_weight(h, e + "conv_in/kernel", &k, {3, 3, 3, C0}));
    LDP_TRY(get_weight(h, e + "conv_in/bias", &b, {C0}));
    LDP_TRY(upload(S.cin_w, k->data.data(), k->data.size() * 4, s));
    LDP_TRY(upload(S.cin_b, b->data.data(), b->data.size() * 4, s));
  }
  S.down.clear();
  S.down.resize(NB);
  int cin = C0;
  for (int i = 0; i < NB; ++i) {
    const std::string p = e + "down_blocks_" + std::to_string(i);
    for (int j = 0; j < 2; ++j) {
      LDP_TRY(load_res(h, p + "/resnets_" + std::to_string(j), cin, S.ch[i], S.down[i].r[j], (S.S >> i) == 8));
      cin = S.ch[i];
    }
    S.down[i].has_ds = i != NB - 1;
    if (S.down[i].has_ds) LDP_TRY(load_conv3(h, p + "/downsamplers_0/conv", cin, cin, cin, cin, S.down[i].ds, true));
  }
  LDP_TRY(load_mid(h, e + "mid_block", CL, S.emid));
  LDP_TRY(load_gn(h, e + "conv_norm_out", CL, S.enorm));
  S.eout_cols = (S.S >> (NB - 1)) == 3 ? 64 : 32;          // the 3-pixel tile is 64 columns wide
  LDP_TRY(load_conv3(h, e + "conv_out", CL, 2 * S.LC, CL, S.eout_cols, S.econv_out));
  {
    const HostTensor *k = nullptr, *b = nullptr;
    LDP_TRY(get_weight(h, "vae/quant_conv/kernel", &k, {1, 1, 2 * S.LC, 2 * S.LC}));
    LDP_TRY(get_weight(h, "vae/quant_conv/bias", &b, {2 * S.LC}));
    LDP_TRY(upload(S.quant_w, k->data.data(), k->data.size() * 4, s));
    LDP_TRY(upload(S.quant_b, b->data.data(), b->data.size() * 4, s));
  }
  S.enc_ready = true;

  // decoder (optional: only when its weights were provided)
  if (h->weights.count("vae/decoder/conv_in/kernel")) {
    const std::string d = "vae/decoder/";
    {
      const HostTensor *k = nullptr, *b = nullptr;
      LDP_TRY(get_weight(h, "vae/post_quant_conv/kernel", &k, {1, 1, S.LC, S.LC}));
      LDP_TRY(get_weight(h, "vae/post_quant_conv/bias", &b, {S.LC}));
      LDP_TRY(upload(S.pq_w, k->data.data(), k->data.size() * 4, s));
      LDP_TRY(upload(S.pq_b, b->data.data(), b->data.size() * 4, s));
    }
    LDP_TRY(load_conv3(h, d + "conv_in", S.LC, CL, 64, CL, S.dconv_in));
    LDP_TRY(load_mid(h, d + "mid_block", CL, S.dmid));
    S.up.clear();
    S.up.resize(NB);
    int c = CL;
    for (int i = 0; i < NB; ++i) {
      const int co = S.ch[NB - 1 - i];
      const std::string p = d + "up_blocks_" + std::to_string(i);
      for (int j = 0; j < 3; ++j) {
        LDP_TRY(load_res(h, p + "/resnets_" + std::to_string(j), c, co, S.up[i].r[j], ((S.S >> (NB - 1)) << i) == 8));
        c = co;
      }
      S.up[i].has_us = i != NB - 1;
      if (S.up[i].has_us) LDP_TRY(load_conv3(h, p + "/upsamplers_0/conv", c, c, c, c, S.up[i].us, ((S.S >> (NB - 1)) << (i + 1)) == 8));
    }
    LDP_TRY(load_gn(h, d + "conv_norm_out", C0, S.dnorm));
    LDP_TRY(load_conv3(h, d + "conv_out", C0, 3, C0, 32, S.dconv_out));
    S.dec_ready = true;
  }
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}

void vae_destroy(ldp_handle* h) {
  delete V(h);
  h->vae = nullptr;
}

}  // namespace ldp

using namespace ldp;

extern "C" {

int ldp_vae_encode(ldp_handle* h, const float* img, float* mean_out, int32_t N, void* stream) {
  if (!h || !img || !mean_out || N <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->vae || !V(h)->enc_ready) return fail(LDP_ESTATE, "vae weights not finalized");
  LDP_TRY(entry_fault_check(h));
  VaeState& S = *V(h);
  hipStream_t s = (hipStream_t)stream;
  const int CHUNK = 256;                                   // images per pass (bounds the workspace)
  const int NB = (int)S.ch.size();
  const int hl = S.S >> (NB - 1);                          // latent side (2 for 64x64)
  for (int n0 = 0; n0 < N; n0 += CHUNK) {
    const int n = std::min(CHUNK, N - n0);
    LDP_TRY(workspace(h, n));
    Run R{h, S, s};
    float *cur = S.b0.f(), *o1 = S.b1.f(), *t0 = S.b2.f(), *t1 = S.b3.f();
    int H = S.S, C = S.ch[0];
    {
      constexpr int PPB = 8;                            // (C/4) * 8 = 256 threads at C = 128
      const size_t lds = std::max((size_t)3 * (H + 2) * 3 * 4, (size_t)(C / 4) * PPB * 2 * 4);
      // its GroupNorm sums go to S.part as part[n][row][C/4][2]: 2 * S * C bytes per image, and (C/4) * PPB threads per block
      const bool stats_fit = !h->opt.vae_no_conv_in_stats && (size_t)n * H * (C / 4) * 8 <= S.part.bytes;
      if ((C / 4) * PPB > 1024) return fail(LDP_EINVAL, "conv_in: %d output channels need %d threads per block", C, (C / 4) * PPB);
      hipLaunchKernelGGL(conv_in3_kernel<PPB>, dim3(n * H), dim3((C / 4) * PPB), lds, s,
                         img + (size_t)n0 * H * H * 3, S.cin_w.f(), S.cin_b.f(), cur, stats_fit ? S.part.f() : nullptr, n, H, C);
      R.wrote(cur);
      // stage-1 GroupNorm sums of `cur` are in S.part (one chunk per image row) when the kernel could write them there
      if (stats_fit) { R.part_for = cur; R.part_nchunk = H; R.part_c = C; }
      LDP_HIP(hipGetLastError());
    }
    for (int i = 0; i < NB; ++i) {
      for (int j = 0; j < 2; ++j) {
        LDP_TRY(R.res(S.down[i].r[j], cur, o1, t0, t1, S.b4.f(), n, H, H));
        std::swap(cur, o1);
      }
      if (S.down[i].has_ds) {
        LDP_TRY(R.conv3(S.down[i].ds, cur, o1, n, H, H, 2, nullptr));
        std::swap(cur, o1);
        H /= 2;
      }
    }
    LDP_TRY(R.mid(S.emid, cur, o1, t0, t1, n, H, H));
    LDP_TRY(R.gn_conv3(S.enorm, S.econv_out, cur, t1, t0, n, H, H, nullptr));          // (n, hl, hl, 32): first 2*LC real
    // quant_conv 1x1 (2LC -> 2LC), keep the mean = first LC channels
    const int64_t rows = (int64_t)n * hl * hl;
    hipLaunchKernelGGL(tiny_dense_kernel, dim3(nblk(rows * S.LC)), dim3(256), 0, s, t1, S.eout_cols, S.quant_w.f(),
                       S.quant_b.f(), mean_out + (size_t)n0 * hl * hl * S.LC, S.LC, rows, 2 * S.LC, 2 * S.LC, S.LC);
    LDP_HIP(hipGetLastError());
  }
  return LDP_OK;
}

int ldp_vae_decode(ldp_handle* h, const float* z, float* img_out, int32_t N, void* stream) {
  if (!h || !z || !img_out || N <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->vae || !V(h)->dec_ready) return fail(LDP_ESTATE, "vae decoder weights not finalized");
  LDP_TRY(entry_fault_check(h));
  VaeState& S = *V(h);
  hipStream_t s = (hipStream_t)stream;
  const int CHUNK = 256;
  const int NB = (int)S.ch.size();
  const int hl = S.S >> (NB - 1);
  for (int n0 = 0; n0 < N; n0 += CHUNK) {
    const int n = std::min(CHUNK, N - n0);
    LDP_TRY(workspace(h, n));
    Run R{h, S, s};
    float *cur = S.b0.f(), *o1 = S.b1.f(), *t0 = S.b2.f(), *t1 = S.b3.f();
    int H = hl;
    const int64_t rows = (int64_t)n * hl * hl;
    // post_quant 1x1 (LC -> LC) into a 64-channel zero-padded tensor (the conv kernel's channel chunk)
    LDP_HIP(hipMemsetAsync(t0, 0, (size_t)rows * 64 * 4, s));
    R.wrote(t0);
    hipLaunchKernelGGL(tiny_dense_kernel, dim3(nblk(rows * S.LC)), dim3(256), 0, s, z + (size_t)n0 * hl * hl * S.LC,
                       S.LC, S.pq_w.f(), S.pq_b.f(), t0, 64, rows, S.LC, S.LC, S.LC);
    LDP_TRY(R.conv3(S.dconv_in, t0, cur, n, H, H, 1, nullptr));
    LDP_TRY(R.mid(S.dmid, cur, o1, t0, t1, n, H, H));
    for (int i = 0; i < NB; ++i) {
      for (int j = 0; j < 3; ++j) {
        LDP_TRY(R.res(S.up[i].r[j], cur, o1, t0, t1, S.b4.f(), n, H, H));
        std::swap(cur, o1);
      }
      if (S.up[i].has_us) {
        const int C = S.up[i].us.cin;
        const int64_t tot = (int64_t)n * 4 * H * H * (C / 4);
        hipLaunchKernelGGL(upsample2_kernel, dim3(nblk(tot)), dim3(256), 0, s, cur, t0, n, H, H, C);
        R.wrote(t0);
        H *= 2;
        LDP_TRY(R.conv3(S.up[i].us, t0, o1, n, H, H, 1, nullptr));
        std::swap(cur, o1);
      }
    }
    LDP_TRY(R.gn_conv3(S.dnorm, S.dconv_out, cur, t1, t0, n, H, H, nullptr));          // (n, S, S, 32): first 3 real
    const int64_t tot = (int64_t)n * 3 * H * H;
    hipLaunchKernelGGL(nhwc_to_nchw3_kernel, dim3(nblk(tot)), dim3(256), 0, s, t1,
                       img_out + (size_t)n0 * 3 * H * H, n, H * H, 32);
    LDP_HIP(hipGetLastError());
  }
  return LDP_OK;
}

}  // extern "C"

// ---- unit-testable primitive: one 3x3 convolution of the VAE (stride 1 pad 1, or stride 2 pad (0,1)) ----
extern "C" int ldp_conv2d_3x3_f32(const float* x, const float* kernel_host, const float* bias_host, float* y,
                                  int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride,
                                  void* stream) {
  if (!x || !kernel_host || !bias_host || !y || N <= 0) return fail(LDP_EINVAL, "bad argument");
  if (Cin % 64 != 0 || Cout % 32 != 0) return fail(LDP_EINVAL, "Cin must be a multiple of 64 and Cout of 32");
  if (H != W || (stride != 1 && stride != 2)) return fail(LDP_EINVAL, "square images, stride 1 or 2");
  hipStream_t s = (hipStream_t)stream;
  std::vector<float> tmp((size_t)9 * Cin * Cout);
  for (int dh = 0; dh < 3; ++dh)
    for (int dw = 0; dw < 3; ++dw)
      std::copy(kernel_host + ((size_t)dh * 3 + dw) * Cin * Cout, kernel_host + ((size_t)dh * 3 + dw + 1) * Cin * Cout,
                tmp.begin() + ((size_t)dw * 3 + dh) * Cin * Cout);
  std::vector<float> packed = pack_conv(tmp.data(), 3, 3 * Cin, Cout, 3 * Cin, Cout);
  DevBuf dw_, db_;
  LDP_TRY(upload(dw_, packed.data(), packed.size() * 4, s));
  LDP_TRY(upload(db_, bias_host, (size_t)Cout * 4, s));
  const int Ho = H / stride, Wo = W / stride;
  const int to = Wo >= 8 ? 8 : Wo;
  ConvPlan p{stride == 1 ? MODE_K3H : MODE_K3S, to, 2, 4, 1, 0};
  if (stride == 1 && to == 8 && Cout % 64 == 0 && Cin % 64 == 0) { p.nwn = 4; p.ks = 1; p.cpi = 2; }      // the tile the engine's convs run on
  ConvArgs a{};
  a.xa = x; a.ca = Cin; a.w = dw_.f(); a.bias = db_.f(); a.out = y; a.cout = Cout;
  a.h_out = Ho; a.w_tiles = Wo / to; a.h_in = H; a.w_in = W;
  a.B = N * Ho * a.w_tiles; a.rows_valid = a.B * to;
  const int r = tconv_launch(p, a, s);
  if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "3x3 conv launch failed (%d)", r);
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}
