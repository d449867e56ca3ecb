// idm.hip -- inverse-dynamics MLP-diffusion head (networks/mlp_diffusion_nets.py:8-68) on the
// same f32-MFMA kernels: every Dense of MLPResNet runs as a 1x1 "convolution" over the rows
// viewed as (R/4, 4, C), i.e. 64-row x BN-column work-group tiles.
//
// Loop-invariant algebra hoisted out of the 100-step loop (exact up to fp32
// re-association):  Dense_0([a | s | cond_k]) = a @ W[:A]  +  (s @ W[A:A+2D] + b)  +  cond_k @ W[A+2D:]
//   * cond_k = MLP_0(FourierFeatures(k)) depends on k only -> table ctab (n_train, H) at finalize
//   * the s-part is computed once per sample() call
//   * the a-part (A = 7 or 14 inputs) is a handful of FMAs per output: VALU kernel `idm_in`.
#include "engine.hpp"

#include <algorithm>

namespace ldp {

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// h0[r][c] = sum_i a[r][i] Wa[i][c] + spart[r][c] + ctab[k_r][c]
__global__ void idm_in_kernel(const float* __restrict__ a_state, int ap, const float* __restrict__ wa,
                              int A, const float* __restrict__ spart, const float* __restrict__ ctab,
                              const int* __restrict__ k_dev, int k, float* __restrict__ h0, int R, int H) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= H || r >= R) return;
  const int kk = k_dev ? k_dev[r] : k;
  float acc = 0.0f;
  for (int i = 0; i < A; ++i) acc = fmaf(a_state[(size_t)r * ap + i], wa[(size_t)i * H + c], acc);
  h0[(size_t)r * H + c] = acc + spart[(size_t)r * H + c] + ctab[(size_t)kk * H + c];
}

// Same, one 256-thread block per row, and the first block's LayerNorm (eps 1e-6, fast variance) of
// that row written alongside: y = LN(h0) * scale + bias.  H <= 1024.
__global__ __launch_bounds__(256) void idm_in_ln_kernel(const float* __restrict__ a_state, int ap,
                                                        const float* __restrict__ wa, int A,
                                                        const float* __restrict__ spart,
                                                        const float* __restrict__ ctab,
                                                        const int* __restrict__ k_dev, int k,
                                                        const float* __restrict__ ln_s,
                                                        const float* __restrict__ ln_b, float* __restrict__ h0,
                                                        float* __restrict__ y, int R, int H) {
  __shared__ float red[8];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int kk = k_dev ? k_dev[r] : k;
  float v[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = tid + 256 * j;
    v[j] = 0.f;
    if (c < H) {
      float acc = 0.0f;
      for (int i = 0; i < A; ++i) acc = fmaf(a_state[(size_t)r * ap + i], wa[(size_t)i * H + c], acc);
      v[j] = acc + spart[(size_t)r * H + c] + ctab[(size_t)kk * H + c];
      h0[(size_t)r * H + c] = v[j];
      s1 += v[j];
      s2 += v[j] * v[j];
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
  __syncthreads();
  s1 = (red[0] + red[1]) + (red[2] + red[3]);
  s2 = (red[4] + red[5]) + (red[6] + red[7]);
  const float mean = s1 / (float)H;
  const float var = fmaxf(s2 / (float)H - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = tid + 256 * j;
    if (c < H) y[(size_t)r * H + c] = (v[j] - mean) * rstd * ln_s[c] + ln_b[c];
  }
}

static int dense_w(ldp_handle* h, const std::string& prefix, int cin, int cout, int cout_p, hipStream_t s,
                   ConvW& out) {
  // Flax Dense kernel (in, out) == 1x1 conv kernel (1, in, out)
  auto it = h->weights.find(prefix + "/kernel");
  if (it == h->weights.end()) return fail(LDP_ESTATE, "weight '%s/kernel' was never set", prefix.c_str());
  HostTensor& t = it->second;
  if (t.shape.size() == 2) t.shape.insert(t.shape.begin(), 1);
  return make_conv(h, prefix, 1, cin, cout, cin, cout_p, nullptr, s, out);
}

int idm_finalize(ldp_handle* h, hipStream_t s) {
  const ldp_config& c = h->cfg;
  IdmState& I = h->idm;
  I = IdmState{};
  I.D = c.obs_dim; I.A = c.action_dim; I.AP = round_up(c.action_dim, 32); I.H = c.idm_hidden;
  I.NB = c.idm_blocks; I.n_train = c.idm_train_steps; I.TD = c.idm_time_dim;
  if (I.H % 128 != 0) return fail(LDP_EINVAL, "idm_hidden must be a multiple of 128");
  const std::string root = "idm/";
  const int H = I.H, A = I.A, S = 2 * I.D, TD = I.TD, NT = I.n_train;

  // cond table: MLP_0 = Dense(TD->H) -> mish -> Dense(H->H) on [cos | sin] features
  const HostTensor *m0k, *m0b, *m1k, *m1b, *w0, *b0;
  LDP_TRY(get_weight(h, root + "MLP_0/Dense_0/kernel", &m0k, {TD, H}));
  LDP_TRY(get_weight(h, root + "MLP_0/Dense_0/bias", &m0b, {H}));
  LDP_TRY(get_weight(h, root + "MLP_0/Dense_1/kernel", &m1k, {H, H}));
  LDP_TRY(get_weight(h, root + "MLP_0/Dense_1/bias", &m1b, {H}));
  LDP_TRY(get_weight(h, root + "MLPResNet_0/Dense_0/kernel", &w0, {A + S + H, H}));
  LDP_TRY(get_weight(h, root + "MLPResNet_0/Dense_0/bias", &b0, {H}));
  std::vector<float> ff;
  sinusoid_table(NT, TD, /*cos_first=*/true, ff);
  DevBuf d_ff, d_m0k, d_m0b, d_m1k, d_m1b, d_t1, d_cond, d_wc;
  LDP_TRY(upload(d_ff, ff.data(), ff.size() * 4, s));
  LDP_TRY(upload(d_m0k, m0k->data.data(), m0k->data.size() * 4, s));
  LDP_TRY(upload(d_m0b, m0b->data.data(), m0b->data.size() * 4, s));
  LDP_TRY(upload(d_m1k, m1k->data.data(), m1k->data.size() * 4, s));
  LDP_TRY(upload(d_m1b, m1b->data.data(), m1b->data.size() * 4, s));
  LDP_TRY(d_t1.alloc((size_t)NT * H * 4));
  LDP_TRY(d_cond.alloc((size_t)NT * H * 4));
  LDP_TRY(dense_launch(d_ff.f(), TD, d_m0k.f(), H, d_m0b.f(), d_t1.f(), H, NT, TD, H, 0, 1, s));
  LDP_TRY(dense_launch(d_t1.f(), H, d_m1k.f(), H, d_m1b.f(), d_cond.f(), H, NT, H, H, 0, 0, s));
  // split Dense_0 rows: [a (A) | s (2D) | cond (H)]
  LDP_TRY(upload(I.in_a.w, w0->data.data(), (size_t)A * H * 4, s));
  LDP_TRY(upload(I.w_in_s, w0->data.data() + (size_t)A * H, (size_t)S * H * 4, s));
  LDP_TRY(upload(d_wc, w0->data.data() + (size_t)(A + S) * H, (size_t)H * H * 4, s));
  LDP_TRY(upload(I.b_in, b0->data.data(), (size_t)H * 4, s));
  LDP_TRY(I.ctab.alloc((size_t)NT * H * 4));
  LDP_TRY(dense_launch(d_cond.f(), H, d_wc.f(), H, nullptr, I.ctab.f(), H, NT, H, H, 0, 0, s));

  I.blks.resize(I.NB);
  for (int i = 0; i < I.NB; ++i) {
    const std::string p = root + "MLPResNet_0/MLPResNetBlock_" + std::to_string(i);
    const HostTensor *ls, *lb;
    LDP_TRY(get_weight(h, p + "/LayerNorm_0/scale", &ls, {H}));
    LDP_TRY(get_weight(h, p + "/LayerNorm_0/bias", &lb, {H}));
    LDP_TRY(upload(I.blks[i].ln_s, ls->data.data(), (size_t)H * 4, s));
    LDP_TRY(upload(I.blks[i].ln_b, lb->data.data(), (size_t)H * 4, s));
    LDP_TRY(dense_w(h, p + "/Dense_0", H, 4 * H, 4 * H, s, I.blks[i].d0));
    LDP_TRY(dense_w(h, p + "/Dense_1", 4 * H, H, H, s, I.blks[i].d1));
  }
  LDP_TRY(dense_w(h, root + "MLPResNet_0/Dense_1", H, A, I.AP, s, I.out));
  LDP_HIP(hipStreamSynchronize(s));
  I.ready = true;
  return LDP_OK;
}

static int idm_workspace(ldp_handle* h, int R) {
  IdmState& I = h->idm;
  if (R <= I.ws_R) return LDP_OK;
  drop_graphs(h);
  const int Rp = round_up(R, 64);
  LDP_TRY(I.state.alloc((size_t)Rp * I.AP * 4));
  LDP_TRY(I.trans.alloc((size_t)Rp * 2 * I.D * 4));
  LDP_TRY(I.spart.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.h0.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.h1.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.y.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.z.alloc((size_t)Rp * 4 * I.H * 4));
  LDP_HIP(hipMemset(I.state.p, 0, (size_t)Rp * I.AP * 4));
  LDP_HIP(hipMemset(I.h0.p, 0, (size_t)Rp * I.H * 4));
  LDP_HIP(hipMemset(I.h1.p, 0, (size_t)Rp * I.H * 4));
  LDP_HIP(hipMemset(I.spart.p, 0, (size_t)Rp * I.H * 4));
  I.ws_R = Rp;
  return LDP_OK;
}

static int p1(ldp_handle* h, const ConvW& w, const float* x, int cin, float* out, int flags,
              const float* res_in, int Bq, hipStream_t s, ConvArgs* extra = nullptr) {
  ConvPlan p;
  const int cout = w.cout_p;
  p.mode = MODE_P1; p.to = 4; p.res_out = 0;
  // wide tiles when there are enough rows to fill the chip, narrower column blocks (more
  // work-groups) otherwise: no GroupNorm here, so the column block is free
  const int nsb = (Bq + 15) / 16;
  if (cout >= 1024) {
    if (nsb * (cout / 128) >= 256) { p.nwn = 8; p.ks = 1; p.cpi = 2; }
    else { p.nwn = 4; p.ks = 2; p.cpi = 2; }
  } else if (cin % 256 == 0 && nsb * (cout / 32) < 256) { p.nwn = 1; p.ks = 8; p.cpi = 2; }
  else if (cin % 128 == 0) { p.nwn = 2; p.ks = 4; p.cpi = 2; }
  else { p.nwn = 2; p.ks = 2; p.cpi = 1; }
  if (cin % p.chunk() != 0 || cout % p.bn() != 0)
    return fail(LDP_EINVAL, "IDM dense %d->%d does not tile (chunk %d, block %d)", cin, cout, p.chunk(), p.bn());
  ConvArgs a{};
  if (extra) a = *extra;
  a.xa = x; a.ca = cin; a.w = w.w.f(); a.bias = w.bias.f();
  a.res_in = res_in; a.out = out; a.B = Bq; a.cout = cout; a.flags = flags;
  const int r = tconv_launch(p, a, s);
  h->last_conv_launches++;
  h->last_total_launches++;
  if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "IDM dense launch failed (%d)", r);
  return LDP_OK;
}

// one eps-model evaluation on the padded action state; optional scheduler update / eps output
static int idm_forward_launch(ldp_handle* h, int R, const int* k_dev, int k, bool step,
                              const StepCoef* coef, const float* noise, int step_idx, float* eps_out,
                              hipStream_t s) {
  IdmState& I = h->idm;
  const int H = I.H, Bq = (R + 3) / 4;
  const bool fuse_ln0 = I.NB > 0 && H <= 1024;        // first block's LayerNorm rides in the input kernel
  if (fuse_ln0) {
    hipLaunchKernelGGL(idm_in_ln_kernel, dim3(R), dim3(256), 0, s, I.state.f(), I.AP, I.in_a.w.f(), I.A,
                       I.spart.f(), I.ctab.f(), k_dev, k, I.blks[0].ln_s.f(), I.blks[0].ln_b.f(), I.h0.f(),
                       I.y.f(), R, H);
  } else {
    dim3 grid((H + 255) / 256, R);
    hipLaunchKernelGGL(idm_in_kernel, grid, dim3(256), 0, s, I.state.f(), I.AP, I.in_a.w.f(), I.A,
                       I.spart.f(), I.ctab.f(), k_dev, k, I.h0.f(), R, H);
  }
  LDP_HIP(hipGetLastError());
  h->last_total_launches++;
  float* cur = I.h0.f();
  float* nxt = I.h1.f();
  for (int i = 0; i < I.NB; ++i) {
    if (!(i == 0 && fuse_ln0)) {
      LDP_TRY(layernorm_launch(cur, I.y.f(), I.blks[i].ln_s.f(), I.blks[i].ln_b.f(), R, H, s));
      h->last_total_launches++;
    }
    LDP_TRY(p1(h, I.blks[i].d0, I.y.f(), H, I.z.f(), EP_RELU, nullptr, Bq, s));
    const int last = (i == I.NB - 1) ? EP_RELU : 0;      // MLPResNet applies relu before Dense_1
    LDP_TRY(p1(h, I.blks[i].d1, I.z.f(), 4 * H, nxt, EP_RESIN | last, cur, Bq, s));
    std::swap(cur, nxt);
  }
  ConvArgs e{};
  e.d_real = I.A; e.rows_valid = R;
  if (coef) e.coef = *coef;
  e.noise = noise; e.seed = h->seed.as<uint64_t>(); e.step = step_idx; e.eps_out = eps_out;
  const int flags = (step ? EP_STEP : 0) | (eps_out ? EP_EPSOUT : 0);
  LDP_TRY(p1(h, I.out, cur, H, I.state.f(), flags, nullptr, Bq, s, &e));
  return LDP_OK;
}

static int idm_prepare(ldp_handle* h, const float* transition, int R, hipStream_t s) {
  IdmState& I = h->idm;
  LDP_HIP(hipMemcpyAsync(I.trans.p, transition, (size_t)R * 2 * I.D * 4, hipMemcpyDeviceToDevice, s));
  return LDP_OK;
}

static int idm_spart(ldp_handle* h, int R, hipStream_t s) {
  IdmState& I = h->idm;
  h->last_total_launches++;
  return dense_launch(I.trans.f(), 2 * I.D, I.w_in_s.f(), I.H, I.b_in.f(), I.spart.f(), I.H, R, 2 * I.D,
                      I.H, 0, 0, s);
}

}  // namespace ldp

using namespace ldp;

extern "C" {

int ldp_idm_forward(ldp_handle* h, const float* sT, const float* a, const int32_t* k_dev, int32_t k,
                    float* eps, int32_t R, void* stream) {
  if (!h || !sT || !a || !eps || R <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->idm.ready) return fail(LDP_ESTATE, "idm weights not finalized");
  IdmState& I = h->idm;
  if (!k_dev && (k < 0 || k >= I.n_train)) return fail(LDP_EINVAL, "timestep %d out of range", k);
  hipStream_t s = (hipStream_t)stream;
  LDP_TRY(idm_workspace(h, R));
  h->last_conv_launches = h->last_total_launches = 0;
  LDP_TRY(idm_prepare(h, sT, R, s));
  LDP_TRY(pad_rows_launch(a, I.state.f(), R, I.A, I.AP, s));
  LDP_TRY(idm_spart(h, R, s));
  return idm_forward_launch(h, R, k_dev, k, false, nullptr, nullptr, 0, eps, s);
}

int ldp_idm_sample(ldp_handle* h, const float* transition, const float* a_init, const float* step_noise,
                   uint64_t seed, int64_t row_offset, int32_t sampler, int32_t n_steps, float* out,
                   int32_t R, int32_t use_graph, void* stream) {
  if (!h || !transition || !out || R <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->idm.ready) return fail(LDP_ESTATE, "idm weights not finalized");
  IdmState& I = h->idm;
  if (sampler == LDP_SAMPLER_DDPM) {
    if (n_steps != I.n_train)
      return fail(LDP_EINVAL, "DDPM visits every training timestep: n_steps must be %d (got %d)",
                  I.n_train, n_steps);
  } else if (sampler == LDP_SAMPLER_DDIM) {
    if (n_steps <= 0 || I.n_train % n_steps != 0)
      return fail(LDP_EINVAL, "DDIM needs n_steps | %d (got %d)", I.n_train, n_steps);
  } else {
    return fail(LDP_EINVAL, "unknown sampler %d", sampler);
  }
  hipStream_t s = (hipStream_t)stream;
  LDP_TRY(idm_workspace(h, R));
  h->last_conv_launches = h->last_total_launches = 0;
  const bool explicit_noise = step_noise != nullptr && sampler == LDP_SAMPLER_DDPM;
  const size_t per_step = (size_t)R * I.A;
  LDP_TRY(idm_prepare(h, transition, R, s));
  // Philox stream of the IDM is decorrelated from the planner's by flipping the seed's top bit
  // (rows are handled in quads, so the Philox key is the global quad index)
  if (row_offset % 4 != 0) return fail(LDP_EINVAL, "row_offset must be a multiple of 4");
  LDP_TRY(set_seed_launch(h->seed.as<uint64_t>(), seed ^ 0x8000000000000000ull, row_offset / 4, s));
  if (a_init) LDP_TRY(pad_rows_launch(a_init, I.state.f(), R, I.A, I.AP, s));
  else LDP_TRY(philox_init_launch(I.state.f(), 4, (R + 3) / 4, I.A, I.AP, h->seed.as<uint64_t>(), s));
  if (explicit_noise) {
    if (per_step * n_steps * 4 > I.noise.bytes) drop_graphs(h);
    LDP_TRY(I.noise.alloc(per_step * n_steps * 4));
    LDP_HIP(hipMemcpyAsync(I.noise.p, step_noise, per_step * n_steps * 4, hipMemcpyDeviceToDevice, s));
  }
  std::vector<StepCoef> coefs;
  make_step_coefs(I.n_train, n_steps, sampler, coefs);
  auto enqueue_loop = [&](hipStream_t q) -> int {
    LDP_TRY(idm_spart(h, R, q));
    for (int i = 0; i < n_steps; ++i) {
      const float* nz = explicit_noise ? I.noise.f() + per_step * i : nullptr;
      LDP_TRY(idm_forward_launch(h, R, nullptr, (int)coefs[i].t, true, &coefs[i], nz, i, nullptr, q));
    }
    return LDP_OK;
  };
  if (!use_graph) {
    LDP_TRY(enqueue_loop(s));
  } else {
    GraphKey key{1, R, n_steps, sampler, explicit_noise ? 1 : 0};
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
      hipGraph_t graph = nullptr;
      LDP_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
      const int r = enqueue_loop(h->cap_stream);
      hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
      if (r != LDP_OK) { if (graph) (void)hipGraphDestroy(graph); return r; }
      if (e != hipSuccess) return fail(LDP_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
      hipGraphExec_t exec = nullptr;
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) return fail(LDP_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      it = h->graphs.emplace(key, GraphEntry{exec, h->last_conv_launches, h->last_total_launches}).first;
    } else {
      h->last_conv_launches = it->second.conv_launches;
      h->last_total_launches = it->second.total_launches;
    }
    LDP_HIP(hipGraphLaunch(it->second.exec, s));
  }
  LDP_TRY(unpad_rows_launch(I.state.f(), out, R, I.A, I.AP, s));
  return LDP_OK;
}

}  // extern "C"
