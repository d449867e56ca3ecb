// engine.hip -- lifecycle, weight store, packing, planner orchestration (eager + hipGraph) and
// the planner half of the C ABI declared in include/ldp_hip.h.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ldp {

// ---------------------------------------------------------------------------------------------
// constant tables (mirror of latent_diffusion_planning_amd/schedule.py; tests compare the two)
// ---------------------------------------------------------------------------------------------
static void betas_squaredcos(int n, std::vector<float>& betas, std::vector<float>& alphas,
                             std::vector<float>& acp) {
  auto abar = [](double t) {
    const double c = std::cos((t + 0.008) / 1.008 * M_PI / 2.0);
    return c * c;
  };
  betas.resize(n);
  alphas.resize(n);
  acp.resize(n);
  float run = 1.0f;
  for (int i = 0; i < n; ++i) {
    const double b = std::min(1.0 - abar((double)(i + 1) / n) / abar((double)i / n), 0.999);
    betas[i] = (float)b;
    alphas[i] = 1.0f - betas[i];
    run = run * alphas[i];              // float32 cumprod
    acp[i] = run;
  }
}

void make_step_coefs(int n_train, int n_steps, int sampler, std::vector<StepCoef>& out) {
  std::vector<float> betas, alphas, acp;
  betas_squaredcos(n_train, betas, alphas, acp);
  out.assign(n_steps, StepCoef{});
  const int stride = n_train / n_steps;
  for (int i = 0; i < n_steps; ++i) {
    const int t = (n_steps - 1 - i) * stride;
    const double a_t = acp[t];
    StepCoef c{};
    c.t = (float)t;
    c.inv_sqrt_ab = (float)(1.0 / std::sqrt(a_t));
    c.sqrt_1mab = (float)std::sqrt(1.0 - a_t);
    if (sampler == LDP_SAMPLER_DDPM) {
      const double a_prev = t > 0 ? (double)acp[t - 1] : 1.0;
      const double beta = betas[t], alpha = alphas[t];
      c.c_x0 = (float)(std::sqrt(a_prev) * beta / (1.0 - a_t));
      c.c_x = (float)(std::sqrt(alpha) * (1.0 - a_prev) / (1.0 - a_t));
      const double var = std::max((1.0 - a_prev) / (1.0 - a_t) * beta, 1e-20);
      c.sigma = t > 0 ? (float)std::sqrt(var) : 0.0f;
    } else {
      const int tp = t - stride;
      const double a_prev = tp >= 0 ? (double)acp[tp] : 1.0;
      c.c_x0 = (float)std::sqrt(a_prev);
      c.c_eps = (float)std::sqrt(1.0 - a_prev);
    }
    out[i] = c;
  }
}

void sinusoid_table(int n, int dim, bool cos_first, std::vector<float>& out) {
  const int half = dim / 2;
  const float step = std::log(10000.0f) / (float)(half - 1);     // float32 like the traced graph
  std::vector<float> f(half);
  for (int j = 0; j < half; ++j) f[j] = std::exp((float)j * -step);
  out.assign((size_t)n * dim, 0.f);
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < half; ++j) {
      const float arg = (float)k * f[j];
      const float s = (float)std::sin((double)arg), c = (float)std::cos((double)arg);
      out[(size_t)k * dim + j] = cos_first ? c : s;
      out[(size_t)k * dim + half + j] = cos_first ? s : c;
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing: Flax (nj, cin, cout) -> [chunk][tap][cout_p/16][lane = kq*16 + n][s]
// holding W[tap][chunk*16 + 4*kq + s][nblk*16 + n]
// ---------------------------------------------------------------------------------------------
std::vector<float> pack_conv(const float* w, int nj, int cin, int cout, int cin_p, int cout_p) {
  const int nchunk = cin_p / 16, nblk = cout_p / 16;
  std::vector<float> p((size_t)nchunk * nj * nblk * 256, 0.0f);
  for (int gc = 0; gc < nchunk; ++gc)
    for (int j = 0; j < nj; ++j)
      for (int nb = 0; nb < nblk; ++nb) {
        float* dst = p.data() + (((size_t)gc * nj + j) * nblk + nb) * 256;
        for (int lane = 0; lane < 64; ++lane) {
          const int kq = lane >> 4, n = lane & 15;
          const int co = nb * 16 + n;
          for (int s = 0; s < 4; ++s) {
            const int ci = gc * 16 + 4 * kq + s;
            if (ci < cin && co < cout) dst[lane * 4 + s] = w[((size_t)j * cin + ci) * cout + co];
          }
        }
      }
  return p;
}

int get_weight(ldp_handle* h, const std::string& path, const HostTensor** out,
               std::initializer_list<int64_t> shape) {
  auto it = h->weights.find(path);
  if (it == h->weights.end()) return fail(LDP_ESTATE, "weight '%s' was never set", path.c_str());
  const HostTensor& t = it->second;
  if (t.shape.size() != shape.size() || !std::equal(shape.begin(), shape.end(), t.shape.begin())) {
    std::string got, want;
    for (auto s : t.shape) got += std::to_string(s) + ",";
    for (auto s : shape) want += std::to_string(s) + ",";
    return fail(LDP_EINVAL, "weight '%s' has shape (%s) expected (%s)", path.c_str(), got.c_str(),
                want.c_str());
  }
  *out = &t;
  return LDP_OK;
}

static int upload_padded(DevBuf& dst, const float* src, int n, int n_p, float fill, hipStream_t s) {
  std::vector<float> tmp(n_p, fill);
  std::copy(src, src + n, tmp.begin());
  LDP_TRY(dst.alloc((size_t)n_p * 4));
  LDP_HIP(hipMemcpy(dst.p, tmp.data(), (size_t)n_p * 4, hipMemcpyHostToDevice));
  (void)s;
  return LDP_OK;
}

// prefix: ".../Conv_0" (expects <prefix>/kernel, <prefix>/bias); gn_prefix: ".../GroupNorm_0" or null
int make_conv(ldp_handle* h, const std::string& prefix, int nj, int cin, int cout, int cin_p,
              int cout_p, const char* gn_prefix, hipStream_t s, ConvW& out) {
  const HostTensor *k = nullptr, *b = nullptr;
  LDP_TRY(get_weight(h, prefix + "/kernel", &k, {nj, cin, cout}));
  LDP_TRY(get_weight(h, prefix + "/bias", &b, {cout}));
  std::vector<float> packed = pack_conv(k->data.data(), nj, cin, cout, cin_p, cout_p);
  LDP_TRY(out.w.alloc(packed.size() * 4));
  LDP_HIP(hipMemcpy(out.w.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  LDP_TRY(upload_padded(out.bias, b->data.data(), cout, cout_p, 0.f, s));
  out.nj = nj; out.cin = cin; out.cout = cout; out.cin_p = cin_p; out.cout_p = cout_p;
  out.has_gn = gn_prefix != nullptr;
  if (gn_prefix) {
    const HostTensor *gs = nullptr, *gb = nullptr;
    LDP_TRY(get_weight(h, std::string(gn_prefix) + "/scale", &gs, {cout}));
    LDP_TRY(get_weight(h, std::string(gn_prefix) + "/bias", &gb, {cout}));
    LDP_TRY(upload_padded(out.gn_scale, gs->data.data(), cout, cout_p, 1.f, s));
    LDP_TRY(upload_padded(out.gn_bias, gb->data.data(), cout, cout_p, 0.f, s));
  }
  return LDP_OK;
}

// Re-packs the conv's weights with the block's 1x1 residual projection as one extra "tap" per chunk
// (tconv RES_OUT layout): one weight stream, one allocation.
static int add_res_proj(ldp_handle* h, const std::string& conv_prefix, const std::string& prefix, int cin,
                        int cout, int cin_p, ConvW& c) {
  const HostTensor *k = nullptr, *b = nullptr, *kc = nullptr;
  LDP_TRY(get_weight(h, prefix + "/kernel", &k, {1, cin, cout}));
  LDP_TRY(get_weight(h, prefix + "/bias", &b, {cout}));
  LDP_TRY(get_weight(h, conv_prefix + "/kernel", &kc, {c.nj, cin, cout}));
  std::vector<float> both(kc->data);
  both.insert(both.end(), k->data.begin(), k->data.end());
  std::vector<float> packed = pack_conv(both.data(), c.nj + 1, cin, cout, cin_p, c.cout_p);
  LDP_TRY(c.w.alloc(packed.size() * 4));
  LDP_HIP(hipMemcpy(c.w.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  LDP_TRY(c.bres.alloc((size_t)cout * 4));
  LDP_HIP(hipMemcpy(c.bres.p, b->data.data(), (size_t)cout * 4, hipMemcpyHostToDevice));
  c.has_res = true;
  return LDP_OK;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// instantiation choice
// ---------------------------------------------------------------------------------------------
// cs (in/out): requested column split of a GroupNorm group over work-groups; reset to 1 when the
// shape has no half-width instantiation.
// mb_want = 2: two 16-sample row blocks per work-group where such an instantiation exists (k=5, T <= 4)
static int pick_plan(int mode, int to, int cout, int cin_total, int ca, bool res_out, ConvPlan& p,
                     int* cs_io = nullptr, int mb_want = 1, bool up_full_depth = false) {
  const int gw = cout >= 256 ? cout / 8 : 32;       // one GroupNorm group (n_groups = 8)
  if (cs_io && *cs_io == 4) {                        // quarter groups (small batches)
    const int qb = gw / 4;
    int ks4 = 0, cpi4 = 0;
    if (mode == MODE_K5) {
      // half-depth chunks: twice the K iterations, so the K split over work-groups goes twice as wide and the MFMA
      // chain of a work-group (the main loop at these batch sizes: the tile costs the same whether 5 or 16 of its
      // samples are real) halves: -6 % plan latency at 1..16 plans, -1 % at 128
      if (to == 2 && qb == 32) { ks4 = 4; cpi4 = 2; }
      else if (to == 4 && qb == 16) { ks4 = 8; cpi4 = 1; }
      // pred_horizon 16 (round 3): its 512-channel level runs at T = 8 and its 1024-channel level at T = 4.  Without
      // these two a T = 16 loop had 8 work-groups per sample block whatever the batch (155 ms per 100 steps at 16 plans)
      else if (to == 8 && qb == 16) { ks4 = 8; cpi4 = 1; }
      else if (to == 4 && qb == 32) { ks4 = 4; cpi4 = res_out ? 1 : 2; }
    }
    const int chunk4 = 16 * ks4 * cpi4;
    if (ks4 && cin_total % chunk4 == 0 && ca % chunk4 == 0) {
      p = ConvPlan{mode, to, qb / 16, ks4, cpi4, res_out ? 1 : 0};
      return LDP_OK;
    }
    *cs_io = 2;
  }
  if (cs_io && *cs_io == 2) {
    const int hb = gw / 2;
    int ks2 = 0, cpi2 = 0;
    if (mode == MODE_K5) {
      if (to == 8 && hb == 16) { ks2 = (cin_total % 128 == 0) ? 8 : 4; cpi2 = 1; }      // 64-channel first layer: four K slices
      // chunk depth (CPI) per shape by measurement (tools/layer_times.py): the kernels that also carry the
      // residual projection and the 16-/32-column tiles run faster on half-depth chunks (more, shorter
      // pipeline stages), the wide plain ones on full depth
      else if (to == 4 && hb == 32) { ks2 = 4; cpi2 = res_out ? 1 : 2; }
      else if (to == 2 && hb == 64) { ks2 = 2; cpi2 = res_out ? 2 : 4; }
      else if (to == 4 && hb == 16) { ks2 = 8; cpi2 = 1; }
      else if (to == 2 && hb == 32) { ks2 = 4; cpi2 = 2; }
      else if (to == 8 && hb == 32) { ks2 = 4; cpi2 = 1; }                   // pred_horizon 16: (8, 512)
      else if (to == 4 && hb == 64) { ks2 = 2; cpi2 = 2; }                   // pred_horizon 16: (4, 1024)
      else if (to == 16 && hb == 16) { ks2 = 4; cpi2 = 1; }                  // pred_horizon 16: (16, 256): two 80 KB staging buffers = all of a CU's LDS
    } else if (mode == MODE_DOWN) {
      if (to == 4 && hb == 16) { ks2 = 8; cpi2 = 1; }
      else if (to == 2 && hb == 32) { ks2 = 4; cpi2 = 2; }
      else if (to == 8 && hb == 16) { ks2 = 4; cpi2 = 1; }                   // pred_horizon 16: 16 -> 8 positions at 256 channels
    } else if (mode == MODE_UP) {
      // half-depth chunks (round 3): with 256-channel chunks the whole T=8 conv and half the T=4 one sat in the
      // prologue -- 128 KB requested before the first MFMA (tools/timeline.py: prologue 2.0-2.4 us against 1.2)
      if (to == 4 && hb == 32) { ks2 = 4; cpi2 = (cin_total % 256 == 0 && up_full_depth) ? 4 : 2; }
      else if (to == 8 && hb == 16) { ks2 = 8; cpi2 = (cin_total % 256 == 0 && up_full_depth) ? 2 : 1; }
      else if (to == 16 && hb == 16) { ks2 = 4; cpi2 = 1; }                  // pred_horizon 16: 8 -> 16 positions at 256 channels
    } else if (mode == MODE_P1) {
      if (to == 8 && hb == 16) { ks2 = 8; cpi2 = 1; }
    }
    const int chunk2 = 16 * ks2 * cpi2;
    if (ks2 && cin_total % chunk2 == 0 && ca % chunk2 == 0) {
      p = ConvPlan{mode, to, hb / 16, ks2, cpi2, res_out ? 1 : 0};
      return LDP_OK;
    }
    *cs_io = 1;
  }
  const int bn = gw;
  int nwn = bn / 16, ks = 0, cpi = 0;
  if (mode == MODE_K5) {
    if (to == 8 && bn == 32) { ks = (cin_total % 64 == 0) ? 4 : 2; cpi = 1; }
    else if (to == 8 && bn == 64) { ks = 2; cpi = 1; }
    else if (to == 4 && bn == 64) { ks = 2; cpi = 2; }
    else if (to == 4 && bn == 32) { ks = 4; cpi = res_out ? 1 : 2; }
    else if (to == 4 && bn == 128) { ks = 1; cpi = 2; }
    else if (to == 2 && bn == 128) { ks = 1; cpi = 4; }
    else if (to == 2 && bn == 64) { ks = 2; cpi = res_out ? 2 : 4; }
    else if (to == 16 && bn == 32) { ks = 2; cpi = 1; }
  } else if (mode == MODE_DOWN) {
    if (to == 4 && bn == 32) { ks = 4; cpi = 1; }
    else if (to == 2 && bn == 64) { ks = 2; cpi = 2; }
    else if (to == 8 && bn == 32) { ks = 2; cpi = 1; }
    else if (to == 4 && bn == 64) { ks = 2; cpi = 1; }
  } else if (mode == MODE_UP) {
    if (to == 4 && bn == 64) { ks = 2; cpi = 4; }
    else if (to == 8 && bn == 32) { ks = 4; cpi = 2; }
    else if (to == 8 && bn == 64) { ks = 2; cpi = 2; }
    else if (to == 16 && bn == 32) { ks = 2; cpi = 1; }
  } else if (mode == MODE_P1) {
    if (to == 8 && bn == 32) { ks = 4; cpi = 1; }
    else if (to == 16 && bn == 32) { ks = 2; cpi = 1; }
    else if (to == 4 && bn == 128) { ks = 1; cpi = 2; }
    else if (to == 4 && bn == 32) { if (cin_total % 128 == 0) { ks = 4; cpi = 2; } else { ks = 2; cpi = 1; } }    else if (to == 2 && bn == 32) { ks = 4; cpi = 1; }
  }
  if (ks == 0)
    return fail(LDP_EINVAL, "no MFMA conv instantiation for mode=%d T_out=%d C_out=%d (group width %d)",
                mode, to, cout, bn);
  p = ConvPlan{mode, to, nwn, ks, cpi, res_out ? 1 : 0};
  if (mb_want == 2 && mode == MODE_K5 &&
      ((to == 4 && !res_out && (bn == 64 || bn == 32)) || (to == 2 && (bn == 128 || bn == 64))))   // T=4 + projection: out of registers
    p.mb = 2;
  if (cin_total % p.chunk() != 0 || ca % p.chunk() != 0)
    return fail(LDP_EINVAL, "input channels (%d, first part %d) not a multiple of the %d-channel chunk "
                "(mode=%d T_out=%d C_out=%d)", cin_total, ca, p.chunk(), mode, to, cout);
  return LDP_OK;
}

static int launch_conv(ldp_handle* h, const ConvPlan& p, const ConvArgs& a_in, hipStream_t s) {
  const int dbg = h->opt.dbg, repeat = h->opt.repeat;      // timing ablations (ldp_set_option), 0 / 1 in production
  ConvArgs a = a_in;
  a.dbg = dbg;
  // tools/timeline.py: stamps of launch i of the call go to slot i (1 MiB each); ignored by production builds
  if (h->opt.timeline_ptr && h->last_conv_launches < 64)
    a.tl = reinterpret_cast<unsigned long long*>(h->opt.timeline_ptr) + (size_t)h->last_conv_launches * 131072;
  if (!(a.flags & EP_STEP))              // idempotent launches may be repeated (L2-warm timing experiments)
    for (int i = 1; i < repeat; ++i) (void)tconv_launch(p, a, s);
  const int r = tconv_launch(p, a, s);
  h->last_conv_launches++;
  h->last_total_launches++;
  if (r == -100)
    return fail(LDP_EINVAL, "conv instantiation missing: mode=%d TO=%d NWN=%d KS=%d CPI=%d res=%d",
                p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out);
  if (r != 0) return fail(LDP_EHIP, "conv launch failed: %s", hipGetErrorString((hipError_t)r));
  return LDP_OK;
}

// ---------------------------------------------------------------------------------------------
// planner: finalize
// ---------------------------------------------------------------------------------------------
int planner_finalize(ldp_handle* h, hipStream_t s) {
  const ldp_config& c = h->cfg;
  PlannerState& P = h->pl;
  P = PlannerState{};
  P.D = c.obs_dim; P.DP = round_up(c.obs_dim, 32); P.G = c.global_cond_dim; P.T = c.pred_horizon;
  P.L = c.n_levels; P.E = c.step_embed_dim; P.n_train = c.planner_train_steps;
  P.C0P = P.DP <= 32 ? 128 : round_up(P.DP, 64);     // <= 32 stored channels: eight K-slice waves, six of them skip (tconv.hpp SKIPZ)       // the first conv's virtual input chunk: 64 channels for D <= 64 (32 stored), else 128
  if (c.kernel_size != 5 || c.n_groups != 8)
    return fail(LDP_EINVAL, "only kernel_size=5 / n_groups=8 kernels are built (got %d / %d)",
                c.kernel_size, c.n_groups);
  if (P.L < 2 || P.L > LDP_MAX_LEVELS) return fail(LDP_EINVAL, "n_levels must be 2..%d", LDP_MAX_LEVELS);
  if (P.T % (1 << (P.L - 1)) != 0)
    return fail(LDP_EINVAL, "pred_horizon=%d is not a multiple of %d: the reference U-Net's skip "
                "connections cannot be concatenated (SURVEY.md fact 5)", P.T, 1 << (P.L - 1));
  P.dims.assign(c.down_dims, c.down_dims + P.L);
  // the conv tiles are one (or a half / quarter of one) GroupNorm group wide and at least 16 columns:
  // a level narrower than 8 groups x 32 channels, or not a whole number of 16-column blocks per group,
  // has no instantiation -- refuse it here instead of normalising over the wrong channel set
  for (int l = 0; l < P.L; ++l)
    if (P.dims[l] < 256 || P.dims[l] % 128 != 0)
      return fail(LDP_EINVAL, "down_dims[%d] = %d: every level must be >= 256 and a multiple of 128 "
                  "(8 GroupNorm groups of whole 16-column MFMA blocks)", l, P.dims[l]);
  const std::string root = "planner/";

  // block list in Flax construction order
  struct BS { int cin, cout; bool proj; };
  std::vector<BS> bs;
  int cin = P.D;
  for (int l = 0; l < P.L; ++l) { bs.push_back({cin, P.dims[l], true}); bs.push_back({P.dims[l], P.dims[l], false}); cin = P.dims[l]; }
  bs.push_back({cin, cin, false});
  bs.push_back({cin, cin, false});
  for (int l = P.L - 2; l >= 0; --l) { bs.push_back({2 * cin, P.dims[l], true}); bs.push_back({P.dims[l], P.dims[l], false}); cin = P.dims[l]; }

  P.blocks.resize(bs.size());
  int F = 0;
  for (size_t i = 0; i < bs.size(); ++i) {
    ResBlock& b = P.blocks[i];
    b.cin = bs[i].cin; b.cout = bs[i].cout; b.proj = bs[i].proj; b.film_off = F;
    F += 2 * b.cout;
    const std::string p = root + "ConditionalResidualBlock1D_" + std::to_string(i);
    const int cin_p = (i == 0) ? P.C0P : b.cin;     // first layer: 128 virtual channels, 32 stored
    LDP_TRY(make_conv(h, p + "/Conv1dBlock_0/Conv_0", 5, b.cin, b.cout, cin_p, b.cout,
                      (p + "/Conv1dBlock_0/GroupNorm_0").c_str(), s, b.c1));
    LDP_TRY(make_conv(h, p + "/Conv1dBlock_1/Conv_0", 5, b.cout, b.cout, b.cout, b.cout,
                      (p + "/Conv1dBlock_1/GroupNorm_0").c_str(), s, b.c2));
    if (b.proj) LDP_TRY(add_res_proj(h, p + "/Conv1dBlock_0/Conv_0", p + "/Conv_0", b.cin, b.cout, cin_p, b.c1));
  }
  P.F = F;
  P.down.resize(P.L - 1);
  P.up.resize(P.L - 1);
  for (int l = 0; l < P.L - 1; ++l) {
    const int cd = P.dims[l];
    LDP_TRY(make_conv(h, root + "Downsample1d_" + std::to_string(l) + "/Conv_0", 3, cd, cd, cd, cd,
                      nullptr, s, P.down[l]));
    const int cu = P.dims[P.L - 2 - l];
    LDP_TRY(make_conv(h, root + "Upsample1d_" + std::to_string(l) + "/ConvTranspose_0", 4, cu, cu, cu,
                      cu, nullptr, s, P.up[l]));
  }
  const int c0 = P.dims[0];
  LDP_TRY(make_conv(h, root + "Conv1dBlock_0/Conv_0", 5, c0, c0, c0, c0,
                    (root + "Conv1dBlock_0/GroupNorm_0").c_str(), s, P.fin_block));
  LDP_TRY(make_conv(h, root + "Conv_0", 1, c0, P.D, c0, P.DP, nullptr, s, P.fin_conv));

  // ---- timestep-only tables: temb = Dense_1(Mish(Dense_0(sinusoid(k)))), FiLM time parts ------
  const int E = P.E, NT = P.n_train;
  const HostTensor *d0k, *d0b, *d1k, *d1b;
  LDP_TRY(get_weight(h, root + "Dense_0/kernel", &d0k, {E, 4 * E}));
  LDP_TRY(get_weight(h, root + "Dense_0/bias", &d0b, {4 * E}));
  LDP_TRY(get_weight(h, root + "Dense_1/kernel", &d1k, {4 * E, E}));
  LDP_TRY(get_weight(h, root + "Dense_1/bias", &d1b, {E}));
  std::vector<float> sintab;
  sinusoid_table(NT, E, /*cos_first=*/false, sintab);
  DevBuf d_sin, d_w0, d_b0, d_w1, d_b1, d_h, d_temb, d_wt, d_bt;
  LDP_TRY(upload(d_sin, sintab.data(), sintab.size() * 4, s));
  LDP_TRY(upload(d_w0, d0k->data.data(), d0k->data.size() * 4, s));
  LDP_TRY(upload(d_b0, d0b->data.data(), d0b->data.size() * 4, s));
  LDP_TRY(upload(d_w1, d1k->data.data(), d1k->data.size() * 4, s));
  LDP_TRY(upload(d_b1, d1b->data.data(), d1b->data.size() * 4, s));
  LDP_TRY(d_h.alloc((size_t)NT * 4 * E * 4));
  LDP_TRY(d_temb.alloc((size_t)NT * E * 4));
  LDP_TRY(dense_launch(d_sin.f(), E, d_w0.f(), 4 * E, d_b0.f(), d_h.f(), 4 * E, NT, E, 4 * E, 0, 1, s));
  LDP_TRY(dense_launch(d_h.f(), 4 * E, d_w1.f(), E, d_b1.f(), d_temb.f(), E, NT, 4 * E, E, 0, 0, s));

  // concatenated FiLM Dense kernels: time rows (E, F) + bias (F) and global-cond rows (G, F)
  std::vector<float> wt((size_t)E * F), bt(F), wg((size_t)std::max(P.G, 1) * F, 0.f);
  for (size_t i = 0; i < bs.size(); ++i) {
    const ResBlock& b = P.blocks[i];
    const HostTensor *fk, *fb;
    const std::string p = root + "ConditionalResidualBlock1D_" + std::to_string(i) + "/Dense_0";
    LDP_TRY(get_weight(h, p + "/kernel", &fk, {E + P.G, 2 * b.cout}));
    LDP_TRY(get_weight(h, p + "/bias", &fb, {2 * b.cout}));
    const int w2 = 2 * b.cout;
    for (int r = 0; r < E; ++r)
      std::copy(fk->data.begin() + (size_t)r * w2, fk->data.begin() + (size_t)(r + 1) * w2,
                wt.begin() + (size_t)r * F + b.film_off);
    for (int r = 0; r < P.G; ++r)
      std::copy(fk->data.begin() + (size_t)(E + r) * w2, fk->data.begin() + (size_t)(E + r + 1) * w2,
                wg.begin() + (size_t)r * F + b.film_off);
    std::copy(fb->data.begin(), fb->data.end(), bt.begin() + b.film_off);
  }
  LDP_TRY(upload(d_wt, wt.data(), wt.size() * 4, s));
  LDP_TRY(upload(d_bt, bt.data(), bt.size() * 4, s));
  LDP_TRY(upload(P.wfilm_g, wg.data(), wg.size() * 4, s));
  LDP_TRY(P.film_t.alloc((size_t)NT * F * 4));
  LDP_TRY(dense_launch(d_temb.f(), E, d_wt.f(), F, d_bt.f(), P.film_t.f(), F, NT, E, F, 1, 0, s));
  LDP_HIP(hipStreamSynchronize(s));     // temporaries go out of scope
  P.ready = true;
  return LDP_OK;
}

// ---------------------------------------------------------------------------------------------
// planner: workspaces
// ---------------------------------------------------------------------------------------------
void drop_graphs(ldp_handle* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);
  h->graphs.clear();
}

int planner_workspace(ldp_handle* h, int B) {
  PlannerState& P = h->pl;
  if (B <= P.ws_B) return LDP_OK;
  // workspaces are baked into captured graphs: drop them when buffers move
  drop_graphs(h);
  const int Bp = round_up(B, 16);
  size_t act = 0;
  for (int l = 0; l < P.L; ++l) act = std::max(act, (size_t)(P.T >> l) * P.dims[l]);
  act *= (size_t)Bp * 4;
  LDP_TRY(P.state.alloc((size_t)Bp * P.T * P.DP * 4));
  LDP_TRY(P.cond.alloc((size_t)Bp * std::max(P.G, 1) * 4));
  LDP_TRY(P.film_g.alloc((size_t)Bp * P.F * 4));
  LDP_TRY(P.bufA.alloc(act));
  LDP_TRY(P.bufB.alloc(act));
  LDP_TRY(P.bufC.alloc(act));
  LDP_TRY(P.bufR.alloc(act));
  P.skip.resize(P.L);
  for (int l = 0; l < P.L; ++l) LDP_TRY(P.skip[l].alloc(act));
  // GroupNorm statistics exchange slabs of the column-split convs: one slab per conv launch of an
  // evaluation; [sample block][8 groups][2 halves][16 samples][2] 8-byte granules, tags start at 0
  P.xchg_stride = (size_t)((Bp + 31) / 32 * 2) * 8 * 4 * 32;      // whole pairs of row blocks (MB = 2 work-groups)
  LDP_TRY(P.xchg.alloc(P.xchg_stride * 8 * 64));            // 64 slots
  LDP_HIP(hipMemset(P.xchg.p, 0, P.xchg_stride * 8 * 64));
  // K split over work-groups (B <= 16): one partial-tile slab shared by all launches (they are serialised),
  // per-launch flag rows (tags repeat within a step)
  LDP_TRY(P.kw_slab.alloc((size_t)256 * 2 * 16 * 512 * 8));      // {value, tag} granules (<= 256 (tile, part) pairs x 2 tiles of <= 16 x 512), tags start at 0
  LDP_HIP(hipMemset(P.kw_slab.p, 0, P.kw_slab.bytes));
  P.ws_B = Bp;
  return LDP_OK;
}

// ---------------------------------------------------------------------------------------------
// planner: one U-Net evaluation = 30 fused conv launches (pred_horizon 8, 3 levels)
// ---------------------------------------------------------------------------------------------
namespace {
struct Fwd {
  ldp_handle* h;
  PlannerState& P;
  int B;
  const int* k_dev;
  int k;
  hipStream_t s;
  int step_idx;
  int cs_want;     // 2: split every GroupNorm group over two work-groups (fills the chip at B <= 256)
  int mb_want;     // 2: two row blocks per work-group (the grid still fills the chip)
  int slot = 0;
  int kslot = 0;

  int conv(const ConvW& w, int mode, int to, const float* xa, int ca, const float* xb, int cb,
           float* out, int flags, const ResBlock* film, const float* res_in, float* res_out) {
    ConvPlan p;
    int cs = cs_want;
    int ca_real = 0;
    if (xa == P.state.f()) {              // the loop state stores DP channels; the first conv's chunk is wider
      ca_real = ca;
      ca = P.C0P;
    }
    if (w.has_res != (res_out != nullptr))
      return fail(LDP_EINVAL, "conv packed %s its residual projection launched %s it", w.has_res ? "with" : "without",
                  res_out ? "with" : "without");
    LDP_TRY(pick_plan(mode, to, w.cout_p, ca + cb, ca, res_out != nullptr, p, &cs, mb_want, h->opt.up_full_depth != 0));
    ConvArgs a{};
    a.cs = cs;
    a.ca_real = ca_real;
    a.ctl = h->ctl_planner();
    a.fault = h->fault_dev;
    a.step = step_idx;
    if (cs > 1 && (flags & EP_GN)) {
      if (slot >= 64) return fail(LDP_EINVAL, "more than 64 GroupNorm convs per evaluation");
      a.xchg = P.xchg.as<unsigned long long>() + (size_t)slot * P.xchg_stride;
      ++slot;
    }
    // few sample blocks (B <= 128): split the input channels over up to 8 work-groups per tile while the
    // grid still fits the chip
    const bool no_kw = h->opt.no_kw || h->safe_mode;
    const int kw_min_it = h->opt.kw_min_it, kw_bmax = h->opt.kw_bmax;
    const int cpi_full = p.cpi;
    if (!no_kw && B <= kw_bmax && mode == MODE_UP) {          // transposed convs: half-depth chunks so that the K split has iterations to share
      if (p.to == 4 && p.nwn == 2 && p.ks == 4 && p.cpi == 4) p.cpi = 2;
      else if (p.to == 8 && p.nwn == 1 && p.ks == 8 && p.cpi == 2) p.cpi = 1;
    }
    const int cpi_nokw = (mode == MODE_UP && !h->opt.up_full_depth) ? p.cpi : cpi_full;
    if (!no_kw && B <= kw_bmax && (mode == MODE_K5 || mode == MODE_DOWN || mode == MODE_UP) && tconv_kw_ok(p.mode, p.to, p.nwn, p.mb)) {
      const int wgs = ((B + 15) / 16) * (w.cout_p / p.bn()), nit = (ca + cb) / p.chunk();
      int kw = 1;
      while (kw < KW_MAX && wgs * kw * 2 <= std::min(h->n_cu, 256) && nit % (kw * 2) == 0 && nit / (kw * 2) >= kw_min_it) kw *= 2;
      static const ConvPlan kws_plans[] = {{MODE_K5, 8, 1, 8, 1, 0}, {MODE_K5, 4, 1, 8, 2, 0}, {MODE_K5, 2, 2, 4, 4, 0},
                                           {MODE_K5, 2, 2, 4, 2, 0}, {MODE_K5, 4, 1, 8, 1, 0}, {MODE_K5, 4, 2, 4, 2, 0}, {MODE_K5, 4, 2, 4, 1, 0},
                                           // pred_horizon 16's (4, 512) stride-2 and (8, 512) transposed convs.  NOT its 16-position tiles: 512 elements
                                           // per sample make the partial-tile exchange dearer than the K it saves (16 plans: 34 -> 60 us with the projection)
                                           {MODE_DOWN, 4, 4, 2, 1, 0}, {MODE_UP, 8, 4, 2, 2, 0},
                                           {MODE_DOWN, 4, 1, 8, 1, 0}, {MODE_DOWN, 2, 2, 4, 2, 0}, {MODE_UP, 4, 2, 4, 2, 0}, {MODE_UP, 8, 1, 8, 1, 0}};
      bool have = false;
      for (const ConvPlan& q : kws_plans) have = have || (q.mode == p.mode && q.to == p.to && q.nwn == p.nwn && q.ks == p.ks && q.cpi == p.cpi);
      if (!have) kw = 1;
      if (kw > 1) {
        p.kws = 1;
        if (kslot >= 64) return fail(LDP_EINVAL, "more than 64 K-split convs per evaluation");
        a.kw = kw;
        a.kw_slab = P.kw_slab.as<unsigned long long>();
        a.kw_slot = kslot;
        ++kslot;
      }
    }
    if (a.kw <= 1) p.cpi = cpi_nokw;
    a.xa = xa; a.xb = xb; a.ca = ca; a.cb = cb;
    a.w = w.w.f(); a.bias = w.bias.f();
    a.bres = w.bres.f(); a.res_out = res_out;
    a.gn_scale = w.gn_scale.f(); a.gn_bias = w.gn_bias.f();
    if (film) {
      a.film_t = P.film_t.f() + film->film_off;
      a.film_g = P.film_g.f() + film->film_off;
      a.film_stride = P.F;
    }
    a.k_dev = k_dev; a.k = k;
    a.res_in = res_in; a.out = out;
    a.B = B; a.cout = w.cout_p; a.flags = flags; a.rows_valid = B * to;
    {
      // Which operand crosses the fabric once and which eight times?  Work-groups are placed on XCD
      // (linear block id % 8); each XCD's L2 fetches what its work-groups read.  Group-major (an XCD = one
      // GroupNorm group of every sample block): the layer's weights cross once, its input activations once
      // per XCD.  Sample-major (an XCD = two sample blocks, all groups): activations once, weights once per
      // XCD.  The work-groups cannot start before their first activation tile arrives, while weights stream
      // behind the MFMAs: measured per layer at 256 plans, sample-major is -1.8 us on the 256-channel T=8 convs
      // (weights 0.6 x activations), about -0.3 us up to a ratio of 5 and +0.4 us on the 1024x1024 T=2 convs
      // (ratio 10).  The default threshold (2) takes the clear wins and leaves the layers where the placement
      // only moves more bytes for the same time (thresholds 2..8 time alike, >= 11 lose 1 %).
      const double wbytes = 4.0 * w.nj * (ca + cb) * w.cout_p, abytes = 4.0 * B * mode_ti(mode, to) * (ca + cb);
      const int thr = h->opt.by_sample;
      if (thr > 0 && a.kw <= 1 && (B + 15) / 16 <= 32768 && wbytes < (double)thr * abytes) a.by_sample = 1;
    }
    return launch_conv(h, p, a, s);
  }

  // ConditionalResidualBlock1D: 2 launches
  int block(const ResBlock& b, int t, const float* xa, int ca, const float* xb, int cb, float* tmp,
            float* rbuf, float* out) {
    LDP_TRY(conv(b.c1, MODE_K5, t, xa, ca, xb, cb, tmp, EP_GN | EP_FILM, &b, nullptr,
                 b.proj ? rbuf : nullptr));
    const float* res = b.proj ? rbuf : xa;        // identity residual only when cb == 0
    LDP_TRY(conv(b.c2, MODE_K5, t, tmp, b.cout, nullptr, 0, out, EP_GN | EP_RESIN, nullptr, res, nullptr));
    return LDP_OK;
  }
};
}  // namespace

// Runs eps = unet(state, k, cond-derived FiLM) on the padded state and then either applies the
// scheduler update in place (step) and/or writes eps to eps_out (unpadded).
int planner_forward_launch(ldp_handle* h, int B, const int* k_dev, int k, bool step,
                           const StepCoef* coef, const float* noise, int step_idx, float* eps_out,
                           hipStream_t s) {
  PlannerState& P = h->pl;
  // column split only while every work-group of the grid is co-resident (2 x 8 x B/16 <= 256 CUs):
  // the two halves of a group wait for each other inside the launch
  // (safe mode -- entered after a peer ever timed out -- never splits: no in-launch dependency at all)
  const bool no_split = h->opt.no_csplit || h->safe_mode;
  const int nsb = (B + 15) / 16;
  const int ncu = h->n_cu;
  const int cs_want = no_split ? 1 : (nsb * 8 * 4 <= ncu ? 4 : (nsb * 8 * 2 <= ncu ? 2 : 1));
  // two row blocks per work-group once that still gives every CU a work-group (8 groups x B/32 >= 256)
  const bool no_mb2 = h->opt.no_mb2 != 0;
  const int mb_want = (!no_mb2 && cs_want == 1 && ((B + 31) / 32) * 8 >= ncu) ? 2 : 1;
  Fwd f{h, P, B, k_dev, k, s, step_idx, cs_want, mb_want};
  float *A = P.bufA.f(), *Bf = P.bufB.f(), *Cc = P.bufC.f(), *R = P.bufR.f();
  auto other = [&](const float* cur) { return cur == Bf ? Cc : Bf; };
  const float* x = P.state.f();
  int xc = P.DP, t = P.T, bi = 0;
  for (int l = 0; l < P.L; ++l) {
    float* o = other(x);
    LDP_TRY(f.block(P.blocks[bi++], t, x, xc, nullptr, 0, A, R, o));
    x = o; xc = P.dims[l];
    LDP_TRY(f.block(P.blocks[bi++], t, x, xc, nullptr, 0, A, R, P.skip[l].f()));
    x = P.skip[l].f();
    if (l < P.L - 1) {
      LDP_TRY(f.conv(P.down[l], MODE_DOWN, t / 2, x, xc, nullptr, 0, Bf, 0, nullptr, nullptr, nullptr));
      x = Bf; t /= 2;
    }
  }
  for (int m = 0; m < 2; ++m) {
    float* o = other(x);
    LDP_TRY(f.block(P.blocks[bi++], t, x, xc, nullptr, 0, A, R, o));
    x = o;
  }
  for (int u = 0; u < P.L - 1; ++u) {
    const float* sk = P.skip[P.L - 1 - u].f();
    float* o = other(x);
    LDP_TRY(f.block(P.blocks[bi], t, x, xc, sk, xc, A, R, o));
    x = o; xc = P.blocks[bi].cout; ++bi;
    o = other(x);
    LDP_TRY(f.block(P.blocks[bi++], t, x, xc, nullptr, 0, A, R, o));
    x = o;
    o = other(x);
    LDP_TRY(f.conv(P.up[u], MODE_UP, t * 2, x, xc, nullptr, 0, o, 0, nullptr, nullptr, nullptr));
    x = o; t *= 2;
  }
  LDP_TRY(f.conv(P.fin_block, MODE_K5, t, x, xc, nullptr, 0, A, EP_GN, nullptr, nullptr, nullptr));
  {
    // The final 1x1 conv has no tap across positions: a (B, T, C) tensor is the same memory as (B*T/2, 2, C), and the
    // epilogue's row index (sample * TO + position) is the same number in both views.  Launched over pairs of
    // positions it has T/2 times the work-groups, each staging 32 KB instead of 16 samples' whole rows (at 256 plans:
    // 64 work-groups instead of 32, 11.4 -> ~7 us).
    ConvPlan p;
    int cs = 1;
    const bool by_rows = !h->opt.no_fin_rows && t % 2 == 0;
    const int tq = by_rows ? 2 : t, Bq = by_rows ? B * (t / 2) : B;
    if (!by_rows) cs = cs_want;
    LDP_TRY(pick_plan(MODE_P1, tq, P.DP, xc, xc, false, p, &cs));
    ConvArgs a{};
    a.cs = cs;
    a.ctl = h->ctl_planner();
    a.fault = h->fault_dev;
    a.xa = A; a.ca = xc; a.w = P.fin_conv.w.f(); a.bias = P.fin_conv.bias.f();
    a.out = P.state.f(); a.B = Bq; a.cout = P.DP; a.d_real = P.D; a.rows_valid = B * t;
    a.flags = (step ? EP_STEP : 0) | (eps_out ? EP_EPSOUT : 0);
    if (coef) a.coef = *coef;
    a.noise = noise; a.seed = h->ctl_planner(); a.step = step_idx; a.eps_out = eps_out;
    a.k_dev = nullptr; a.k = k;                     // the timestep only selects FiLM rows: none in this layer
    if (by_rows && h->opt.by_sample > 0 && P.DP == p.bn()) {
      // same XCD as the 16-sample blocks of the layers around it (tconv.hpp, ConvArgs::sb_qs)
      const int q = t / 2;
      int qs = 0;
      while ((1 << qs) < q) ++qs;
      if ((1 << qs) == q && (Bq + 15) / 16 <= 32768) { a.by_sample = 1; a.sb_qs = qs; }
    }
    LDP_TRY(launch_conv(h, p, a, s));
  }
  return LDP_OK;
}

static int planner_prepare(ldp_handle* h, const float* cond, int B, hipStream_t s) {
  PlannerState& P = h->pl;
  // Exchange tags carry 20 bits of the planner call epoch (control block 0, advanced once per planner call:
  // every caller of this function follows it with exactly one set_seed_launch on ctl_planner): wipe the
  // granule / flag slabs every 2^19 calls so that no tag written 2^20 calls ago can ever be mistaken for a
  // current one (a long-running service gets there).  IDM calls use their own control block.
  ++P.calls;
  if ((P.calls & ((1ull << 19) - 1)) == 0) LDP_HIP(hipMemsetAsync(P.xchg.p, 0, P.xchg.bytes, s));
  // the K-split granules carry 12 bits of the epoch (plus step and launch slot): wiped every 2^11 calls
  if ((P.calls & ((1ull << 11) - 1)) == 0) LDP_HIP(hipMemsetAsync(P.kw_slab.p, 0, P.kw_slab.bytes, s));
  if (P.G > 0 && cond) {
    LDP_HIP(hipMemcpyAsync(P.cond.p, cond, (size_t)B * P.G * 4, hipMemcpyDeviceToDevice, s));
  }
  return LDP_OK;
}

// FiLM global-cond part, once per call: film_g = Mish(cond) @ W[E:]   (B, F)
static int planner_film_g(ldp_handle* h, int B, hipStream_t s) {
  PlannerState& P = h->pl;
  if (P.G > 0) {
    LDP_TRY(dense_launch(P.cond.f(), P.G, P.wfilm_g.f(), P.F, nullptr, P.film_g.f(), P.F, B, P.G, P.F,
                         1, 0, s));
  } else {
    LDP_HIP(hipMemsetAsync(P.film_g.p, 0, (size_t)B * P.F * 4, s));
  }
  h->last_total_launches++;
  return LDP_OK;
}

int check_sampler(int sampler, int n_steps, int n_train, const char* what) {
  // the exchange tags of the in-launch hand-offs carry step + 1 in 12 bits (statistics) / 14 bits (K split): more
  // steps would carry into the call-epoch bits and a neighbouring call's granules could validate
  if (n_steps > 4094)
    return fail(LDP_EINVAL, "%s: at most 4094 denoising steps (exchange tags carry the step in 12 bits), got %d", what, n_steps);
  if (sampler == LDP_SAMPLER_DDPM) {
    if (n_steps != n_train)
      return fail(LDP_EINVAL, "%s: DDPM visits every training timestep: n_steps must be %d (got %d)", what,
                  n_train, n_steps);
  } else if (sampler == LDP_SAMPLER_DDIM) {
    if (n_steps <= 0 || n_train % n_steps != 0)
      return fail(LDP_EINVAL, "%s: DDIM needs n_steps | %d (got %d)", what, n_train, n_steps);
  } else {
    return fail(LDP_EINVAL, "unknown sampler %d", sampler);
  }
  return LDP_OK;
}

// A split work-group that gave up on its peer stored 1 into the pinned fault word.  Every sampling
// entry point looks at it first (a plain host read): results enqueued since the previous look are
// then invalid, the handle switches to safe mode (no in-launch exchange, captured graphs dropped)
// and the call fails with LDP_EFAULT until the caller acknowledges through ldp_poll_fault.
int entry_fault_check(ldp_handle* h) {
  if (h->fault_host && *h->fault_host != 0u) {
    *h->fault_host = 0u;
    h->fault_pending = true;
    h->faults_seen++;
    if (!h->safe_mode) {
      h->safe_mode = true;
      drop_graphs(h);
    }
  }
  if (h->fault_pending)
    return fail(LDP_EFAULT, "a split work-group timed out waiting for its peer: results enqueued since the last "
                "ldp_poll_fault are invalid; the handle now runs in safe mode (no in-launch exchange) -- "
                "acknowledge with ldp_poll_fault and re-issue the call");
  return LDP_OK;
}

// inputs -> handle-owned buffers (graph nodes only ever reference handle memory), seeds, x_T
int planner_pre(ldp_handle* h, const float* cond, const float* x_init, const float* step_noise, uint64_t seed,
                int64_t row_offset, const LoopSpec& L, int B, hipStream_t s) {
  PlannerState& P = h->pl;
  const size_t per_step = (size_t)B * P.T * P.D;
  LDP_TRY(planner_prepare(h, cond, B, s));
  LDP_TRY(set_seed_launch(h->ctl_planner(), seed, row_offset * P.T, ++h->epoch_planner, s));
  if (x_init) LDP_TRY(pad_rows_launch(x_init, P.state.f(), (int64_t)B * P.T, P.D, P.DP, s));
  else LDP_TRY(philox_init_launch(P.state.f(), P.T, B, P.D, P.DP, h->ctl_planner(), s));
  if (L.explicit_noise) {
    if (per_step * L.n_steps * 4 > P.noise.bytes) drop_graphs(h);     // captured pointers move
    LDP_TRY(P.noise.alloc(per_step * L.n_steps * 4));
    LDP_HIP(hipMemcpyAsync(P.noise.p, step_noise, per_step * L.n_steps * 4, hipMemcpyDeviceToDevice, s));
  }
  return LDP_OK;
}

int planner_loop(ldp_handle* h, int B, const LoopSpec& L, hipStream_t q) {
  PlannerState& P = h->pl;
  std::vector<StepCoef> coefs;
  make_step_coefs(P.n_train, L.n_steps, L.sampler, coefs);
  const size_t per_step = (size_t)B * P.T * P.D;
  LDP_TRY(planner_film_g(h, B, q));
  for (int i = 0; i < L.n_steps; ++i) {
    const int t = (int)coefs[i].t;
    const float* nz = L.explicit_noise ? P.noise.f() + per_step * i : nullptr;
    LDP_TRY(planner_forward_launch(h, B, nullptr, t, true, &coefs[i], nz, i, nullptr, q));
  }
  return LDP_OK;
}

// Enqueue `enqueue` on s directly, or capture it once into a hipGraph (cached under `key`) and replay.
int run_or_replay(ldp_handle* h, const GraphKey& key, bool use_graph, hipStream_t s,
                  const std::function<int(hipStream_t)>& enqueue) {
  if (!use_graph) return enqueue(s);
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    LDP_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    const int r = enqueue(h->cap_stream);
    hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
    if (r != LDP_OK) { if (graph) (void)hipGraphDestroy(graph); return r; }
    if (e != hipSuccess) return fail(LDP_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(LDP_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    if (h->graph_cap > 0 && (int)h->graphs.size() >= h->graph_cap) {       // evict the least recently replayed graph
      auto lru = h->graphs.begin();
      for (auto jt = h->graphs.begin(); jt != h->graphs.end(); ++jt)
        if (jt->second.last_use < lru->second.last_use) lru = jt;
      (void)hipGraphExecDestroy(lru->second.exec);
      h->graphs.erase(lru);
      h->graphs_evicted++;
    }
    it = h->graphs.emplace(key, GraphEntry{exec, h->last_conv_launches, h->last_total_launches}).first;
    h->graphs_captured++;
  } else {
    // counters of a replay = counters of the captured loop
    h->last_conv_launches = it->second.conv_launches;
    h->last_total_launches = it->second.total_launches;
  }
  it->second.last_use = ++h->graph_clock;
  LDP_HIP(hipGraphLaunch(it->second.exec, s));
  return LDP_OK;
}

}  // namespace ldp

// =============================================================================================
// C ABI
// =============================================================================================
using namespace ldp;

extern "C" {

const char* ldp_last_error(void) { return last_error().c_str(); }
const char* ldp_version(void) { return "ldp_hip 0.1.0 (gfx950, f32 MFMA 16x16x4)"; }

int ldp_create(const ldp_config* cfg, ldp_handle** out) {
  if (!cfg || !out) return fail(LDP_EINVAL, "null argument");
  if (cfg->obs_dim <= 0 || cfg->obs_dim > 128) return fail(LDP_EINVAL, "obs_dim must be in 1..128");
  if (cfg->n_levels < 2 || cfg->n_levels > LDP_MAX_LEVELS) return fail(LDP_EINVAL, "bad n_levels");
  LDP_HIP(hipSetDevice(cfg->device));
  ldp_handle* h = new (std::nothrow) ldp_handle();
  if (!h) return fail(LDP_ENOMEM, "out of host memory");
  h->cfg = *cfg;
  {
    // the column-split kernels need their whole grid co-resident (one work-group per CU): size the
    // split by the CUs this device really has (a partitioned MI355X exposes fewer than 256)
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || ncu <= 0) {
      delete h;
      return fail(LDP_EHIP, "cannot query the compute-unit count of device %d", cfg->device);
    }
    h->n_cu = ncu;
  }
  if (hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(LDP_EHIP, "hipStreamCreate failed");
  }
  if (h->seed.alloc(64) != LDP_OK) { delete h; return LDP_ENOMEM; }
  {
    void* fh = nullptr;
    void* fd = nullptr;
    if (hipHostMalloc(&fh, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&fd, fh, 0) != hipSuccess) {
      if (fh) (void)hipHostFree(fh);
      delete h;
      return fail(LDP_EHIP, "cannot allocate the pinned fault word");
    }
    memset(fh, 0, 64);
    h->fault_host = static_cast<volatile unsigned int*>(fh);
    h->fault_dev = static_cast<unsigned int*>(fd);
  }
  {
    const int r = tconv_init_all();
    if (r != 0) { delete h; return fail(LDP_EHIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s",
                                        hipGetErrorString((hipError_t)r)); }
  }
  (void)hipMemset(h->seed.p, 0, 64);      // 2 x {seed, first global row, call epoch, -}
  *out = h;
  return LDP_OK;
}

int ldp_destroy(ldp_handle* h) {
  if (!h) return LDP_OK;
  (void)hipDeviceSynchronize();
  drop_graphs(h);
  vae_destroy(h);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  if (h->fault_host) (void)hipHostFree(const_cast<unsigned int*>(h->fault_host));
  delete h;
  return LDP_OK;
}

int ldp_set_weight(ldp_handle* h, const char* path, const float* host, const int64_t* shape,
                   int32_t ndim) {
  if (!h || !path || !host || !shape || ndim < 1 || ndim > 4) return fail(LDP_EINVAL, "bad argument");
  std::string p(path);
  if (p.rfind("planner/", 0) != 0 && p.rfind("idm/", 0) != 0 && p.rfind("vae/", 0) != 0)
    return fail(LDP_EKEY, "weight path '%s' must start with planner/, idm/ or vae/", path);
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const int64_t n = t.numel();
  if (n <= 0) return fail(LDP_EINVAL, "empty weight '%s'", path);
  t.data.assign(host, host + n);
  h->weights[p] = std::move(t);
  if (p[0] == 'p') h->pl.ready = false;
  if (p[0] == 'i') h->idm.ready = false;
  return LDP_OK;
}

int ldp_finalize(ldp_handle* h, int32_t modules, void* stream) {
  if (!h) return fail(LDP_EINVAL, "null handle");
  hipStream_t s = (hipStream_t)stream;
  LDP_HIP(hipSetDevice(h->cfg.device));
  drop_graphs(h);
  if (modules & 1) LDP_TRY(planner_finalize(h, s));
  if (modules & 2) LDP_TRY(idm_finalize(h, s));
  if (modules & 4) LDP_TRY(vae_finalize(h, s));
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}

int ldp_unet_forward(ldp_handle* h, const float* x, const int32_t* k_dev, int32_t k,
                     const float* cond, float* eps, int32_t B, void* stream) {
  if (!h || !x || !eps || B <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->pl.ready) return fail(LDP_ESTATE, "planner weights not finalized");
  PlannerState& P = h->pl;
  if (P.G > 0 && !cond) return fail(LDP_EINVAL, "cond is required (global_cond_dim=%d)", P.G);
  if (!k_dev && (k < 0 || k >= P.n_train)) return fail(LDP_EINVAL, "timestep %d out of range", k);
  hipStream_t s = (hipStream_t)stream;
  LDP_TRY(entry_fault_check(h));
  LDP_TRY(planner_workspace(h, B));
  h->last_conv_launches = h->last_total_launches = 0;
  LDP_TRY(planner_prepare(h, cond, B, s));
  LDP_TRY(set_seed_launch(h->ctl_planner(), 0, 0, ++h->epoch_planner, s));      // advances the call epoch
  LDP_TRY(pad_rows_launch(x, P.state.f(), (int64_t)B * P.T, P.D, P.DP, s));
  LDP_TRY(planner_film_g(h, B, s));
  return planner_forward_launch(h, B, k_dev, k, false, nullptr, nullptr, 0, eps, s);
}

// The time of a sampling loop is a staircase in the batch: the launch plans are built around 16 / 32 / 64 / 128 sample
// blocks of 16 plans on 256 CUs, and a batch just above one of those sizes costs as much as the next one (576 plans:
// 143 ms, the time of 1024; 1280 plans: 251 ms, nearly the time of 2048 -- profiles/r03_batch_staircase.txt).  Rows are
// independent and every random draw is keyed by the global row, so such a batch runs as two loops: its leading
// power-of-two part and the rest (576 = 512 + 64: 107 ms).  Worth it when the rest is at most half the leading part and
// the batch is beyond 512 plans (below, the rest's own launch-bound loop costs more than the step; beyond 2048 plans:
// at most a quarter).  Results equal the
// single loop's to fp32 round-off (the two parts may use different launch regimes), like any re-sharding of the rows.
// Calls with explicit per-step noise are never split (their noise tensor is laid out by step, not by row).
static int batch_split(const ldp_handle* h, int B) {
  if (h->opt.no_batch_split) return 0;
  const int nsb = (B + 15) / 16;
  if (nsb <= 32 || nsb > 256) return 0;
  int p = 32;
  while (p * 2 <= nsb) p *= 2;
  const int rem = nsb - p;
  // beyond 2048 plans the loops run several rounds of work-groups anyway and the step is smaller: a quarter pays, a half does not
  return (rem > 0 && rem <= (p >= 128 ? p / 4 : p / 2)) ? p * 16 : 0;
}

int ldp_plan_sample(ldp_handle* h, const float* cond, const float* x_init, const float* step_noise,
                    uint64_t seed, int64_t row_offset, int32_t sampler, int32_t n_steps, float* out,
                    int32_t B, int32_t use_graph, void* stream) {
  if (!h || !out || B <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->pl.ready) return fail(LDP_ESTATE, "planner weights not finalized");
  PlannerState& P = h->pl;
  if (P.G > 0 && !cond) return fail(LDP_EINVAL, "cond is required (global_cond_dim=%d)", P.G);
  if (!step_noise) {
    if (const int b1 = batch_split(h, B)) {
      const size_t row = (size_t)P.T * P.D;
      LDP_TRY(ldp_plan_sample(h, cond, x_init, nullptr, seed, row_offset, sampler, n_steps, out, b1, use_graph, stream));
      return ldp_plan_sample(h, cond ? cond + (size_t)b1 * P.G : nullptr, x_init ? x_init + b1 * row : nullptr, nullptr, seed,
                             row_offset + b1, sampler, n_steps, out + b1 * row, B - b1, use_graph, stream);
    }
  }
  LDP_TRY(check_sampler(sampler, n_steps, P.n_train, "planner"));
  LDP_TRY(entry_fault_check(h));
  hipStream_t s = (hipStream_t)stream;
  LDP_TRY(planner_workspace(h, B));
  h->last_conv_launches = h->last_total_launches = 0;
  LoopSpec L{n_steps, sampler, step_noise != nullptr && sampler == LDP_SAMPLER_DDPM};
  LDP_TRY(planner_pre(h, cond, x_init, step_noise, seed, row_offset, L, B, s));
  const int Bg = bucket_rows(B, L.explicit_noise);
  GraphKey key{0, Bg, n_steps, sampler, L.explicit_noise ? 1 : 0};
  LDP_TRY(run_or_replay(h, key, use_graph != 0, s, [&](hipStream_t q) { return planner_loop(h, Bg, L, q); }));
  LDP_TRY(unpad_rows_launch(P.state.f(), out, (int64_t)B * P.T, P.D, P.DP, s));
  return LDP_OK;
}

// sample_viz_step without the decode (agent/ldp_agent.py:452-505) as ONE captured graph:
// planner loop -> plan assembly + transitions -> IDM loop.
int ldp_agent_sample(ldp_handle* h, const float* obs_emb, int32_t obs_frames, int32_t obs_horizon,
                     const float* x_init, const float* x_noise, const float* a_init, const float* a_noise,
                     uint64_t seed, int64_t row_offset, int32_t sampler, int32_t planner_steps,
                     int32_t idm_steps, float* x_out, float* plan_out, float* action_out,
                     const float* act_lo, const float* act_hi, int32_t act_dim, int32_t act_mode,
                     int32_t B, int32_t use_graph, void* stream) {
  if (!h || !obs_emb || !plan_out || !action_out || B <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->pl.ready) return fail(LDP_ESTATE, "planner weights not finalized");
  if (!h->idm.ready) return fail(LDP_ESTATE, "idm weights not finalized");
  PlannerState& P = h->pl;
  IdmState& I = h->idm;
  const int ah = h->cfg.action_horizon, D = P.D, A = I.A;
  if (obs_horizon <= 0 || obs_frames < obs_horizon || obs_horizon * D != P.G)
    return fail(LDP_EINVAL, "obs_horizon %d x obs_dim %d does not match global_cond_dim %d (frames given: %d)",
                obs_horizon, D, P.G, obs_frames);
  if (ah < 1 || ah > P.T) return fail(LDP_EINVAL, "action_horizon %d must be in 1..pred_horizon %d", ah, P.T);
  if (act_dim != 0 && (act_dim != 1 && act_dim != A)) return fail(LDP_EINVAL, "action bounds of length %d (A = %d)", act_dim, A);
  if (act_dim != 0 && (!act_lo || !act_hi)) return fail(LDP_EINVAL, "action bounds missing");
  if (!x_noise && !a_noise) {
    if (const int b1 = batch_split(h, B)) {       // the leading power-of-two part and the rest as two calls (batch_split())
      const size_t xr = (size_t)P.T * D, pr = (size_t)(ah + 1) * D, ar = (size_t)ah * A;
      LDP_TRY(ldp_agent_sample(h, obs_emb, obs_frames, obs_horizon, x_init, nullptr, a_init, nullptr, seed, row_offset, sampler,
                               planner_steps, idm_steps, x_out, plan_out, action_out, act_lo, act_hi, act_dim, act_mode, b1,
                               use_graph, stream));
      return ldp_agent_sample(h, obs_emb + (size_t)b1 * obs_frames * D, obs_frames, obs_horizon, x_init ? x_init + b1 * xr : nullptr,
                              nullptr, a_init ? a_init + b1 * ar : nullptr, nullptr, seed, row_offset + b1, sampler, planner_steps,
                              idm_steps, x_out ? x_out + b1 * xr : nullptr, plan_out + b1 * pr, action_out + b1 * ar, act_lo, act_hi,
                              act_dim, act_mode, B - b1, use_graph, stream);
    }
  }
  LDP_TRY(check_sampler(sampler, planner_steps, P.n_train, "planner"));
  LDP_TRY(check_sampler(sampler, idm_steps, I.n_train, "idm"));
  LDP_TRY(entry_fault_check(h));
  hipStream_t s = (hipStream_t)stream;
  const int R = B * ah;
  LDP_TRY(planner_workspace(h, B));
  LDP_TRY(idm_workspace(h, (B + 15) / 16 * 16 * ah));       // the loops run over whole 16-plan tiles (bucket_rows)
  if ((size_t)B * (ah + 1) * D * 4 > h->plan_out.bytes || (size_t)B * D * 4 > h->obs_last.bytes) {
    drop_graphs(h);
    LDP_TRY(h->plan_out.alloc((size_t)((B + 15) / 16 * 16) * (ah + 1) * D * 4));
    LDP_TRY(h->obs_last.alloc((size_t)((B + 15) / 16 * 16) * D * 4));
  }
  h->last_conv_launches = h->last_total_launches = 0;
  LoopSpec LP{planner_steps, sampler, x_noise != nullptr && sampler == LDP_SAMPLER_DDPM};
  LoopSpec LI{idm_steps, sampler, a_noise != nullptr && sampler == LDP_SAMPLER_DDPM};
  // eager prologue on the caller's stream: inputs -> handle memory, both control blocks, x_T and a_T
  LDP_TRY(gather_obs_launch(obs_emb, P.cond.f(), h->obs_last.f(), B, obs_frames, D, obs_horizon, s));
  LDP_TRY(planner_pre(h, nullptr, x_init, x_noise, seed, row_offset, LP, B, s));
  LDP_TRY(idm_pre(h, nullptr, a_init, a_noise, seed, row_offset * ah, LI, R, s));
  const int Bg = bucket_rows(B, LP.explicit_noise || LI.explicit_noise);
  GraphKey key{2, Bg, planner_steps, sampler, LP.explicit_noise ? 1 : 0, idm_steps, LI.explicit_noise ? 1 : 0};
  LDP_TRY(run_or_replay(h, key, use_graph != 0, s, [&](hipStream_t q) -> int {
    LDP_TRY(planner_loop(h, Bg, LP, q));
    LDP_TRY(assemble_plan_launch(P.state.f(), h->obs_last.f(), h->plan_out.f(), I.trans.f(), nullptr, Bg, P.T, D,
                                 P.DP, ah, q));
    h->last_total_launches++;
    return idm_loop(h, Bg * ah, LI, q);
  }));
  // epilogue: results -> caller memory (x unpadded, plan, actions un-normalised as utils/data_utils.py:12-15,61-65)
  if (x_out) LDP_TRY(unpad_rows_launch(P.state.f(), x_out, (int64_t)B * P.T, D, P.DP, s));
  LDP_HIP(hipMemcpyAsync(plan_out, h->plan_out.p, (size_t)B * (ah + 1) * D * 4, hipMemcpyDeviceToDevice, s));
  LDP_TRY(unpad_rows_launch(idm_result(h), action_out, R, A, I.AP, s));
  if (act_dim != 0) LDP_TRY(normalize_launch(action_out, action_out, (int64_t)R * A, act_lo, act_hi, act_dim, act_mode, s));
  return LDP_OK;
}

int ldp_normalize_bounds(const float* x, float* y, int64_t n, const float* lo, const float* hi,
                         int32_t dim, int32_t normalize, void* stream) {
  if (!x || !y || !lo || !hi || dim <= 0) return fail(LDP_EINVAL, "bad argument");
  return normalize_launch(x, y, n, lo, hi, dim, normalize, (hipStream_t)stream);
}

int ldp_mean_sq_diff(const float* a, const float* b, int64_t n, float* out, void* stream) {
  if (!a || !b || !out || n <= 0) return fail(LDP_EINVAL, "bad argument");
  return mean_sq_diff_launch(a, b, n, out, (hipStream_t)stream);
}

int ldp_check_fault(ldp_handle* h, void* stream) {
  if (!h) return fail(LDP_EINVAL, "null handle");
  LDP_HIP(hipStreamSynchronize((hipStream_t)stream));
  int32_t f = 0;
  LDP_TRY(ldp_poll_fault(h, &f));
  if (f)
    return fail(LDP_EFAULT, "a split work-group timed out waiting for its peer: the results enqueued since the "
                "previous check are invalid (the handle now runs in safe mode; re-issue the calls)");
  return LDP_OK;
}

int ldp_poll_fault(ldp_handle* h, int32_t* faulted) {
  if (!h || !faulted) return fail(LDP_EINVAL, "bad argument");
  bool f = h->fault_pending;
  if (h->fault_host && *h->fault_host != 0u) {
    *h->fault_host = 0u;
    h->faults_seen++;
    f = true;
  }
  if (f && !h->safe_mode) {
    h->safe_mode = true;
    drop_graphs(h);
  }
  h->fault_pending = false;
  *faulted = f ? 1 : 0;
  return LDP_OK;
}

int ldp_set_option(ldp_handle* h, const char* name, int64_t value) {
  if (!h || !name) return fail(LDP_EINVAL, "bad argument");
  const std::string n(name);
  Options& o = h->opt;
  const int v = (int)value;
  if (n == "no_csplit") o.no_csplit = v;
  else if (n == "no_mb2") o.no_mb2 = v;
  else if (n == "no_kw") o.no_kw = v;
  else if (n == "kw_min_it") o.kw_min_it = v;
  else if (n == "kw_bmax") o.kw_bmax = v;
  else if (n == "no_fin_rows") o.no_fin_rows = v;
  else if (n == "no_batch_split") o.no_batch_split = v;
  else if (n == "up_full_depth") o.up_full_depth = v;
  else if (n == "vae_w8") o.vae_w8 = v;
  else if (n == "vae_split") o.vae_split = v;
  else if (n == "vae_split_dual") o.vae_split_dual = v;
  else if (n == "graph_cap") { h->graph_cap = v; return LDP_OK; }
  else if (n == "timeline_ptr") o.timeline_ptr = value;
  else if (n == "idm_unfused") o.idm_unfused = v;
  else if (n == "idm_rt_major") o.idm_rt_major = v;
  else if (n == "idm_noring") o.idm_noring = v;
  else if (n == "idm_stream") o.idm_stream = v;
  else if (n == "by_sample") o.by_sample = v;
  else if (n == "idm_hs") { if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) return fail(LDP_EINVAL, "idm_hs must be 0, 1, 2, 4 or 8"); o.idm_hs = v; }
  else if (n == "idm_rows32") o.idm_rows32 = v;
  else if (n == "idm_rows32_min") o.idm_rows32_min = v;
  else if (n == "idm_hs32") { if (v != 0 && v != 2 && v != 4 && v != 8) return fail(LDP_EINVAL, "idm_hs32 must be 0, 2, 4 or 8"); o.idm_hs32 = v; }
  else if (n == "idm_rt_major32") o.idm_rt_major32 = v != 0;
  else if (n == "dbg") o.dbg = v;
  else if (n == "repeat") o.repeat = v < 1 ? 1 : v;
  else if (n == "safe_mode") h->safe_mode = v != 0;
  else if (n == "inject_fault") { if (v && h->fault_host) *h->fault_host = 1u; return LDP_OK; }   // test hook
  else return fail(LDP_EKEY, "unknown option '%s'", name);
  drop_graphs(h);                                   // captured graphs bake the launch plan
  return LDP_OK;
}

int ldp_get_option(ldp_handle* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return fail(LDP_EINVAL, "bad argument");
  const std::string n(name);
  const Options& o = h->opt;
  if (n == "no_csplit") *value = o.no_csplit;
  else if (n == "no_mb2") *value = o.no_mb2;
  else if (n == "no_kw") *value = o.no_kw;
  else if (n == "kw_min_it") *value = o.kw_min_it;
  else if (n == "kw_bmax") *value = o.kw_bmax;
  else if (n == "no_fin_rows") *value = o.no_fin_rows;
  else if (n == "no_batch_split") *value = o.no_batch_split;
  else if (n == "up_full_depth") *value = o.up_full_depth;
  else if (n == "vae_w8") *value = o.vae_w8;
  else if (n == "vae_split") *value = o.vae_split;
  else if (n == "vae_split_dual") *value = o.vae_split_dual;
  else if (n == "timeline_ptr") *value = o.timeline_ptr;
  else if (n == "idm_unfused") *value = o.idm_unfused;
  else if (n == "idm_rt_major") *value = o.idm_rt_major;
  else if (n == "idm_noring") *value = o.idm_noring;
  else if (n == "idm_stream") *value = o.idm_stream;
  else if (n == "by_sample") *value = o.by_sample;
  else if (n == "idm_hs") *value = o.idm_hs;
  else if (n == "idm_rows32") *value = o.idm_rows32;
  else if (n == "idm_rows32_min") *value = o.idm_rows32_min;
  else if (n == "idm_hs32") *value = o.idm_hs32;
  else if (n == "idm_rt_major32") *value = o.idm_rt_major32;
  else if (n == "dbg") *value = o.dbg;
  else if (n == "repeat") *value = o.repeat;
  else if (n == "safe_mode") *value = h->safe_mode ? 1 : 0;
  else if (n == "any_debug") *value = o.any_debug() ? 1 : 0;
  else if (n == "faults_seen") *value = h->faults_seen;
  else if (n == "n_cu") *value = h->n_cu;
  else if (n == "graphs") *value = (int64_t)h->graphs.size();
  else if (n == "graph_cap") *value = h->graph_cap;
  else if (n == "graphs_captured") *value = h->graphs_captured;
  else if (n == "graphs_evicted") *value = h->graphs_evicted;
  else return fail(LDP_EKEY, "unknown option '%s'", name);
  return LDP_OK;
}

int ldp_philox_raw(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, uint32_t* out_dev,
                   int64_t n, void* stream) {
  if (!out_dev || n <= 0) return fail(LDP_EINVAL, "bad argument");
  return philox_raw_launch(seed, elem0, step, stream_id, out_dev, n, (hipStream_t)stream);
}

int ldp_philox_normal(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id, float* out_dev,
                      int64_t n, void* stream) {
  if (!out_dev || n <= 0) return fail(LDP_EINVAL, "bad argument");
  return philox_normal_launch(seed, elem0, step, stream_id, out_dev, n, (hipStream_t)stream);
}

int ldp_launch_count(ldp_handle* h, int32_t which, int64_t* launches) {
  if (!h || !launches) return fail(LDP_EINVAL, "bad argument");
  *launches = which == 0 ? h->last_conv_launches : h->last_total_launches;
  return LDP_OK;
}

// ---- unit-testable primitives -----------------------------------------------------------------
static int prim_conv(int mode, const float* x, const float* kernel_host, const float* bias_host,
                     const float* gs, const float* gb, const float* film, float* y, int B, int t_in,
                     int cin, int cout, void* stream) {
  if (!x || !kernel_host || !bias_host || !y || B <= 0) return fail(LDP_EINVAL, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int nj = mode_taps(mode);
  const int to = mode == MODE_DOWN ? t_in / 2 : mode == MODE_UP ? t_in * 2 : t_in;
  if (cout % 128 != 0) return fail(LDP_EINVAL, "Cout must be a multiple of 128 (8 groups x 16)");
  const int cin_p = round_up(cin, 32);
  ConvPlan p;
  LDP_TRY(pick_plan(mode, to, cout, cin_p, cin_p, false, p));
  const int cin_pp = round_up(cin_p, p.chunk());
  if (cin_pp != cin_p) LDP_TRY(pick_plan(mode, to, cout, cin_pp, cin_pp, false, p));
  std::vector<float> packed = pack_conv(kernel_host, nj, cin, cout, cin_pp, cout);
  DevBuf dw, db, dgs, dgb, dx, dfg, dft;
  LDP_TRY(upload(dw, packed.data(), packed.size() * 4, s));
  LDP_TRY(upload(db, bias_host, (size_t)cout * 4, s));
  ConvArgs a{};
  if (gs) {
    LDP_TRY(upload(dgs, gs, (size_t)cout * 4, s));
    LDP_TRY(upload(dgb, gb, (size_t)cout * 4, s));
    a.flags |= EP_GN;
  }
  const float* xin = x;
  if (cin_pp != cin) {                                  // zero-pad channels
    LDP_TRY(dx.alloc((size_t)B * t_in * cin_pp * 4));
    LDP_TRY(pad_rows_launch(x, dx.f(), (int64_t)B * t_in, cin, cin_pp, s));
    xin = dx.f();
  }
  if (film) {                                            // film (B, 2*Cout): use as film_g, zero film_t
    LDP_TRY(dft.alloc((size_t)2 * cout * 4));
    LDP_HIP(hipMemsetAsync(dft.p, 0, (size_t)2 * cout * 4, s));
    a.film_t = dft.f(); a.film_g = film; a.film_stride = 2 * cout; a.flags |= EP_FILM;
  }
  a.xa = xin; a.ca = cin_pp; a.w = dw.f(); a.bias = db.f();
  a.gn_scale = dgs.f(); a.gn_bias = dgb.f();
  a.out = y; a.B = B; a.cout = cout; a.k = 0; a.rows_valid = B * to;
  const int r = tconv_launch(p, a, s);
  if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "primitive conv launch failed (%d)", r);
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}

int ldp_conv1d_gn_mish_film_f32(const float* x, const float* kernel_host, const float* bias_host,
                                const float* gn_scale_host, const float* gn_bias_host,
                                const float* film, float* y, int32_t B, int32_t T, int32_t Cin,
                                int32_t Cout, void* stream) {
  if (!gn_scale_host || !gn_bias_host) return fail(LDP_EINVAL, "GroupNorm parameters required");
  return prim_conv(MODE_K5, x, kernel_host, bias_host, gn_scale_host, gn_bias_host, film, y, B, T, Cin,
                   Cout, stream);
}

int ldp_downsample1d_f32(const float* x, const float* kernel_host, const float* bias_host, float* y,
                         int32_t B, int32_t T, int32_t C, void* stream) {
  if (T % 2) return fail(LDP_EINVAL, "T must be even");
  return prim_conv(MODE_DOWN, x, kernel_host, bias_host, nullptr, nullptr, nullptr, y, B, T, C, C, stream);
}

int ldp_upsample1d_f32(const float* x, const float* kernel_host, const float* bias_host, float* y,
                       int32_t B, int32_t T, int32_t C, void* stream) {
  return prim_conv(MODE_UP, x, kernel_host, bias_host, nullptr, nullptr, nullptr, y, B, T, C, C, stream);
}

}  // extern "C"
