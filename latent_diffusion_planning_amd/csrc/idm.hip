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

// =============================================================================================
// Fused MLPResNetBlock (networks/mlp_diffusion_nets.py:8-30):  h += Dense_1(relu(Dense_0(LayerNorm(h))))
// as ONE launch per block, 3 launches per denoising step (+1 at the end of the loop).
//
// Work split: a work-group owns 16 rows x one slice of the 4H = 1024 hidden units (HS slices, so that
// R/16 * HS work-groups fill the chip: HS = 4 at R = 1024).  It computes LayerNorm(h) for its rows,
// z = relu(LN(h) @ W0[:, slice] + b0[slice]) into LDS (never to HBM), then its K-partial
// z @ W1[slice, :] (16 x 256) which it writes to part[slice].  The partials are NOT reduced inside
// the launch (no cross-work-group wait, no co-residency requirement): the NEXT launch's prologue adds
// them in slice order -- every work-group of a row tile redundantly, 64 KB of L2 reads -- together
// with b1 and the residual.  The first block's prologue also carries the tail of the previous
// denoising step: eps = relu(h) @ W_out + b_out (A <= 32 outputs: VALU dot products + DPP wave sums),
// the DDPM/DDIM update of a, and the input Dense  h0 = a @ W[:A] + spart + ctab[k]  (idm.hip header).
// MFMA mapping as in tconv.hpp: rows = 16 samples, v_mfma_f32_16x16x4_f32, B fragments streamed
// global -> VGPR from the same packed layout (pack_conv, one tap), A fragments from swizzled LDS tiles.
// =============================================================================================
#pragma clang fp contract(off)

enum : int { IF_BLOCK = 1, IF_IN = 2, IF_RED = 4, IF_TAIL = 8, IF_STEP = 16, IF_EPSOUT = 32 };

struct IdmFusedArgs {
  // result of the previous launch's block (IF_RED / IF_TAIL): h = hprev + ((sum_j part_prev[j]) + b1_prev)
  const float* hprev;       // (Rp, H)
  const float* part_prev;   // (hs_prev, Rp, H)
  const float* b1_prev;     // (H)
  int hs_prev;
  // IF_TAIL: eps = relu(h) @ W_out + b_out, then the scheduler update (IF_STEP) and/or eps output
  const float* wout_t;      // (A, H): MLPResNet_0/Dense_1 kernel, transposed
  const float* bout;        // (A)
  const float* state_in;    // (Rp, AP): a_t
  float* state_out;         // (Rp, AP): a_{t-1}, written by slice 0
  StepCoef coef;
  const float* noise;       // (R, A) explicit N(0,1) of this step or nullptr -> Philox
  const uint64_t* ctl;      // IDM control block {seed, first global row, ...}
  int step;
  float* eps_out;           // (R, A)
  // IF_IN: h = a @ Wa + spart + ctab[k]
  const float* wa;          // (A, H)
  const float* spart;       // (Rp, H)
  const float* ctab;        // (n_train, H)
  const int* k_dev;
  int k;
  // IF_BLOCK
  float* hcur;              // (Rp, H): the block's input, kept for the residual; written by slice 0
  const float* ln_s;
  const float* ln_b;
  const float* w0;          // packed [H/16][4H/16][64][4]
  const float* b0;          // (4H)
  const float* w1;          // packed [4H/16][H/16][64][4]
  float* part_out;          // (HS, Rp, H)
  int R, Rp, A, AP, flags;
  int rt_major;
  int stream_parts;         // K-partials: 0 plain loads/stores (they stay in the XCD's L2 for the next launch), 1 non-temporal chosen at run
                            // time, 2 the instantiation that only has the non-temporal path (>= 2048 rows: 3 MB per launch and XCD
                            // must not evict the weights, and the run-time choice costs 1.5 % there)
  int dbg;                  // timing ablations (tools/): 256 no partial loads, 512 no Dense_0, 1024 no Dense_1, 2048 no partial stores
  unsigned int* fault;      // fp16-plane kernel: [1] set to 1 when an operand left the planes' range (tconv.hpp ConvArgs::fault)
};

// Streaming (non-temporal) accesses for the K-partials: they are written once and read once per slice by the
// next launch -- 3 MB per launch and XCD at R = 1024, against 0.5 MB of weights that the same L2 should keep
// from one denoising step to the next.
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}

// acc[c] += A(16 x 16*NCH, LDS tile) @ B(chunks ch0.., column blocks cb0..cb0+NCB-1 of ncb_total)
// FR = VGPRs a wave may spend on weight fragments in flight.  Every work-group of an XCD asks for the same
// lines at the same time, so what hides the L2-miss latency is this depth alone.
template <int NCB, int NCH, int FR>
__device__ __forceinline__ void idm_gemm(const float* __restrict__ tile, const float* __restrict__ wpk,
                                         int ncb_total, int ch0, int cb0, f32x4 (&acc)[NCB], int lane) {
  constexpr int PF0 = FR / (4 * NCB) < 1 ? 1 : FR / (4 * NCB);
  constexpr int PF = PF0 > NCH ? NCH : PF0;
  static_assert(NCH % PF == 0, "chunk count must be a multiple of the prefetch depth");
  const int r = lane & 15, kq = lane >> 4;
  f32x4 wb[PF][NCB];
  auto wload = [&](int ch, f32x4 (&b)[NCB]) {
#pragma unroll
    for (int c = 0; c < NCB; ++c)
      b[c] = *reinterpret_cast<const f32x4*>(wpk + ((size_t)(ch0 + ch) * ncb_total + cb0 + c) * 256 + lane * 4);
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) wload(p, wb[p]);
  for (int ch = 0; ch < NCH; ch += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int cur = ch + p;
      const f32x4 av = *reinterpret_cast<const f32x4*>(tile + (cur * 16 + r) * 16 + swz(r, kq) * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wb[p][c][s], acc[c], 0, 0, 0);
      const int nx = cur + PF < NCH ? cur + PF : NCH - 1;      // the tail harmlessly re-requests the last chunk
      wload(nx, wb[p]);
    }
  }
}

#if LDP_KERNARG_PRELOAD
#define IDM_KERNEL_PARAMS const float* h_w0, const float* h_w1, const float* h_part_prev, const float* h_hprev, \
                          const float* h_state_in, int h_R, int h_Rp, int h_flags, int h_pk, const IdmFusedArgs a_in
#define IDM_KERNEL_ARGS(a) (a).w0, (a).w1, (a).part_prev, (a).hprev, (a).state_in, (a).R, (a).Rp, (a).flags, \
                           (((a).AP & 255) | (((a).rt_major & 1) << 8) | (((a).stream_parts & 3) << 9) | ((a).dbg << 12)), (a)
#else
#define IDM_KERNEL_PARAMS const IdmFusedArgs a
#define IDM_KERNEL_ARGS(a) (a)
#endif

// RINGED: the register-hungry variant (one work-group per CU) for launches that give every work-group its own CU
template <int HS, bool RINGED, bool STREAM>
__global__ __launch_bounds__(512) void idm_block_kernel(IDM_KERNEL_PARAMS) {
#if LDP_KERNARG_PRELOAD
  // the operands of the weight ring and of the prologue's loads arrive in SGPRs with the wave launch (tconv.hpp)
  IdmFusedArgs a = a_in;
  a.w0 = h_w0; a.w1 = h_w1; a.part_prev = h_part_prev; a.hprev = h_hprev; a.state_in = h_state_in;
  a.R = h_R; a.Rp = h_Rp; a.flags = h_flags;
  a.AP = h_pk & 255; a.rt_major = (h_pk >> 8) & 1; a.stream_parts = (h_pk >> 9) & 3; a.dbg = (unsigned)h_pk >> 12;
#endif
  constexpr int H = 256, HID = 4 * H, HSW = HID / HS;
  constexpr int NCH1 = H / 16, NCB1 = HSW / 16 / 8, NCH2 = HSW / 16, NCB2 = 2;
  // split launches (few rows: one work-group per CU, latency-bound) prefetch deep; the unsplit ones run two
  // work-groups per CU and must stay under 128 VGPRs
  constexpr int FR = HS >= 4 ? 128 : 64;
  static_assert(NCB1 >= 1, "at most 8 hidden slices");
  extern __shared__ f32x4 smem4[];
  float* tA = reinterpret_cast<float*>(smem4);        // LayerNorm(h): 16 chunks of 16 x 16, swizzled
  float* tZ = tA + NCH1 * 256;                        // relu(Dense_0): NCH2 chunks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x = row tile: the HS work-groups of a row tile land on one XCD (linear id % 8), so the partial sums they
  // exchange from launch to launch stay XCD-local; a.rt_major == 0 keeps the slice-major order (an XCD pair per slice:
  // each XCD streams 1/HS of the weights)
  const int j = a.rt_major ? blockIdx.y : blockIdx.x, r0 = (a.rt_major ? blockIdx.x : blockIdx.y) * 16;
  const int flags = a.flags;
  const int ecol = lane & 15, erow0 = (lane >> 4) * 4;
  const int cb0 = wave * NCB1;                          // Dense_0 column blocks (of this slice) owned by this wave
  const int ob0 = wave * NCB2;                          // Dense_1 output column blocks owned by this wave

  // Launches with one work-group per CU (every stall exposed; with the row-tile-major placement an XCD streams
  // all slices' weights from the Infinity Cache, not from its L2) run BOTH weight matrices through one
  // ring of RING fragments, fully unrolled: fragment f of the concatenated list [W0 slice: NF1 | W1 slice: NF2]
  // lives in ring[f % RING] and is re-filled with fragment f + RING right after its four MFMAs.  The first RING
  // fragments are requested before the prologue -- weights do not depend on the previous launch -- and Dense_1's
  // weights arrive while Dense_0 is still multiplying.
  constexpr int NF1 = NCH1 * NCB1, NF2 = NCH2 * NCB2, NF = NF1 + NF2;
  constexpr int RING = RINGED ? (NF1 < 32 ? NF1 : 32) : 1;
  f32x4 ring[RING];
  auto fload = [&](int f) -> f32x4 {
    if (f < NF1) {
      const int ch = f / NCB1, c = f % NCB1;
      return *reinterpret_cast<const f32x4*>(a.w0 + ((size_t)ch * (HID / 16) + j * (HSW / 16) + cb0 + c) * 256 + lane * 4);
    }
    const int g = f - NF1, ch = g / NCB2, c = g % NCB2;
    return *reinterpret_cast<const f32x4*>(a.w1 + ((size_t)(j * NCH2 + ch) * (H / 16) + ob0 + c) * 256 + lane * 4);
  };
  if (RINGED && (flags & IF_BLOCK)) {
#pragma unroll
    for (int f = 0; f < RING; ++f) ring[f] = fload(f);
    __builtin_amdgcn_sched_barrier(0);                  // keep these requests ahead of the prologue (the prologue's own operands
  }                                                     // first, the ring right behind them: measured 2.4 % slower at 1024 rows)

  // ---- prologue: this wave's two rows, four columns per lane ------------------------------------
  // every global operand of both rows is requested first (one exposed memory latency, not one per row)
  int rowq[2], rowcq[2];
  f32x4 pv[2][HS], hp[2];
  float av[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    rowq[q] = r0 + 2 * wave + q;
    rowcq[q] = rowq[q] < a.R ? rowq[q] : a.R - 1;      // clamped for loads; nothing of a dead row is stored
    if (flags & (IF_RED | IF_TAIL)) {
#pragma unroll
      for (int jj = 0; jj < HS; ++jj)                   // the previous launch used the same split (a.hs_prev == HS)
        pv[q][jj] = LDP_ABL(256) ? f32x4{0.f, 0.f, 0.f, 0.f}
                  : (STREAM || a.stream_parts) ? ld_stream(a.part_prev + ((size_t)jj * a.Rp + rowcq[q]) * H + 4 * lane)
                                   : *reinterpret_cast<const f32x4*>(a.part_prev + ((size_t)jj * a.Rp + rowcq[q]) * H + 4 * lane);
      hp[q] = *reinterpret_cast<const f32x4*>(a.hprev + (size_t)rowcq[q] * H + 4 * lane);
    }
    av[q] = 0.0f;                                       // lane i < A: a[row][i]
    if (lane < a.AP) av[q] = a.state_in[(size_t)rowcq[q] * a.AP + lane];
  }
  // Non-temporal loads were seen to be overtaken by later plain loads (and vice versa) on MI355X, while the compiler's
  // partial s_waitcnt vmcnt(N) bookkeeping assumes loads return in issue order: everything requested above has landed
  // before any of it is used (they are all needed right away anyway).
  if (STREAM || a.stream_parts) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 b1v = f32x4{0.f, 0.f, 0.f, 0.f}, ls = b1v, lb = b1v;
  if (flags & (IF_RED | IF_TAIL)) b1v = *reinterpret_cast<const f32x4*>(a.b1_prev + 4 * lane);
  if (flags & IF_BLOCK) {
    ls = *reinterpret_cast<const f32x4*>(a.ln_s + 4 * lane);
    lb = *reinterpret_cast<const f32x4*>(a.ln_b + 4 * lane);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rr = 2 * wave + q;
    const int row = rowq[q], rowc = rowcq[q];
    const bool live = row < a.R;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (flags & (IF_RED | IF_TAIL)) {
      f32x4 acc = pv[q][0];
#pragma unroll
      for (int jj = 1; jj < HS; ++jj) acc = acc + pv[q][jj];
      acc = acc + b1v;
      v = hp[q] + acc;
    }
    float aval = av[q];
    if (flags & IF_TAIL) {
      // eps = relu(h) @ W_out + b_out: one dot product of 256 per output -- 4 columns per lane, then the wave
      // sums of 8 outputs at a time (independent DPP chains interleave; one at a time is pure latency)
      const f32x4 hl = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      float my_eps = 0.0f;
      for (int a0 = 0; a0 < a.A; a0 += 8) {
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ai = (a0 + u) < a.A ? a0 + u : a.A - 1;
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.wout_t + (size_t)ai * H + 4 * lane);
          float t = hl[0] * w4[0];
          t = fmaf(hl[1], w4[1], t);
          t = fmaf(hl[2], w4[2], t);
          t = fmaf(hl[3], w4[3], t);
          p[u] = t;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0xB1, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x4E, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x114, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x118, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x142, 0xA>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x143, 0xC>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[u]), 63));
          if (lane == a0 + u) my_eps = tot;
        }
      }
      if (lane < a.A) {
        const float y = my_eps + a.bout[lane];
        if ((flags & IF_EPSOUT) && live) a.eps_out[(size_t)row * a.A + lane] = y;
        if (flags & IF_STEP) {
          const float xt = aval;
          float z = 0.f;
          if (a.coef.sigma != 0.f) {
            if (a.noise) z = a.noise[(size_t)rowc * a.A + lane];
            else z = philox_normal(a.ctl[0], (a.ctl[1] + (uint64_t)row) * (uint64_t)a.AP + (uint64_t)lane,
                                   (uint32_t)a.step, 0u);
          }
          float x0 = (xt - a.coef.sqrt_1mab * y) * a.coef.inv_sqrt_ab;
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          aval = a.coef.c_x0 * x0 + a.coef.c_x * xt + a.coef.c_eps * y + a.coef.sigma * z;
          if (j == 0 && live) a.state_out[(size_t)row * a.AP + lane] = aval;
        }
      }
    }
    if (!(flags & IF_BLOCK)) continue;
    if (flags & IF_IN) {
      int kk = a.k;
      if (a.k_dev) kk = a.k_dev[rowc];
      const f32x4 sp = *reinterpret_cast<const f32x4*>(a.spart + (size_t)rowc * H + 4 * lane);
      const f32x4 ct = *reinterpret_cast<const f32x4*>(a.ctab + (size_t)kk * H + 4 * lane);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < a.A; ++i) {
        const float ai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(aval), i));
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.wa + (size_t)i * H + 4 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(ai, w4[e], acc[e]);
      }
      v = (acc + sp) + ct;
    }
    if (j == 0 && live) *reinterpret_cast<f32x4*>(a.hcur + (size_t)row * H + 4 * lane) = v;
    // LayerNorm over the row (eps 1e-6, fast variance), straight into the A-fragment tile
    const float s1 = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
    const float s2 = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
    const float mean = s1 * (1.0f / (float)H);
    const float var = fmaxf(s2 * (1.0f / (float)H) - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (v[e] - mean) * rstd * ls[e] + lb[e];
    *reinterpret_cast<f32x4*>(tA + (((lane >> 2) * 16 + rr) * 16 + swz(rr, lane & 3) * 4)) = y;
  }
  if (!(flags & IF_BLOCK)) return;
  __syncthreads();

  // ---- Dense_0 slice + relu -> LDS ------------------------------------------------------------------
  const int r = lane & 15, kq = lane >> 4;
  {
    f32x4 acc[NCB1];
#pragma unroll
    for (int c = 0; c < NCB1; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (RINGED) {
      if (!LDP_ABL(512)) {
#pragma unroll
        for (int ch = 0; ch < NCH1; ++ch) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(tA + (ch * 16 + r) * 16 + swz(r, kq) * 4);
#pragma unroll
          for (int c = 0; c < NCB1; ++c) {
            const int f = ch * NCB1 + c;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
              acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], ring[f % RING][s2], acc[c], 0, 0, 0);
            if (f + RING < NF) ring[f % RING] = fload(f + RING);
          }
        }
      }
    } else if (!LDP_ABL(512)) {
      idm_gemm<NCB1, NCH1, FR>(tA, a.w0, HID / 16, 0, j * (HSW / 16) + cb0, acc, lane);
    }
#pragma unroll
    for (int c = 0; c < NCB1; ++c) {
      const float b = a.b0[j * HSW + (cb0 + c) * 16 + ecol];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rw = erow0 + i;
        tZ[(((cb0 + c) * 16 + rw) * 16 + swz(rw, ecol >> 2) * 4 + (ecol & 3))] = fmaxf(acc[c][i] + b, 0.0f);
      }
    }
  }
  __syncthreads();

  // ---- K-partial of Dense_1 over this slice's hidden units -> part_out[j] ----------------------------
  {
    f32x4 acc[NCB2];
#pragma unroll
    for (int c = 0; c < NCB2; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (RINGED) {
      if (!LDP_ABL(1024)) {
#pragma unroll
        for (int ch = 0; ch < NCH2; ++ch) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(tZ + (ch * 16 + r) * 16 + swz(r, kq) * 4);
#pragma unroll
          for (int c = 0; c < NCB2; ++c) {
            const int f = NF1 + ch * NCB2 + c;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
              acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], ring[f % RING][s2], acc[c], 0, 0, 0);
            if (f + RING < NF) ring[f % RING] = fload(f + RING);
          }
        }
      }
    } else if (!LDP_ABL(1024)) {
      idm_gemm<NCB2, NCH2, FR>(tZ, a.w1, H / 16, j * NCH2, ob0, acc, lane);
    }
    float* po = a.part_out + ((size_t)j * a.Rp + r0) * H;
    if (!LDP_ABL(2048) || acc[0][0] == 12345.f) {
#pragma unroll
      for (int c = 0; c < NCB2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (STREAM || a.stream_parts) __builtin_nontemporal_store(acc[c][i], po + (size_t)(erow0 + i) * H + (ob0 + c) * 16 + ecol);
          else po[(size_t)(erow0 + i) * H + (ob0 + c) * 16 + ecol] = acc[c][i];
    }
  }
}


template <int HS, bool RINGED, bool STREAM>
static int idm_block_launch_ts(const IdmFusedArgs& a, int nrt, hipStream_t s) {
  constexpr int LDS = (16 * 256 + (1024 / HS) * 16) * 4;      // dynamic-LDS limit raised per device by idm_fused_init
  const bool block = (a.flags & IF_BLOCK) != 0;
  if (a.rt_major) hipLaunchKernelGGL((idm_block_kernel<HS, RINGED, STREAM>), dim3(nrt, block ? HS : 1), dim3(512), LDS, s, IDM_KERNEL_ARGS(a));
  else hipLaunchKernelGGL((idm_block_kernel<HS, RINGED, STREAM>), dim3(block ? HS : 1, nrt), dim3(512), LDS, s, IDM_KERNEL_ARGS(a));
  return (int)hipGetLastError();
}

template <int HS, bool RINGED>
static int idm_block_launch_t(const IdmFusedArgs& a, int nrt, hipStream_t s) {
  return a.stream_parts == 2 ? idm_block_launch_ts<HS, RINGED, true>(a, nrt, s) : idm_block_launch_ts<HS, RINGED, false>(a, nrt, s);
}

static int idm_block_launch(int hs, bool ringed, const IdmFusedArgs& a, int nrt, hipStream_t s) {
  switch (hs) {
    case 1: return idm_block_launch_t<1, false>(a, nrt, s);
    case 2: return idm_block_launch_t<2, false>(a, nrt, s);
    case 4: return ringed ? idm_block_launch_t<4, true>(a, nrt, s) : idm_block_launch_t<4, false>(a, nrt, s);
    case 8: return ringed ? idm_block_launch_t<8, true>(a, nrt, s) : idm_block_launch_t<8, false>(a, nrt, s);
  }
  return (int)hipErrorInvalidValue;
}

// =============================================================================================
// The same block on TWO fp16 planes per operand (round 5; tconv.hpp SPLIT = 3: x ~ h + l' / 2^11, three exact products on
// v_mfma_f32_16x16x32_f16, fp32 accumulate) over 32-row tiles, for every batch above 256 plans (1040 rows up).  The 16-row fp32 kernel is bound by its weight
// fragments there (1 KB of weights per four matrix instructions: 537 MB through the L2s per launch at 4096 rows, 0.53-0.61 of the fp32 MFMA
// peak): here a fragment pair feeds 2 row blocks x 3 products over four hidden slices per row tile (64 x 4 work-groups at 2048 rows)
// and the matrix instructions cost a fifth.  Same launch structure, flags, partial-sum hand-over and XCD placement as
// idm_block_kernel; LayerNorm(h) and relu(Dense_0) live in LDS as plane images ([row block][32-channel step][plane][64 lanes][8 halves] = the
// A fragment of the 16x16x32 instruction, lane = 16 (k / 8) + row); the weights are pack_conv_split16h(kernel, 1 tap).  Range guard as in
// tconv.hpp: an operand that leaves the planes' range (|x| >= 65504 or not finite) raises fault word [1]; the call is recomputed on the
// exact-fp32 kernel (ldp_handle::range_fallback).
// =============================================================================================
template <int MB, int NCB, int NK, int PF>
__device__ __forceinline__ void idm_gemm_h16(const float* __restrict__ tile, const float* __restrict__ wpk, int ncb_total, int k0, int cb0,
                                             f32x4 (&acc)[MB][NCB], f32x4 (&lo)[MB][NCB], int lane) {
  static_assert(PF <= NK, "prefetch depth");
  f32x4 wb[PF][NCB][2];
  auto wload = [&](int k, f32x4 (&b)[NCB][2]) {
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        b[c][pl] = *reinterpret_cast<const f32x4*>(wpk + ((((size_t)(k0 + k) * ncb_total + cb0 + c) * 2 + pl) * 256 + lane * 4));
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) wload(p, wb[p]);
  constexpr int FA[3] = {1, 0, 0}, FB[3] = {0, 1, 0};      // l' h and h l' into the low accumulator, h h into the main one
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    f32x4 av[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        av[m][pl] = *reinterpret_cast<const f32x4*>(tile + ((((m * NK + k) * 2 + pl) * 64 + lane) * 4));
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int pi = 0; pi < 3; ++pi) {
        const f16x8_t bv = __builtin_bit_cast(f16x8_t, wb[k % PF][c][FB[pi]]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const f16x8_t a8 = __builtin_bit_cast(f16x8_t, av[m][FA[pi]]);
          f32x4& dst = pi < 2 ? lo[m][c] : acc[m][c];
          dst = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, bv, dst, 0, 0, 0);
        }
      }
    if (k + PF < NK) wload(k + PF, wb[k % PF]);
  }
}

template <int HS>
__global__ __launch_bounds__(512) void idm_block_h16_kernel(IDM_KERNEL_PARAMS) {
#if LDP_KERNARG_PRELOAD
  IdmFusedArgs a = a_in;
  a.w0 = h_w0; a.w1 = h_w1; a.part_prev = h_part_prev; a.hprev = h_hprev; a.state_in = h_state_in;
  a.R = h_R; a.Rp = h_Rp; a.flags = h_flags;
  a.AP = h_pk & 255; a.rt_major = (h_pk >> 8) & 1; a.stream_parts = (h_pk >> 9) & 3; a.dbg = (unsigned)h_pk >> 12;
#endif
  constexpr int MB = 2, NR = 16 * MB, RPW = NR / 8;        // rows per work-group / per wave
  constexpr int H = 256, HID = 4 * H, HSW = HID / HS;
  constexpr int NK1 = H / 32, NCB1 = HSW / 16 / 8, NK2 = HSW / 32, NCB2 = 2;
  static_assert(NCB1 >= 1 && NCB1 <= 4, "2 or 4 hidden slices");
  extern __shared__ f32x4 smem4[];
  float* tA = reinterpret_cast<float*>(smem4);        // LayerNorm(h) as plane images: MB * NK1 * 2 units of 256 floats
  float* tZ = tA + MB * NK1 * 2 * 256;                // relu(Dense_0) likewise: MB * NK2 * 2 units
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = a.rt_major ? blockIdx.y : blockIdx.x, r0 = (a.rt_major ? blockIdx.x : blockIdx.y) * NR;
  const int flags = a.flags;
  const int ecol = lane & 15, erow0 = (lane >> 4) * 4;
  const int cb0 = wave * NCB1, ob0 = wave * NCB2;
  bool range_bad = false;

  // ---- prologue: this wave's four rows, four columns per lane (idm_block_kernel's, over RPW rows) ------------------
  int rowq[RPW], rowcq[RPW];
  f32x4 pv[RPW][HS], hp[RPW];
  float av[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    rowq[q] = r0 + wave + 8 * q;
    rowcq[q] = rowq[q] < a.R ? rowq[q] : a.R - 1;
    if (flags & (IF_RED | IF_TAIL)) {
#pragma unroll
      for (int jj = 0; jj < HS; ++jj) pv[q][jj] = a.stream_parts ? ld_stream(a.part_prev + ((size_t)jj * a.Rp + rowcq[q]) * H + 4 * lane)
                                                                  : *reinterpret_cast<const f32x4*>(a.part_prev + ((size_t)jj * a.Rp + rowcq[q]) * H + 4 * lane);
      hp[q] = *reinterpret_cast<const f32x4*>(a.hprev + (size_t)rowcq[q] * H + 4 * lane);
    }
    av[q] = 0.0f;
    if (lane < a.AP) av[q] = a.state_in[(size_t)rowcq[q] * a.AP + lane];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // non-temporal and plain loads may overtake each other (idm_block_kernel)
  f32x4 b1v = f32x4{0.f, 0.f, 0.f, 0.f}, ls = b1v, lb = b1v;
  if (flags & (IF_RED | IF_TAIL)) b1v = *reinterpret_cast<const f32x4*>(a.b1_prev + 4 * lane);
  if (flags & IF_BLOCK) {
    ls = *reinterpret_cast<const f32x4*>(a.ln_s + 4 * lane);
    lb = *reinterpret_cast<const f32x4*>(a.ln_b + 4 * lane);
  }
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int rr = wave + 8 * q;                       // row of the work-group's tile: row block rr >> 4, row rr & 15
    const int row = rowq[q], rowc = rowcq[q];
    const bool live = row < a.R;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (flags & (IF_RED | IF_TAIL)) {
      f32x4 acc = pv[q][0];
#pragma unroll
      for (int jj = 1; jj < HS; ++jj) acc = acc + pv[q][jj];
      acc = acc + b1v;
      v = hp[q] + acc;
    }
    float aval = av[q];
    if (flags & IF_TAIL) {
      const f32x4 hl = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      float my_eps = 0.0f;
      for (int a0 = 0; a0 < a.A; a0 += 8) {
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ai = (a0 + u) < a.A ? a0 + u : a.A - 1;
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.wout_t + (size_t)ai * H + 4 * lane);
          float t = hl[0] * w4[0];
          t = fmaf(hl[1], w4[1], t);
          t = fmaf(hl[2], w4[2], t);
          t = fmaf(hl[3], w4[3], t);
          p[u] = t;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0xB1, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x4E, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x114, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x118, 0xF>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x142, 0xA>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = dpp_add<0x143, 0xC>(p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[u]), 63));
          if (lane == a0 + u) my_eps = tot;
        }
      }
      if (lane < a.A) {
        const float y = my_eps + a.bout[lane];
        if ((flags & IF_EPSOUT) && live) a.eps_out[(size_t)row * a.A + lane] = y;
        if (flags & IF_STEP) {
          const float xt = aval;
          float z = 0.f;
          if (a.coef.sigma != 0.f) {
            if (a.noise) z = a.noise[(size_t)rowc * a.A + lane];
            else z = philox_normal(a.ctl[0], (a.ctl[1] + (uint64_t)row) * (uint64_t)a.AP + (uint64_t)lane, (uint32_t)a.step, 0u);
          }
          float x0 = (xt - a.coef.sqrt_1mab * y) * a.coef.inv_sqrt_ab;
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          aval = a.coef.c_x0 * x0 + a.coef.c_x * xt + a.coef.c_eps * y + a.coef.sigma * z;
          if (j == 0 && live) a.state_out[(size_t)row * a.AP + lane] = aval;
        }
      }
    }
    if (!(flags & IF_BLOCK)) continue;
    if (flags & IF_IN) {
      int kk = a.k;
      if (a.k_dev) kk = a.k_dev[rowc];
      const f32x4 sp = *reinterpret_cast<const f32x4*>(a.spart + (size_t)rowc * H + 4 * lane);
      const f32x4 ct = *reinterpret_cast<const f32x4*>(a.ctab + (size_t)kk * H + 4 * lane);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < a.A; ++i) {
        const float ai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(aval), i));
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.wa + (size_t)i * H + 4 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(ai, w4[e], acc[e]);
      }
      v = (acc + sp) + ct;
    }
    if (j == 0 && live) *reinterpret_cast<f32x4*>(a.hcur + (size_t)row * H + 4 * lane) = v;
    const float s1 = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
    const float s2 = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
    const float mean = s1 * (1.0f / (float)H);
    const float var = fmaxf(s2 * (1.0f / (float)H) - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // Plain expression.  Round 5 saw rows normalised with a ZERO mean in lanes 48..63 here and papered over it with empty asm statements; round 6
      // found the cause (tools/r6/pk_f32_repro.hip): the compiler packed the subtraction into `v_pk_add_f32 d, x, (E[x^2], mean) op_sel:[0,1]`
      // -- the LOW result lane reading the HIGH dword of a register pair -- and on gfx950 that operand selection returns 0.0 in lanes 48..63 when another
      // wave's v_mfma_f32_16x16x32_f16 issues next to it on the same SIMD.  The library is therefore built WITHOUT packed-fp32 instructions
      // (csrc/Makefile NOPKF32; tests/test_abi.py audits the code objects).
      y[e] = (v[e] - mean) * rstd * ls[e] + lb[e];
      range_bad = range_bad || !(fabsf(y[e]) < 65504.0f);
    }
    // columns 4 lane .. 4 lane + 3 = half a 16-byte unit of step lane >> 3, k quarter (lane & 7) >> 1
    uint2 ph, pl;
    split4h(y, ph, pl);
    float* dst = tA + (((((rr >> 4) * NK1 + (lane >> 3)) * 2) * 64 + ((lane & 7) >> 1) * 16 + (rr & 15)) * 4 + (lane & 1) * 2);
    *reinterpret_cast<uint2*>(dst) = ph;
    *reinterpret_cast<uint2*>(dst + 256) = pl;
  }
  if (!(flags & IF_BLOCK)) return;
  __syncthreads();

  // ---- Dense_0 slice + relu -> plane images in LDS --------------------------------------------------
  {
    f32x4 acc[MB][NCB1], lo[MB][NCB1];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int c = 0; c < NCB1; ++c) acc[m][c] = lo[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    idm_gemm_h16<MB, NCB1, NK1, (NCB1 >= 4 ? 2 : 4)>(tA, a.w0, HID / 16, 0, j * (HSW / 16) + cb0, acc, lo, lane);
    unsigned short* zh = reinterpret_cast<unsigned short*>(tZ);
#pragma unroll
    for (int c = 0; c < NCB1; ++c) {
      const float b = a.b0[j * HSW + (cb0 + c) * 16 + ecol];
      const int k32 = ((cb0 + c) & 1) * 16 + ecol;            // this column as a K index of Dense_1: step (cb0 + c) >> 1, element k32 of it
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = fmaxf((acc[m][c][i] + lo[m][c][i] * (1.0f / 2048.0f)) + b, 0.0f);
          range_bad = range_bad || !(z < 65504.0f);
          const _Float16 zhh = (_Float16)z;
          const _Float16 zll = (_Float16)((z - (float)zhh) * 2048.0f);
          const int u = (((m * NK2 + ((cb0 + c) >> 1)) * 2) * 64 + (k32 >> 3) * 16 + erow0 + i) * 8 + (k32 & 7);
          zh[u] = __builtin_bit_cast(unsigned short, zhh);
          zh[u + 512] = __builtin_bit_cast(unsigned short, zll);
        }
    }
  }
  __syncthreads();

  // ---- K-partial of Dense_1 over this slice's hidden units -> part_out[j] ----------------------------
  {
    f32x4 acc[MB][NCB2], lo[MB][NCB2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int c = 0; c < NCB2; ++c) acc[m][c] = lo[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    idm_gemm_h16<MB, NCB2, NK2, 4>(tZ, a.w1, H / 16, j * NK2, ob0, acc, lo, lane);
    float* po = a.part_out + ((size_t)j * a.Rp + r0) * H;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int c = 0; c < NCB2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          const float pvv = acc[m][c][i] + lo[m][c][i] * (1.0f / 2048.0f);
          float* pp = po + (size_t)(m * 16 + erow0 + i) * H + (ob0 + c) * 16 + ecol;
          if (a.stream_parts) __builtin_nontemporal_store(pvv, pp); else *pp = pvv;
        }
  }
  if (LDP_RANGE_GUARD && range_bad && a.fault) a.fault[1] = 1u;
}

template <int HS>
static int idm_block_h16_launch_t(const IdmFusedArgs& a, int nrt, hipStream_t s) {
  constexpr int LDS = 2 * (256 / 32 + (1024 / HS) / 32) * 2 * 256 * 4;
  const bool block = (a.flags & IF_BLOCK) != 0;
  if (a.rt_major) hipLaunchKernelGGL((idm_block_h16_kernel<HS>), dim3(nrt, block ? HS : 1), dim3(512), LDS, s, IDM_KERNEL_ARGS(a));
  else hipLaunchKernelGGL((idm_block_h16_kernel<HS>), dim3(block ? HS : 1, nrt), dim3(512), LDS, s, IDM_KERNEL_ARGS(a));
  return (int)hipGetLastError();
}
static int idm_block_h16_launch(int hs, const IdmFusedArgs& a, int nrt, hipStream_t s) {
  return hs == 2 ? idm_block_h16_launch_t<2>(a, nrt, s) : hs == 4 ? idm_block_h16_launch_t<4>(a, nrt, s) : (int)hipErrorInvalidValue;
}

// raise the dynamic-LDS limit of every instantiation outside any stream capture
static int idm_fused_init() {
  auto set = [](const void* k, int lds) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
  auto lds = [](int hs) { return (16 * 256 + (1024 / hs) * 16) * 4; };
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<1, false, false>), lds(1));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<1, false, true>), lds(1));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<2, false, false>), lds(2));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<2, false, true>), lds(2));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<4, false, false>), lds(4));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<4, false, true>), lds(4));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<4, true, false>), lds(4));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<4, true, true>), lds(4));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<8, false, false>), lds(8));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<8, false, true>), lds(8));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<8, true, false>), lds(8));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_kernel<8, true, true>), lds(8));
  auto ldsh = [](int hs) { return 2 * (256 / 32 + (1024 / hs) / 32) * 2 * 256 * 4; };
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_h16_kernel<2>), ldsh(2));
  if (e == hipSuccess) e = set(reinterpret_cast<const void*>(idm_block_h16_kernel<4>), ldsh(4));
  if (e != hipSuccess) return fail(LDP_EHIP, "hipFuncSetAttribute(idm_block_kernel): %s", hipGetErrorString(e));
  return LDP_OK;
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
  if (I.A < 1 || I.A > 32) return fail(LDP_EINVAL, "action_dim must be in 1..32");
  LDP_TRY(idm_fused_init());
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
  {
    // fused tail: W_out transposed to (A, H) so a lane's four columns are one float4
    const HostTensor *ok = nullptr, *ob = nullptr;
    auto it = h->weights.find(root + "MLPResNet_0/Dense_1/kernel");
    ok = &it->second;                                   // shape (1, H, A) after dense_w
    LDP_TRY(get_weight(h, root + "MLPResNet_0/Dense_1/bias", &ob, {A}));
    std::vector<float> wt((size_t)A * H);
    for (int c = 0; c < H; ++c)
      for (int ai = 0; ai < A; ++ai) wt[(size_t)ai * H + c] = ok->data[(size_t)c * A + ai];
    LDP_TRY(upload(I.wout_t, wt.data(), wt.size() * 4, s));
    LDP_TRY(upload(I.bout, ob->data.data(), (size_t)A * 4, s));
  }
  LDP_HIP(hipStreamSynchronize(s));
  I.ready = true;
  return LDP_OK;
}

static bool idm_use_fused(const ldp_handle* h) {
  return !h->opt.idm_unfused && h->idm.H == 256 && h->idm.NB >= 1;
}

// hidden slices per row tile.  Measured (profiles/r02_idm_split_sweep.json, profiles/r03_idm_hs_sweep.txt): the loop time is a
// staircase in the number of ROUNDS r = ceil(row tiles x slices / CUs) the grid needs, with a per-round price that falls
// with the split (less work per work-group) while the partial-sum traffic grows with its square.  The split with the
// cheapest staircase wins; prices in ms per 100-step loop on MI355X (only their ratios matter):
//   1 slice : 3.7 + 12.6 r        2 slices: 9.3, then 14.0 + 6.7 (r - 2)        4 slices: 5.1, 8.5, then 8.5 + 4.1 (r - 2)
//   8 slices only while the whole grid is one round (latency-bound batches: <= 32 row tiles on 256 CUs).
// Round 2's rule (largest split with at most two work-groups per CU) ignored the staircase: 24 % slower at 5120-6144
// rows, 8-11 % at 2560-3072, 14 % at 10240.
static int idm_hidden_split(const ldp_handle* h, int R) {
  if (h->opt.idm_hs) return h->opt.idm_hs;
  const int nrt = (R + 15) / 16, cu = h->n_cu > 0 ? h->n_cu : 256;
  if (nrt * 8 <= cu) return 8;
  auto rounds = [&](int hs) { return (nrt * hs + cu - 1) / cu; };
  const int r1 = rounds(1), r2 = rounds(2), r4 = rounds(4);
  const float c1 = 3.7f + 12.6f * r1;
  const float c2 = r2 == 1 ? 9.3f : 14.0f + 6.7f * (r2 - 2);
  const float c4 = r4 == 1 ? 5.1f : 8.5f + 4.1f * (r4 - 2);
  if (c1 <= c2 && c1 <= c4) return 1;
  return c2 <= c4 ? 2 : 4;
}

static int idm_split_prepare(ldp_handle* h);

int idm_workspace(ldp_handle* h, int R) {
  IdmState& I = h->idm;
  if (h->opt.idm_f16 && I.H == 256 && !h->opt.idm_unfused && round_up(R, 16) >= h->opt.idm_f16_min_rows) LDP_TRY(idm_split_prepare(h));
  if (R <= I.ws_R) return LDP_OK;
  drop_graphs(h);
  const int Rp = round_up(R, 64);
  LDP_TRY(I.state.alloc((size_t)Rp * I.AP * 4));
  LDP_TRY(I.state2.alloc((size_t)Rp * I.AP * 4));
  LDP_TRY(I.trans.alloc((size_t)Rp * 2 * I.D * 4));
  LDP_TRY(I.spart.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.h0.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.h1.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.y.alloc((size_t)Rp * I.H * 4));
  LDP_TRY(I.z.alloc((size_t)Rp * 4 * I.H * 4));
  LDP_TRY(I.part0.alloc((size_t)8 * Rp * I.H * 4));
  LDP_TRY(I.part1.alloc((size_t)8 * Rp * I.H * 4));
  LDP_HIP(hipMemset(I.state.p, 0, (size_t)Rp * I.AP * 4));
  LDP_HIP(hipMemset(I.state2.p, 0, (size_t)Rp * I.AP * 4));
  LDP_HIP(hipMemset(I.h0.p, 0, (size_t)Rp * I.H * 4));
  LDP_HIP(hipMemset(I.h1.p, 0, (size_t)Rp * I.H * 4));
  LDP_HIP(hipMemset(I.spart.p, 0, (size_t)Rp * I.H * 4));
  // the loops run over whole 16-row buckets: the rows behind the last real one are computed and never stored, but they must be FINITE --
  // whatever hipMalloc hands out could trip the range guard of the fp16-plane kernel (planner_workspace does the same for its state)
  LDP_HIP(hipMemset(I.trans.p, 0, (size_t)Rp * 2 * I.D * 4));
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
  a.dbg = h->opt.dbg;
  const int r = tconv_launch(p, a, s);
  h->last_conv_launches++;
  h->last_total_launches++;
  if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "IDM dense launch failed (%d)", r);
  return LDP_OK;
}

// ---- round-1 path: one launch per Dense / LayerNorm (kept for idm_hidden != 256 and as a cross-check) ----
static int idm_forward_unfused(ldp_handle* h, int R, const int* k_dev, int k, bool step,
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
  e.noise = noise; e.seed = h->ctl_idm(); e.step = step_idx; e.eps_out = eps_out;
  const int flags = (step ? EP_STEP : 0) | (eps_out ? EP_EPSOUT : 0);
  LDP_TRY(p1(h, I.out, cur, H, I.state.f(), flags, nullptr, Bq, s, &e));
  return LDP_OK;
}

// ---- fused path ---------------------------------------------------------------------------------------
namespace {
struct FusedSeq {            // ping-pong bookkeeping of one enqueue sequence (host side, deterministic per R / n_steps)
  ldp_handle* h;
  IdmState& I;
  int R, hs, nrt;
  bool f16 = false;            // the 32-row kernel on fp16 planes (idm_block_h16_kernel); nrt then counts 32-row tiles
  int hi = 0, pi = 0, si = 0;
  hipStream_t s = nullptr;
  float* hbuf(int i) const { return i ? I.h1.f() : I.h0.f(); }
  float* pbuf(int i) const { return i ? I.part1.f() : I.part0.f(); }
  float* sbuf(int i) const { return i ? I.state2.f() : I.state.f(); }

  IdmFusedArgs base() const {
    IdmFusedArgs a{};
    a.R = R; a.Rp = I.ws_R; a.A = I.A; a.AP = I.AP;
    a.hs_prev = hs;
    a.wout_t = I.wout_t.f(); a.bout = I.bout.f();
    a.wa = I.in_a.w.f(); a.spart = I.spart.f(); a.ctab = I.ctab.f();
    a.ctl = h->ctl_idm();
    a.dbg = h->opt.dbg;
    a.rt_major = h->opt.idm_rt_major;
    a.stream_parts = h->opt.idm_stream < 0 ? (nrt >= 128 ? 2 : 0) : h->opt.idm_stream;
    a.fault = h->fault_dev;
    return a;
  }
  int launch(IdmFusedArgs& a) {
    // the ringed variant needs a CU per work-group (238 VGPRs); with two work-groups per CU the plain one is faster
    // (measured: -3 % at 64..256 plans, nothing to gain below 16 row tiles where the launch floor is all there is)
    const bool ringed = hs >= 4 && nrt * hs <= h->n_cu && nrt >= 16 && !h->opt.idm_noring;
    const int r = f16 ? idm_block_h16_launch(hs, a, nrt, s) : idm_block_launch(hs, ringed, a, nrt, s);
    if (f16 && (a.flags & IF_BLOCK)) h->stat_f16_launches++;
    h->last_total_launches++;
    if (a.flags & IF_BLOCK) h->last_conv_launches++;
    if (r != 0) return fail(LDP_EHIP, "fused IDM block launch failed: %s", hipGetErrorString((hipError_t)r));
    return LDP_OK;
  }
  // one evaluation's NB block launches; the first carries the tail (scheduler update) of the previous step
  int blocks(const int* k_dev, int k, bool tail_prev, const StepCoef* coef_prev, const float* noise_prev,
             int step_prev) {
    for (int b = 0; b < I.NB; ++b) {
      IdmFusedArgs a = base();
      a.flags = IF_BLOCK | (b == 0 ? IF_IN : IF_RED);
      a.hprev = hbuf(hi); a.part_prev = pbuf(pi);
      a.b1_prev = I.blks[(b + I.NB - 1) % I.NB].d1.bias.f();
      a.hcur = hbuf(1 - hi); a.part_out = pbuf(1 - pi);
      a.state_in = sbuf(si); a.state_out = sbuf(1 - si);
      a.k_dev = k_dev; a.k = k;
      if (b == 0 && tail_prev) {
        a.flags |= IF_TAIL | IF_STEP;
        a.coef = *coef_prev; a.noise = noise_prev; a.step = step_prev;
      }
      a.ln_s = I.blks[b].ln_s.f(); a.ln_b = I.blks[b].ln_b.f();
      a.w0 = I.blks[b].d0.w.f(); a.b0 = I.blks[b].d0.bias.f(); a.w1 = I.blks[b].d1.w.f();
      if (f16) { a.w0 = I.blks[b].d0.wsplit16h.f(); a.w1 = I.blks[b].d1.wsplit16h.f(); }
      LDP_TRY(launch(a));
      if (b == 0 && tail_prev) si = 1 - si;
      hi = 1 - hi; pi = 1 - pi;
    }
    return LDP_OK;
  }
  // tail of the last evaluation: eps (+ scheduler update)
  int tail(bool step, const StepCoef* coef, const float* noise, int step_idx, float* eps_out) {
    IdmFusedArgs a = base();
    a.flags = IF_TAIL | (step ? IF_STEP : 0) | (eps_out ? IF_EPSOUT : 0);
    a.hprev = hbuf(hi); a.part_prev = pbuf(pi);
    a.b1_prev = I.blks[I.NB - 1].d1.bias.f();
    a.state_in = sbuf(si); a.state_out = sbuf(1 - si);
    if (coef) a.coef = *coef;
    a.noise = noise; a.step = step_idx; a.eps_out = eps_out;
    LDP_TRY(launch(a));
    if (step) si = 1 - si;
    return LDP_OK;
  }
};
}  // namespace

// From idm_f16_min_rows rows (1040: every batch above 256 plans) the blocks run on fp16 planes over 32-row tiles (idm_block_h16_kernel) unless the
// handle fell back to the fp32 range (range guard), a weight does not fit the planes, or the split was forced by option.  Always FOUR hidden slices
// (a row's values then do not depend on the batch size): same-box loops, ms per 100 steps (tools/r5/idm_f16_probe.py and the sweep behind
// profiles/r05_idm_f16_probe.txt): rows 1040 / 2048 / 3072 / 4096 / 6144: exact fp32 8.47 / 8.86 / 13.25 / 15.15 / 21.29, four slices 6.54 / 6.32 /
// 7.65 / 8.19 / 14.00, two slices 8.19 / 8.21 / 8.33 / 8.58 / 15.98 (two work-groups per CU from 2080 rows up: soaked, profiles/r05_idm_f16_soak.txt).
static bool idm_f16_at(const ldp_handle* h, int R) {
  const IdmState& I = h->idm;
  if (!h->opt.idm_f16 || h->range_fallback || h->opt.idm_hs || !idm_use_fused(h) || R < h->opt.idm_f16_min_rows) return false;
  for (const auto& b : I.blks)
    if (!b.d0.wsplit16h.p || !b.d1.wsplit16h.p) return false;
  return true;
}
static FusedSeq make_seq(ldp_handle* h, int R, hipStream_t s) {
  IdmState& I = h->idm;
  FusedSeq f{h, I, R, idm_hidden_split(h, R), (R + 15) / 16};
  if (idm_f16_at(h, R)) {
    f.f16 = true;
    f.nrt = (R + 31) / 32;
    f.hs = 4;
    if (h->opt.idm_f16_hs == 2 || h->opt.idm_f16_hs == 4) f.hs = h->opt.idm_f16_hs;      // A/B: 2 slices
  }
  f.s = s;
  return f;
}

// plane-packed copies of the blocks' two Dense kernels, built the first time a batch of idm_f16_min_rows rows arrives (4 MB)
static int idm_split_prepare(ldp_handle* h) {
  IdmState& I = h->idm;
  for (int i = 0; i < I.NB; ++i) {
    const std::string p = "idm/MLPResNet_0/MLPResNetBlock_" + std::to_string(i);
    struct { ConvW* c; const char* name; int cin, cout; } ds[2] = {{&I.blks[i].d0, "/Dense_0/kernel", I.H, 4 * I.H}, {&I.blks[i].d1, "/Dense_1/kernel", 4 * I.H, I.H}};
    for (auto& d : ds) {
      if (d.c->wsplit16h.p || d.c->f16_refused) continue;
      const HostTensor* k = nullptr;
      LDP_TRY(get_weight(h, p + d.name, &k, {1, d.cin, d.cout}));
      if (!fits_f16_planes(k->data.data(), k->data.size())) { d.c->f16_refused = true; continue; }
      const std::vector<uint16_t> sp = pack_conv_split16h(k->data.data(), 1, d.cin, d.cout);
      LDP_TRY(d.c->wsplit16h.alloc(sp.size() * 2));
      LDP_HIP(hipMemcpy(d.c->wsplit16h.p, sp.data(), sp.size() * 2, hipMemcpyHostToDevice));
    }
  }
  return LDP_OK;
}

// Every IDM call gets a fresh epoch in its control block (exchange tags of the 32-row kernel: 18 bits of it plus the
// launch index); the granule slab is wiped every 2^17 calls so that a tag written 2^18 calls ago can never validate.
static int idm_new_epoch(ldp_handle* h, uint64_t seed, int64_t row_offset, hipStream_t s) {
  IdmState& I = h->idm;
  ++h->epoch_idm;
  return set_seed_launch(h->ctl_idm(), seed, row_offset, h->epoch_idm, s);
}

int idm_pre(ldp_handle* h, const float* transition, const float* a_init, const float* step_noise, uint64_t seed,
            int64_t row_offset, const LoopSpec& L, int R, hipStream_t s) {
  IdmState& I = h->idm;
  const size_t per_step = (size_t)R * I.A;
  if (transition)
    LDP_HIP(hipMemcpyAsync(I.trans.p, transition, (size_t)R * 2 * I.D * 4, hipMemcpyDeviceToDevice, s));
  // the IDM's Philox stream is decorrelated from the planner's by flipping the seed's top bit; draws are
  // keyed by the global ROW (row_offset + local row), so any row_offset shards consistently
  LDP_TRY(idm_new_epoch(h, seed ^ 0x8000000000000000ull, row_offset, s));
  if (a_init) LDP_TRY(pad_rows_launch(a_init, I.state.f(), R, I.A, I.AP, s));
  else LDP_TRY(philox_init_launch(I.state.f(), 1, R, I.A, I.AP, h->ctl_idm(), s));
  if (L.explicit_noise) {
    if (per_step * L.n_steps * 4 > I.noise.bytes) drop_graphs(h);
    LDP_TRY(I.noise.alloc(per_step * L.n_steps * 4));
    LDP_HIP(hipMemcpyAsync(I.noise.p, step_noise, per_step * L.n_steps * 4, hipMemcpyDeviceToDevice, s));
  }
  // where a_0 will be once the loop has run: the fused path ping-pongs the state once per step
  I.state_cur = idm_use_fused(h) ? (L.n_steps & 1) : 0;
  return LDP_OK;
}

static int idm_spart(ldp_handle* h, int R, hipStream_t s) {
  IdmState& I = h->idm;
  h->last_total_launches++;
  return dense_launch(I.trans.f(), 2 * I.D, I.w_in_s.f(), I.H, I.b_in.f(), I.spart.f(), I.H, R, 2 * I.D,
                      I.H, 0, 0, s);
}

int idm_loop(ldp_handle* h, int R, const LoopSpec& L, hipStream_t q) {
  IdmState& I = h->idm;
  std::vector<StepCoef> coefs;
  make_step_coefs(I.n_train, L.n_steps, L.sampler, coefs);
  const size_t per_step = (size_t)R * I.A;
  auto nz = [&](int i) { return L.explicit_noise ? I.noise.f() + per_step * i : nullptr; };
  LDP_TRY(idm_spart(h, R, q));
  if (!idm_use_fused(h)) {
    for (int i = 0; i < L.n_steps; ++i)
      LDP_TRY(idm_forward_unfused(h, R, nullptr, (int)coefs[i].t, true, &coefs[i], nz(i), i, nullptr, q));
    return LDP_OK;
  }
  FusedSeq f = make_seq(h, R, q);
  for (int i = 0; i < L.n_steps; ++i)
    LDP_TRY(f.blocks(nullptr, (int)coefs[i].t, i > 0, i > 0 ? &coefs[i - 1] : nullptr, i > 0 ? nz(i - 1) : nullptr, i - 1));
  LDP_TRY(f.tail(true, &coefs[L.n_steps - 1], nz(L.n_steps - 1), L.n_steps - 1, nullptr));
  if (f.si != I.state_cur) return fail(LDP_ESTATE, "IDM state ping-pong out of step");
  return LDP_OK;
}

const float* idm_result(ldp_handle* h) { return h->idm.state_cur ? h->idm.state2.f() : h->idm.state.f(); }

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
  LDP_HIP(hipMemcpyAsync(I.trans.p, sT, (size_t)R * 2 * I.D * 4, hipMemcpyDeviceToDevice, s));
  LDP_TRY(pad_rows_launch(a, I.state.f(), R, I.A, I.AP, s));
  LDP_TRY(idm_new_epoch(h, 0, 0, s));                 // no random draws here: only the exchange tags need it
  LDP_TRY(idm_spart(h, R, s));
  if (!idm_use_fused(h)) return idm_forward_unfused(h, R, k_dev, k, false, nullptr, nullptr, 0, eps, s);
  FusedSeq f = make_seq(h, R, s);
  LDP_TRY(f.blocks(k_dev, k, false, nullptr, nullptr, 0));
  return f.tail(false, nullptr, nullptr, 0, eps);
}

int ldp_idm_sample(ldp_handle* h, const float* transition, const float* a_init, const float* step_noise,
                   uint64_t seed, int64_t row_offset, int32_t sampler, int32_t n_steps, float* out,
                   int32_t R, int32_t use_graph, void* stream) {
  if (!h || !transition || !out || R <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!h->idm.ready) return fail(LDP_ESTATE, "idm weights not finalized");
  IdmState& I = h->idm;
  LDP_TRY(check_sampler(sampler, n_steps, I.n_train, "idm"));
  LDP_TRY(entry_fault_check(h));
  hipStream_t s = (hipStream_t)stream;
  LDP_TRY(idm_workspace(h, R));
  h->last_conv_launches = h->last_total_launches = 0;
  LoopSpec L{n_steps, sampler, step_noise != nullptr && sampler == LDP_SAMPLER_DDPM};
  LDP_TRY(idm_pre(h, transition, a_init, step_noise, seed, row_offset, L, R, s));
  const int Rg = bucket_rows(R, L.explicit_noise);
  GraphKey key{1, Rg, n_steps, sampler, L.explicit_noise ? 1 : 0};
  LDP_TRY(run_or_replay(h, key, use_graph != 0, s, [&](hipStream_t q) { return idm_loop(h, Rg, L, q); }));
  LDP_TRY(unpad_rows_launch(idm_result(h), out, R, I.A, I.AP, s));
  return LDP_OK;
}

}  // extern "C"
