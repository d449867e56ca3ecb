// tconv.hpp -- "Toeplitz" implicit-GEMM 1-D convolution on gfx950 f32 MFMA, with the
// U-Net's epilogues fused (bias, GroupNorm, Mish, FiLM, residual add, DDPM/DDIM update).
//
// Computes, for the reference's channels-last (B, T, C) tensors
// (networks/diffusion_nets_v2.py:51-102,162-167):
//     out[b, to, :] = sum_{(ti, j) in S(to)}  x[b, ti, :] @ W[j]        (+ epilogue)
// where S(to) is the static tap set of the layer kind:
//     K5   : Conv(k=5, pad 2)                     ti = to + j - 2
//     DOWN : Conv(k=3, stride 2, XLA SAME (0,1))  ti = 2 to + j
//     UP   : ConvTranspose(k=4, s=2, SAME, kernel not flipped)
//            out[2q] = x[q-1] K0 + x[q] K2 ; out[2q+1] = x[q] K1 + x[q+1] K3
//     P1   : Conv(k=1) / Dense
// Taps that fall on zero padding are never issued (at T=2 a k=5 conv is 4 of 10 MFMAs).
//
// Mapping to CDNA4 (wave64, v_mfma_f32_16x16x4_f32, exact fp32 = fmaf chain):
//   * MFMA rows = 16 *samples* at one time position, MFMA cols = 16 output channels.  A wave
//     owns all TO positions of its 16 samples x 16 channels: acc[TO] (4 VGPRs each).  For
//     every 16-channel input chunk it needs TI A-fragments (one per input position) and NJ
//     B-fragments (one per tap) and issues |S| MFMAs per k-step: the activations are reused
//     across taps in registers (13 fragment loads feed 136 MFMAs at T=8).
//   * A work-group is 16 samples x TO x BN channels, BN = 16*NWN = one GroupNorm group, so
//     the GroupNorm statistics never leave the work-group.  Its 64*NWN*KS threads are NWN
//     waves along channels x KS waves splitting the input channels; the KS partial sums are
//     combined through LDS in the epilogue.
//   * Weights are pre-packed so that one wave's B fragment for (chunk, tap, 16-col block) is
//     1 KiB contiguous (lane-linear float4): streamed global->VGPR, no LDS, prefetched one
//     iteration ahead.  Activations are staged global->LDS (double buffered, one barrier per
//     iteration) in 16x16 sub-tiles whose 16-byte slots are XOR-swizzled so the ds_read_b128
//     fragment reads are bank-conflict-free.
//   * blockIdx % 8 selects the channel block, so (observed) each XCD's L2 holds the weight
//     columns of one GroupNorm group only; all sample blocks of that group hit in L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// No implicit mul+add fusion anywhere in the epilogues: with contraction left to the optimiser, two
// unrolled copies of the same expression (sample i vs sample i+8 of a tile) may be fused differently
// and a plan's value would depend on its position in the batch (breaks shard invariance by 1 ulp).
#pragma clang fp contract(off)

#ifndef LDP_W_NT
#define LDP_W_NT 0
#endif
#ifndef LDP_SPLIT_NPROD        // A/B builds only: 9 = all nine plane products in the 16-row split tiles (the exact product; VERDICT r3's form for the headline), default 6
#define LDP_SPLIT_NPROD 6
#endif
#ifndef LDP_S16_LF             // 16-row split tiles: the next step's weight loads are issued over the first LDP_S16_LF percent of a step's matrix instructions
#define LDP_S16_LF 100
#endif
#ifndef LDP_F16_MINW           // minimum waves per SIMD asked of the four-wave fp16-plane tiles (A/B build: 1)
#define LDP_F16_MINW 2
#endif
#ifndef LDP_RANGE_GUARD        // 0: A/B build without the range guard of the fp16-plane tiles (`make noguard`: measures what the guard costs)
#define LDP_RANGE_GUARD 1
#endif
#ifndef LDP_KERNARG_TOUCH
#define LDP_KERNARG_TOUCH 1
#endif
#ifndef LDP_KERNARG_PRELOAD        // 1 needs -mllvm -amdgpu-kernarg-preload-count=14 to have any effect (csrc/Makefile)
#define LDP_KERNARG_PRELOAD 1
#endif

namespace ldp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum : int { MODE_K5 = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_P1 = 3,
             // 3x3 convolutions over NHWC images (StableVAE): a "sample" is one row tile (n, h, w-tile)
             // of TO output pixels, the taps along W are the Toeplitz taps and the three image rows
             // dh = 0..2 are folded into the K loop (virtual input channel = dh * Cin + c)
             MODE_K3H = 4,    // stride 1, pad 1:            in(h + dh - 1, w + dw - 1), TI = TO + 2 (halo)
             MODE_K3S = 5 };  // stride 2, pad (0,1),(0,1):  in(2h + dh, 2w + dw),       TI = 2 TO + 2

enum : int {
  EP_GN = 1,       // GroupNorm(eps 1e-6, one group per work-group tile) + Mish
  EP_FILM = 2,     // out = scale * out + bias, (scale|bias) = film_t[k] + film_g[b]
  EP_RESIN = 4,    // out += res_in
  EP_RELU = 8,     // out = max(out, 0)
  EP_STEP = 16,    // scheduler update of the padded state (DDPM / DDIM), in place
  EP_EPSOUT = 32,  // write eps to an unpadded (B, T, D) tensor
};

struct StepCoef {   // one row of schedule.step_coefficients()
  float t, inv_sqrt_ab, sqrt_1mab, c_x0, c_x, c_eps, sigma, pad;
};

constexpr int KW_MAX = 8;          // most work-groups one K range is split over (ConvArgs::kw)

struct ConvArgs {
  const float* xa;        // (B, TI, ca)
  const float* xb;        // (B, TI, cb) second input of a channel concat, or nullptr
  int ca, cb;
  const float* w;         // packed [chunk][tap][cout/16][64 lanes][4]; RES_OUT: one extra "tap" = the 1x1 projection
  const float* bias;      // (cout)
  const float* bres;      // RES_OUT: (cout)
  float* res_out;         // RES_OUT: (B, TO, cout) = Conv1x1(x) + bres
  const float* gn_scale;  // (cout)
  const float* gn_bias;   // (cout)
  const float* film_t;    // (n_train, film_stride) already offset to this block's slice
  const float* film_g;    // (B, film_stride)       already offset to this block's slice
  int film_stride;
  const int* k_dev;       // (B) per-sample timestep or nullptr
  int k;                  // scalar timestep
  const float* res_in;    // (B, TO, cout)
  float* out;             // (B, TO, cout)
  int B, cout, flags;
  // EP_STEP / EP_EPSOUT (final 1x1 conv of the planner, cout = padded D)
  int d_real;             // D
  int rows_valid;         // number of real (sample, position) rows = B*TO unless the last sample is partial
  StepCoef coef;          // coefficients of this step
  const float* noise;     // (B, TO, D) explicit N(0,1) for this step, or nullptr -> Philox
  const uint64_t* seed;   // device: {seed, first global row (= row_offset * rows per sample)}
  int step;               // executed-step index (Philox stream id)
  float* eps_out;         // (B, TO, D)
  int dbg;                // ablation switches for tools/ (0 in production): 8 no main loop, 16 no epilogue, 32 no stats exchange, 64 empty kernel, 128 no output stores, 256 main loop on cache-hot operands
  // column split of a GroupNorm group over `cs` work-groups (1, 2 or 4): the parts exchange
  // their partial (sum, sum of squares) per sample through 8-byte {value, tag} granules
  int cs;
  unsigned long long* xchg;   // this launch's granule slab: [sample block][group][4 parts][16 samples][2]
  // K split over work-groups (small batches: a few sample blocks leave most CUs idle and each work-group
  // streams its whole weight slice at one CU's load rate).  kw = 1, 2, 4 or 8 work-groups share the input
  // channels of one (sample block, column block); parts 1.. publish their K-partial tiles as 8-byte
  // {value, tag} granules (one untorn store each, tag unique per call, step and launch: no drain, no flag --
  // a granule that is not there yet simply carries an older tag), part 0 polls them, adds them in part
  // order and runs the epilogue.
  int kw;
  unsigned long long* kw_slab;   // [sample block][column block][kw][main tile | projection tile], tile = 16*MB*TO*BN <= 8192 {value, tag} granules
  int kw_slot;                   // index of this launch among the K-split launches of an evaluation (part of the tag)
  const uint64_t* ctl;        // device control words: [0] seed, [1] first global row, [2] call epoch
  unsigned int* fault;        // [0] set to 1 when a peer never answered (bounded spin); [1] set to 1 by the fp16-plane tiles when an operand left the planes' range
  // 2-D modes: B = N * h_out * w_tiles row tiles; input image is (h_in, w_in, ca)
  int h_out, w_tiles, h_in, w_in;
  // 1-D modes: only the first ca_real channels of xa exist (row stride ca_real); the rest of the
  // (zero-weighted) chunk reads as zero.  0 = all of ca is real.
  int ca_real;
  // 1: blockIdx.x = sample block, blockIdx.z = GroupNorm group (work-groups of one sample block share an XCD:
  // its L2 then fetches 1/8 of the activations and all of the layer's weights -- for the layers whose weights
  // are smaller than their activations); 0: blockIdx.x = group (an XCD holds one group's weight columns)
  int by_sample;
  // 2-D modes with 64-column tiles: per work-group (sum, sum of squares) of every output column over its 16 row
  // tiles -> stats_part[(sample block * cout + column) * 2 + {0,1}] (nullptr: off).  The StableVAE's next GroupNorm
  // takes its statistics from these instead of re-reading the tensor it normalises.
  float* stats_part;
  // by_sample launches of the final 1x1 conv over position pairs (MODE_P1): a 16-row block here is 1/2^sb_qs of a
  // 16-sample block of the neighbouring layers (q = T/2 pair rows per sample).  blockIdx.x -> row block such that
  // all q sub-blocks of sample block rb run on XCD rb % 8, where the layer before wrote and the next evaluation's
  // first conv will read those rows (0: identity mapping)
  int sb_qs;
  // tools/timeline.py (library built with -DLDP_TIMELINE only): per-wave s_memtime stamps of this launch,
  // [work-group][16 waves][8 stamps]; nullptr in production
  unsigned long long* tl;
};

// Timing ablations (ConvArgs::dbg; results wrong by construction) exist only in the -DLDP_ABLATE build
// (`make ablate` -> libldp_hip_abl.so, loaded by tools/ through --lib): the product kernels do not carry the switches.
#ifdef LDP_ABLATE
#define LDP_ABL(bit) ((a.dbg & (bit)) != 0)
#else
#define LDP_ABL(bit) (false)
#endif

#ifdef LDP_TIMELINE
#define LDP_TL(i)                                                                                         \
  do {                                                                                                    \
    if (a.tl) {                                                                                           \
      const unsigned long long t_ = (i) == 0 ? ldp_t_entry : __builtin_amdgcn_s_memtime();                \
      if (lane == 0)                                                                                      \
        a.tl[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + wave) * 8 + (i)] = t_; \
    }                                                                                                     \
  } while (0)
#else
#define LDP_TL(i) do { } while (0)
#endif

__host__ __device__ constexpr int mode_taps(int mode) {
  return mode == MODE_K5 ? 5 : (mode == MODE_DOWN || mode == MODE_K3H || mode == MODE_K3S) ? 3
       : mode == MODE_UP ? 4 : 1;
}
__host__ __device__ constexpr bool mode_2d(int mode) { return mode == MODE_K3H || mode == MODE_K3S; }
__host__ __device__ constexpr int mode_ti(int mode, int to) {
  return mode == MODE_DOWN ? 2 * to : mode == MODE_UP ? to / 2 : mode == MODE_K3H ? to + 2
       : mode == MODE_K3S ? 2 * to + 2 : to;
}
// input position read by output position `to` through tap `j`; <0 or >=TI: zero padding
__host__ __device__ constexpr int tap_src(int mode, int to, int j) {
  if (mode == MODE_K5) return to + j - 2;
  if (mode == MODE_DOWN || mode == MODE_K3S) return 2 * to + j;
  if (mode == MODE_K3H) return to + j;
  if (mode == MODE_UP) {
    const int q = to >> 1;
    if ((to & 1) == 0) return j == 0 ? q - 1 : (j == 2 ? q : -1);
    return j == 1 ? q : (j == 3 ? q + 1 : -1);
  }
  return j == 0 ? to : -1;
}
// output position fed by input position `ti` through tap `j` (inverse of tap_src; at most one), or -1
__host__ __device__ constexpr int tap_dst(int mode, int to_n, int ti, int j) {
  for (int to = 0; to < to_n; ++to)
    if (tap_src(mode, to, j) == ti) return to;
  return -1;
}
__host__ __device__ constexpr bool tap_used(int mode, int to_n, int j) {
  const int ti_n = mode_ti(mode, to_n);
  for (int to = 0; to < to_n; ++to) {
    const int ti = tap_src(mode, to, j);
    if (ti >= 0 && ti < ti_n) return true;
  }
  return false;
}
// number of (to, j) pairs that hit real data = MFMAs per k-step per wave
__host__ __device__ constexpr int valid_pairs(int mode, int to_n) {
  const int ti_n = mode_ti(mode, to_n);
  int n = 0;
  for (int to = 0; to < to_n; ++to)
    for (int j = 0; j < mode_taps(mode); ++j) {
      const int ti = tap_src(mode, to, j);
      if (ti >= 0 && ti < ti_n) ++n;
    }
  return n;
}

// 16-byte-slot swizzle of a 16x16 f32 sub-tile: slot' = slot ^ H(row >> 2), H = {0,3,2,1}
__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((4 - (row >> 2)) & 3); }

// x * tanh(softplus(x)) = x * n / (n + 2),  n = e^x (e^x + 2)   (exact algebra, one exp).
// Hardware exp2 / rcp (1 ulp each): the epilogue is issue-bound (a wave64 VALU op holds the SIMD
// for 4 cycles and the epilogue runs once per launch), so IEEE division and libm expf -- ~25
// instructions per element -- cost more than their last ulp is worth at a 1e-4 tolerance.
#ifndef LDP_EPILOGUE_IEEE      // A/B build (make ieee): libm expf, IEEE divide and 1 / sqrtf in the epilogues -- what the hardware exp2 / rcp / rsq cost in accuracy (round 6)
#define LDP_EPILOGUE_IEEE 0
#endif
__device__ __forceinline__ float mish_f(float x) {
#if LDP_EPILOGUE_IEEE
  const float e = expf(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  return x * (n / (n + 2.0f));
#else
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  const float n = e * (e + 2.0f);
  return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
#endif
}

// Sum over the 64 lanes, returned in every lane.  Six DPP adds (row_shr / row_bcast: no LDS
// round trips, unlike ds_bpermute shuffles) leave the total in lane 63; a readlane makes it
// uniform.  Fixed association order -> bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x114, 0xF>(v);     // row_shr 4
  v = dpp_add<0x118, 0xF>(v);     // row_shr 8  -> lane 15 of each row = row sum
  v = dpp_add<0x142, 0xA>(v);     // row_bcast 15 into rows 1, 3
  v = dpp_add<0x143, 0xC>(v);     // row_bcast 31 into rows 2, 3 -> lane 63 = total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Exchange granules: 8 bytes {value bits, tag ^ value bits}.  A consumer accepts a granule only if its two halves
// agree on the tag, so value and tag travel in one naturally aligned 8-byte store and a granule that is stale (an
// earlier step or call), zeroed (tags are never 0), or -- were it ever to happen -- half-written reads as "not there
// yet".  The granule alone is not enough: the consumer must also wait for ALL its poll loads (vmcnt(0)) before it
// validates any of them, see the statistics exchange below and DESIGN.md section 4.5.
__device__ __forceinline__ unsigned long long granule_pack(unsigned int tag, float v) {
  const unsigned int b = __float_as_uint(v);
  return ((unsigned long long)(tag ^ b) << 32) | b;
}
// 16 bytes = the two granules {sum, sum of squares} of one sample, moved by ONE sc1 (agent-visible: write-through /
// L1-bypassing) buffer access: each 8-byte half is untorn (observed on gfx950) and validates itself by its tag.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;      // cache-policy bit of the raw-buffer builtins on gfx940+

__device__ __forceinline__ bool granule_ok(unsigned long long g, unsigned int tag) {
  return ((unsigned int)(g >> 32) ^ (unsigned int)g) == tag;
}

// Philox4x32-10 (Salmon et al., SC'11; Random123 `philox4x32_R(10, ctr, key)`): counter
// (c0,c1,c2,c3) = (elem lo, elem hi, step, stream), key (k0,k1) = (seed lo, seed hi).
// tests/test_philox.py checks the raw words against the Random123 known-answer vectors.
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t elem, uint32_t step, uint32_t stream,
                                              uint32_t (&out)[4]) {
  uint32_t c0 = (uint32_t)elem, c1 = (uint32_t)(elem >> 32), c2 = step, c3 = stream;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one N(0,1): Box-Muller on the first two words, u = (top 24 bits + 0.5) / 2^24 in (0,1)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t elem, uint32_t step,
                                               uint32_t stream) {
  uint32_t w[4];
  philox4x32_10(seed, elem, step, stream, w);
  const float u1 = ((float)(w[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u2 = ((float)(w[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// epilogue features a mode can be asked for / always has: what is not listed is compiled out
__host__ __device__ constexpr int mode_flag_mask(int mode) {
  return mode == MODE_K5 ? (EP_GN | EP_FILM | EP_RESIN)
       : mode == MODE_P1 ? (EP_RESIN | EP_RELU | EP_STEP | EP_EPSOUT)
       : mode_2d(mode) ? EP_RESIN : 0;
}
__host__ __device__ constexpr int mode_flag_forced(int mode) { return mode == MODE_K5 ? EP_GN : 0; }

// MB = 16-sample row blocks per work-group: with MB = 2 every weight fragment fetched from L2 feeds
// twice the MFMAs (the T <= 4 layers are bound by the weight stream, not by the matrix pipe)
// SPLIT (round 4, planner layers at >= 512 plans): the main loop runs on v_mfma_f32_32x32x16_bf16 with three-plane split operands
// (sconv.hpp: x = h + m + l exactly, six plane products, fp32 accumulate).  A wave then owns 32 samples (MB = 2) x 32 columns (two
// 16-column blocks: NWN / 2 column waves), K granularity stays the 16-channel sub-chunk; the activations are split once, on their way
// into LDS; weights are packed as planes at finalize.  Only the accumulators -> LDS step of the epilogue knows the tile shape.
// SPLIT with MB = 1: v_mfma_f32_16x16x32_bf16 on the fp32 kernel's own wave tile (16 samples x 16 columns x TO positions, the same
// accumulator layout, so the epilogue is untouched); one matrix instruction spans two 16-channel sub-chunks (CPI even).  Serves the
// tiles whose 32-sample form does not fit the register file: T = 8 (eight accumulators) and the T = 4 convs that carry the projection.
// SPLIT = 2: the 16-row matrix instruction over MB = 2 row blocks per wave -- every weight fragment a wave pulls through L2 feeds
// 32 samples (the 16-row tiles are bound by their weight stream and the LDS stage + barrier per step); accumulators,
// K order and epilogue are those of the MB = 1 form, so a plan's values do not depend on which of the two served it.
template <int MODE, int TO, int NWN, int KS, int CPI, int MB = 1, int SPLIT = 0>
struct TConvCfg {
  static constexpr int TI = mode_ti(MODE, TO);
  static constexpr int NJ = mode_taps(MODE);
  static constexpr bool S32 = (SPLIT == 1 || SPLIT == 4) && MB == 2;    // v_mfma_f32_32x32x16_{bf16,f16}: a wave owns 32 samples x 32 columns
  static constexpr bool S16 = (SPLIT == 1 && MB == 1) || SPLIT == 2 || SPLIT == 3;   // v_mfma_f32_16x16x32_{bf16,f16}: the fp32 kernel's wave tile(s), K in 32-channel steps
  // SPLIT = 3: the 16-row tile (MB = 1 or 2) on TWO fp16 planes per operand and THREE products (v_mfma_f32_16x16x32_f16): x ~ h + l' / 2^11 with
  // h = fp16(x), l' = fp16((x - h) * 2^11) (the scaling keeps l' a normal fp16 with its 11 bits whatever |x|): 22 significand bits, exact
  // products; h h goes to one accumulator, h l' + l' h to a second one that is added with weight 2^-11 behind the K loop; the l' l' term
  // (2^-22 |ab|) is dropped.  Half the matrix instructions and two thirds of the operand bytes of the six-product bf16 form.
  static constexpr bool F16 = SPLIT == 3 || SPLIT == 4;      // (SPLIT = 4: the 32-row tile on fp16 planes)
  static constexpr int NPL = F16 ? 2 : 3;               // operand planes
  static constexpr int NWC = S32 ? NWN / 2 : NWN;       // waves along the columns
  static constexpr int NW = NWC * KS;
  static constexpr int NT = 64 * NW;
  static constexpr int BN = 16 * NWN;
  static constexpr int BNP = BN + 4;                    // padded row of the epilogue tile
  static constexpr int NC = KS * CPI;                   // 16-channel sub-chunks per iteration
  static constexpr int CH_IT = 16 * NC;                 // input channels per iteration
  static constexpr int XT = MB * TI * NC * 256 * (SPLIT ? NPL : 2) / 2; // floats per staged X buffer (split: three bf16 planes = 6 B per element, two fp16 planes = 4 B)
  static constexpr int NLD = (MB * TI * NC * 64) / NT;  // float4 staging loads per thread
  static constexpr int EPI = MB * KS * TO * 16 * BNP;   // floats of the epilogue tile
  static constexpr int TILE_FLOATS = (2 * XT > EPI) ? 2 * XT : EPI;
  static constexpr bool STATS = MODE == MODE_K3H && BN == 64 && MB == 1 && (NW == 8 || NW == 4);   // ConvArgs::stats_part supported
  static constexpr int LDS_FLOATS = TILE_FLOATS + (STATS ? 2 * NW * 64 : 0);
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static constexpr int EPL = (TO * BN) / 64;            // elements per lane per sample
  static_assert((MB * TI * NC * 64) % NT == 0, "staging loads must divide evenly");
  static_assert((TO * BN) % 64 == 0, "epilogue needs TO*BN multiple of 64");
  static_assert(NW <= 16, "at most 16 waves");
  static_assert(!SPLIT || (S32 && MODE == MODE_K5 && NWN % 2 == 0) || (S16 && (MODE == MODE_K5 || MODE == MODE_DOWN || MODE == MODE_UP || ((MODE == MODE_K3S || MODE == MODE_K3H) && SPLIT == 3 && MB == 1)) && MB <= 2 && CPI % 2 == 0),
                "split operands: 32-sample x 32-column wave tiles (SPLIT = 1, MB = 2, k = 5) or 16 x 16 tiles over 32-channel steps (MB = 1, or SPLIT = 2 with MB = 2; k = 5, stride-2, transposed)");
};

// KWS: compiled with the K-split-over-work-groups path (small-batch plans only: the epilogue is issue-bound,
// the B >= 129 instantiations do not carry its instructions)
// Kernel-argument preload (LDP_KERNARG_PRELOAD, built with -mllvm -amdgpu-kernarg-preload-count=14): the fields the
// first global loads depend on travel as leading scalar arguments, which gfx950 delivers in SGPRs with the wave
// launch; the struct behind them (a by-reference aggregate is never preloaded) is fetched while those loads fly.
struct ConvHot {
  // zfold: more than 32768 sample blocks, folded into blockIdx.y (the only case that needs gridDim.z: an implicit
  // argument at the far end of the kernarg segment, i.e. one more scalar-cache miss in front of the first load)
  static __host__ __device__ int pack(const ConvArgs& a, bool zfold) {
    return (a.cs & 15) | ((a.kw & 15) << 4) | ((a.by_sample & 1) << 8) | ((a.sb_qs & 15) << 9) | ((zfold ? 1 : 0) << 13) | (((a.by_sample >> 1) & 1) << 14);
  }
};
#if LDP_KERNARG_PRELOAD
#define LDP_KERNEL_PARAMS const float* h_xa, const float* h_xb, const float* h_w, int h_B, int h_ca, int h_cb, int h_cout, \
                          int h_ca_real, int h_pk, int h_dbg, const ConvArgs a_in
#define LDP_KERNEL_ARGS(a, zfold) (a).xa, (a).xb, (a).w, (a).B, (a).ca, (a).cb, (a).cout, (a).ca_real, ConvHot::pack(a, zfold), (a).dbg, (a)
#else
#define LDP_KERNEL_PARAMS const ConvArgs a
#define LDP_KERNEL_ARGS(a, zfold) (a)
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// fp32 -> three bf16 planes, four values at a time: (h, m, l) as pairs of packed words
__device__ __forceinline__ void split4(const f32x4 v, uint2& h, uint2& m, uint2& l) {
  unsigned short hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 bh = (__bf16)v[i];
    const float r1 = v[i] - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    hh[i] = __builtin_bit_cast(unsigned short, bh); mm[i] = __builtin_bit_cast(unsigned short, bm); ll[i] = __builtin_bit_cast(unsigned short, bl);
  }
  h = uint2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16)};
  m = uint2{(unsigned)mm[0] | ((unsigned)mm[1] << 16), (unsigned)mm[2] | ((unsigned)mm[3] << 16)};
  l = uint2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16)};
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// fp32 -> two fp16 planes, four values at a time: h = fp16(x), l' = fp16((x - h) * 2^11)
__device__ __forceinline__ void split4h(const f32x4 v, uint2& h, uint2& l) {
  unsigned short hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 bh = (_Float16)v[i];
    const _Float16 bl = (_Float16)((v[i] - (float)bh) * 2048.0f);
    hh[i] = __builtin_bit_cast(unsigned short, bh); ll[i] = __builtin_bit_cast(unsigned short, bl);
  }
  h = uint2{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16)};
  l = uint2{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16)};
}

template <int MODE, int TO, int NWN, int KS, int CPI, bool RES_OUT, int MB = 1, bool KWS = false, int SPLIT = 0>
__global__ __launch_bounds__(64 * (((SPLIT == 1 || SPLIT == 4) && MB == 2) ? NWN / 2 : NWN) * KS,
                              // four-wave fp16-plane tiles: two waves per SIMD, i.e. at most 256 registers -- two work-groups per CU as the bf16 form had
                              // (the second accumulator set pushed them to 268 .. 440 registers, one work-group of four waves per CU)
                              // -- not the T = 8 tiles with the projection: 218 .. 262 bytes of spills under that limit)
                              // (round 5: the same for the eight-wave forms of those tiles -- twice the K slices -- that serve 257 .. 512 plans, one work-group per CU)
                              (SPLIT == 3 && (NWN * KS == 4 || (NWN * KS == 8 && KS >= 2 && NWN <= 4 && (!RES_OUT || TO == 2)) || (NWN == 2 && KS == 4)) && TO <= 8 && !(TO == 8 && RES_OUT)) ? LDP_F16_MINW : 1) void tconv_kernel(LDP_KERNEL_PARAMS) {
#if LDP_KERNARG_PRELOAD
  ConvArgs a = a_in;
  a.xa = h_xa; a.xb = h_xb; a.w = h_w; a.B = h_B; a.ca = h_ca; a.cb = h_cb; a.cout = h_cout; a.ca_real = h_ca_real;
  a.cs = h_pk & 15; a.kw = (h_pk >> 4) & 15; a.by_sample = (h_pk >> 8) & 1; a.sb_qs = (h_pk >> 9) & 15; a.dbg = h_dbg;
#endif
  using C = TConvCfg<MODE, TO, NWN, KS, CPI, MB, SPLIT>;
  constexpr int TI = C::TI, NJ = C::NJ, NC = C::NC, NT = C::NT, BN = C::BN, BNP = C::BNP;
  static_assert(!RES_OUT || MODE == MODE_K5, "RES_OUT only for k=5 convs");
  static_assert(!SPLIT || !KWS, "split operands: no K split over work-groups");
  constexpr bool S32 = C::S32, S16 = C::S16;

  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);

#ifdef LDP_TIMELINE
  const unsigned long long ldp_t_entry = __builtin_amdgcn_s_memtime();      // before any kernel argument is asked for
#endif
#if LDP_KERNARG_TOUCH && !LDP_KERNARG_PRELOAD
  // The 320-byte argument block spans five 64-byte lines of the kernarg segment, cold in the scalar cache at every
  // launch.  Left alone the compiler fetches fields where it first needs them: three batches of s_load, each waited
  // for before the next is issued -- three serialized misses in front of the first global load.  One field of every
  // line is demanded here, so all five lines are requested together (later s_loads hit the scalar cache).
  asm volatile("" ::"s"(a.xa), "s"(a.gn_bias), "s"(a.B), "s"(a.seed), "s"(a.ctl));
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: per-wave indices and branches go scalar
  const int wn = wave % C::NWC, ks = wave / C::NWC;
  // block -> (group g, half h, sample block sb).  blockIdx % ngroups = g, so (observed dispatch:
  // block b runs on XCD b % 8) all sample blocks and both halves of a GroupNorm group share one
  // XCD's L2, which then holds only that group's weight columns.  Speed only, never correctness.
  // grid = (groups, cs * zf, sample blocks / zf): no integer division in the prologue
  const int cs = a.cs > 1 ? a.cs : 1;
  const int ngroups = a.by_sample ? gridDim.z : gridDim.x;
  int grp = a.by_sample ? blockIdx.z : blockIdx.x;
  // blockIdx.y = half + cs * (kpart + kw * zf-index)
  const int kw = (KWS && a.kw > 1) ? a.kw : 1;
  const int half = blockIdx.y & (cs - 1);
  const int kpart = (blockIdx.y >> (cs >> 1)) & (kw - 1);
#if LDP_KERNARG_PRELOAD
  int sb = a.by_sample ? blockIdx.x : blockIdx.z;
  if ((h_pk >> 13) & 1) sb += gridDim.z * (blockIdx.y >> ((cs >> 1) + (__ffs(kw) - 1)));
#else
  int sb = a.by_sample ? blockIdx.x : blockIdx.z + gridDim.z * (blockIdx.y >> ((cs >> 1) + (__ffs(kw) - 1)));
#endif
  if (MODE == MODE_P1 && a.by_sample && a.sb_qs > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    sb = ((((j >> a.sb_qs) << 3) + xcd) << a.sb_qs) + (j & ((1 << a.sb_qs) - 1));
  }
#if LDP_KERNARG_PRELOAD
  if constexpr (SPLIT != 0) {
    // Two-dimensional placement of a group-major launch (8 groups = 8 XCDs, sample blocks a multiple of 4; launch_one checks): XCD X works on
    // groups X >> 2, + 2, + 4, + 6 and on the sample blocks congruent to X & 3 mod 4 -- half of the weight columns and a quarter of the
    // activations per L2 (fabric: 4 x weights + 2 x activations instead of 1 x + 8 x).  Both halves of a (group, sample block) stay on one XCD.
    if ((h_pk >> 14) & 1) {
      const int X = grp, si = sb;
      grp = (X >> 2) + 2 * (si & 3);
      sb = (X & 3) + 4 * (si >> 2);
    }
  }
#endif
  if (sb * (16 * MB) >= a.B) return;
  const int cbk = grp * cs + half;
  const int b0 = sb * (16 * MB);
  const int r = lane & 15, kq = lane >> 4;
  const int nblk_total = S32 ? a.cout >> 5 : a.cout >> 4;        // weight fragments are 16 (32-row split tiles: 32) columns wide
  const int nblk = cbk * C::NWC + wn;
  const int cin = a.ca + a.cb;
  int nit_all = LDP_ABL(8) ? 0 : (mode_2d(MODE) ? 3 * cin : cin) / C::CH_IT;
  // narrow first layer: the virtual channels behind the ca_real stored ones carry exact-zero weights -- the K range ends with the last
  // iteration that holds a stored channel (round 5: the tiles of > 256 plans ran 2 (T = 8) and 4 (T = 16) iterations, all but the first on zeros)
  if constexpr (MODE == MODE_K5 && RES_OUT && SPLIT == 0 && !KWS) {
    if (a.ca_real > 0 && !LDP_ABL(8)) nit_all = (a.ca_real + C::CH_IT - 1) / C::CH_IT;
  }
  const int it0 = kpart * (nit_all / kw);            // this work-group's K range: iterations [it0, nit)
  const int nit = it0 + nit_all / kw;
  if LDP_ABL(64) return;
  LDP_TL(0);

  f32x4 acc[MB][TO];
  f32x4 racc[MB][RES_OUT ? TO : 1];
  f32x4 acc_lo[C::F16 ? MB : 1][C::F16 ? TO : 1], racc_lo[C::F16 ? MB : 1][C::F16 && RES_OUT ? TO : 1];    // fp16 planes: the h l' + l' h products (x 2^11)
#pragma unroll
  for (int m = 0; m < (C::F16 ? MB : 1); ++m) {
#pragma unroll
    for (int t = 0; t < (C::F16 ? TO : 1); ++t) acc_lo[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (C::F16 && RES_OUT ? TO : 1); ++t) racc_lo[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int t = 0; t < TO; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (RES_OUT ? TO : 1); ++t) racc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- per-thread staging coordinates (fixed across iterations) -------------------------
  int st_goff[C::NLD];   // offset (floats) of the float4 within the (B,TI,C) tensor, minus c0
  int st_loff[C::NLD];   // offset (floats) in the LDS buffer
  int st_cc[C::NLD];
  int st_mask[mode_2d(MODE) ? C::NLD : 1];
#pragma unroll
  for (int i = 0; i < C::NLD; ++i) {
    const int idx = tid + i * NT;
    const int q = idx & 3;
    const int cc = (idx >> 2) % NC;
    const int rr = (idx / (4 * NC)) & 15;
    const int tm = idx / (64 * NC);
    const int tt = tm % TI, mb = tm / TI;
    st_cc[i] = cc * 16 + q * 4;
    // rows of samples beyond B are clamped to the last sample: they compute garbage that is
    // never stored, and the staging loads need no predication (keeps the loop one basic block)
    const int bb = (b0 + mb * 16 + rr) < a.B ? (b0 + mb * 16 + rr) : (a.B - 1);
    st_goff[i] = bb * TI + tt;                           // row index; multiplied by C later
    st_loff[i] = (((mb * TI + tt) * NC + cc) * 16 + rr) * 16 + swz(rr, q) * 4;
    // split: [position][sub-chunk][plane][k half][32 samples] units of 16 B (8 channels): the A fragment of
    // v_mfma_f32_32x32x16_bf16 is lane-linear; this thread's 4 channels are half a unit
    if constexpr (S32) st_loff[i] = (((tt * NC + cc) * C::NPL) * 64 + (q >> 1) * 32 + mb * 16 + rr) * 4 + (q & 1) * 2;
    // 16-row split tiles: [position][32-channel step][plane][k quarter][16 samples]: the A fragment of v_mfma_f32_16x16x32_bf16
    // (lane = 16 * (k / 8) + row); this thread's 4 channels are half a unit of k quarter 2 (cc & 1) + (q >> 1)
    // (SPLIT = 2: one such image per row block)
    if constexpr (S16) st_loff[i] = ((((mb * TI + tt) * (NC / 2) + (cc >> 1)) * C::NPL) * 64 + ((cc & 1) * 2 + (q >> 1)) * 16 + rr) * 4 + (q & 1) * 2;
    if (mode_2d(MODE)) {
      // row tile bb = (n, h, wt); st_goff = input pixel index for dh = 0, st_mask bit dh = that
      // pixel lies inside the image (zero padding otherwise, applied after the load)
      const int wt = bb % a.w_tiles, hh = (bb / a.w_tiles) % a.h_out, n = bb / (a.w_tiles * a.h_out);
      const int w = MODE == MODE_K3H ? wt * TO - 1 + tt : 2 * wt * TO + tt;
      const int h0 = MODE == MODE_K3H ? hh - 1 : 2 * hh;
      const bool wok = w >= 0 && w < a.w_in;
      int m = 0;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) m |= (wok && h0 + dh >= 0 && h0 + dh < a.h_in) ? (1 << dh) : 0;
      st_mask[i] = m;
      st_goff[i] = (n * a.h_in + h0) * a.w_in + w;
    }
  }

  f32x4 xst[C::NLD];
  auto stage_load = [&](int it) {
    const int c0 = it * C::CH_IT;
    if (mode_2d(MODE)) {
      const int dh = c0 / a.ca, cbase = c0 - dh * a.ca;
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        const bool ok = (st_mask[i] >> dh) & 1;
        const int pix = ok ? st_goff[i] + dh * a.w_in : 0;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.xa + (size_t)pix * a.ca + cbase + st_cc[i]);
        xst[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      return;
    }
    const bool second = (c0 >= a.ca);
    const float* base = second ? a.xb : a.xa;
    const int cw = second ? a.cb : a.ca;
    const int cbase = second ? c0 - a.ca : c0;
    // narrow first layer (ca_real > 0): only ca_real channels exist (row stride ca_real); the virtual channels
    // beyond them re-read the row's first channels.  Their weights are packed as exact zeros, so whatever finite
    // value arrives contributes +-0 -- the loaded value is NOT masked: a select on it is a VALU op the scheduler
    // is free to place right behind the load, and it did, with a full s_waitcnt vmcnt(0) in front of every one of
    // them (round-3 finding: each staging load's latency was exposed in the middle of the MFMA stream).
    const int creal = a.ca_real > 0 ? a.ca_real : 0x7fffffff;
    const int stride = a.ca_real > 0 ? a.ca_real : cw;
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const bool ok = (cbase + st_cc[i]) < creal;
      xst[i] = *reinterpret_cast<const f32x4*>(base + (size_t)st_goff[i] * stride + (ok ? cbase + st_cc[i] : 0));
    }
  };
  auto stage_store = [&](float* buf) {
    if constexpr (C::F16) {
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        uint2 ph, pl;
        split4h(xst[i], ph, pl);
        *reinterpret_cast<uint2*>(buf + st_loff[i]) = ph;
        *reinterpret_cast<uint2*>(buf + st_loff[i] + 256) = pl;
      }
      return;
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        uint2 ph, pm, pl;
        split4(xst[i], ph, pm, pl);
        *reinterpret_cast<uint2*>(buf + st_loff[i]) = ph;
        *reinterpret_cast<uint2*>(buf + st_loff[i] + 256) = pm;
        *reinterpret_cast<uint2*>(buf + st_loff[i] + 512) = pl;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) *reinterpret_cast<f32x4*>(buf + st_loff[i]) = xst[i];
  };

  // ---- weight fragment streaming ---------------------------------------------------------
  // Two register buffers with *explicitly* swapped roles (the loop is unrolled by two): with a
  // single loop body and a `cur = next` copy the register coalescer merges the two buffers,
  // which forces every load behind the last MFMA that reads its destination and destroys the
  // prefetch distance.
  // The projection's fragments live in the conv's own buffer, right behind the taps of their chunk: a
  // second weight stream from a separate allocation cost 20-25 % of the loop time of these layers.
  constexpr bool SKIPZ = MODE == MODE_K5 && RES_OUT && SPLIT == 0 && TO >= 8 && KS > 1;
  constexpr int NJW = NJ + (RES_OUT ? 1 : 0);
  // 16-row split tiles: an iteration is NSTEP 32-channel steps; a register buffer holds ONE step and the two buffers roll step by
  // step (the LDS stage + barrier of an iteration then amortises over NSTEP steps without more weight registers)
  constexpr int NSTEP = S16 ? CPI / 2 : 1;
  constexpr int WCH = S16 ? 1 : CPI;                     // weight fragments along K held by one register buffer
  constexpr int RN = RES_OUT ? WCH : 1;
  constexpr int WPL = SPLIT ? C::NPL : 1;                // weight planes
  f32x4 wb0[NJ][WCH * WPL], wb1[NJ][WCH * WPL];
  f32x4 rb0[RN * WPL], rb1[RN * WPL];
  // S16: `it` counts 32-channel steps of this wave's K slice (iteration * NSTEP + step)
  auto wload = [&](int it, f32x4 (&b)[NJ][WCH * WPL], f32x4 (&rb)[RN * WPL]) {
#pragma unroll
    for (int ci = 0; ci < WCH; ++ci) {
      const int gc = S16 ? (it / NSTEP) * (NC / 2) + ks * NSTEP + (it % NSTEP) : it * NC + ks * CPI + ci;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (SPLIT) {
          if (tap_used(MODE, TO, j)) {
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) {
              const size_t off = ((((size_t)gc * NJW + j) * nblk_total + nblk) * WPL + pl) * 256 + lane * 4;
              b[j][ci * WPL + pl] = *reinterpret_cast<const f32x4*>(a.w + off);
            }
          }
        } else
        if (tap_used(MODE, TO, j)) {
          const size_t off = (((size_t)gc * NJW + j) * nblk_total + nblk) * 256 + lane * 4;
          // LDP_W_NT (A/B build only, round 4): non-temporal weight loads in the small-batch (KWS) instantiations, where one
          // CU reads its weight slice once per launch (MI355X_MICROARCH.md row nt-weights); measured, see DESIGN 4.1
          b[j][ci] = (LDP_W_NT && KWS) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.w + off))
                                       : *reinterpret_cast<const f32x4*>(a.w + off);
        }
      }
      if constexpr (RES_OUT && SPLIT) {
#pragma unroll
        for (int pl = 0; pl < WPL; ++pl) {
          const size_t off = ((((size_t)gc * NJW + NJ) * nblk_total + nblk) * WPL + pl) * 256 + lane * 4;
          rb[ci * WPL + pl] = *reinterpret_cast<const f32x4*>(a.w + off);
        }
      } else
      if (RES_OUT) {
        const size_t off = (((size_t)gc * NJW + NJ) * nblk_total + nblk) * 256 + lane * 4;
        rb[ci] = (LDP_W_NT && KWS) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.w + off))
                                   : *reinterpret_cast<const f32x4*>(a.w + off);
      }
    }
  };

  // one iteration = CH_IT input channels; bc/rc hold its weights, bl/rl receive the next one's
  // LAST: the final iteration of the K range has nothing to prefetch: its own copy of the body carries no
  // loads, no LDS writes and no wait for either (round 3; the branch-free version re-requested its own chunk
  // and waited for it before the closing barrier)
  f32x16 acc32[S32 ? TO : 1], racc32[S32 && RES_OUT ? TO : 1];
  f32x16 acc32_lo[S32 && C::F16 ? TO : 1], racc32_lo[S32 && C::F16 && RES_OUT ? TO : 1];      // fp16 planes: the h l' + l' h products (x 2^11)
  if constexpr (S32 && C::F16) {
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc32_lo[t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < (RES_OUT ? TO : 1); ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) racc32_lo[t][i] = 0.f;
  }
  if constexpr (S32) {
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc32[t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < (RES_OUT ? TO : 1); ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) racc32[t][i] = 0.f;
  }
  auto iteration = [&](auto last_tag, int it, f32x4 (&bc)[NJ][WCH * WPL], f32x4 (&rc)[RN * WPL], f32x4 (&bl)[NJ][WCH * WPL],
                       f32x4 (&rl)[RN * WPL]) {
    constexpr bool LAST = decltype(last_tag)::value;
    float* xcur = smem + (it & 1) * C::XT;
    float* xnext = smem + ((it + 1) & 1) * C::XT;
    // branch-free body so that loads, LDS traffic and MFMAs share one scheduling region and can be
    // interleaved below
    const int itn = LDP_ABL(256) ? it0 : (LAST || (it + 1) < nit) ? it + 1 : it;     // dbg 256: every iteration re-requests the first chunk (cache-hot operands)
    if (!LAST) {
      // ablations of the split tiles only (the fp32 instantiations of the ablation build keep one straight-line body): 2048 no staging
      // loads, 1024 no weight loads, 512 no LDS writes and no barrier
      if (!(SPLIT && LDP_ABL(2048))) stage_load(itn);
      if (!S16 && !(SPLIT && LDP_ABL(1024))) wload(itn, bl, rl);
    }
    if constexpr (S16) {
      // position-major: the three planes of position ti are read right ahead of the (tap, output position) pairs that use
      // them, so that a window of positions is live instead of all TI (TO = 8: 96 registers of fragments otherwise)
      constexpr int NPAIR = valid_pairs(MODE, TO) + (RES_OUT ? TO : 0);
      constexpr int NUSED = (tap_used(MODE, TO, 0) ? 1 : 0) + (NJ > 1 && tap_used(MODE, TO, 1) ? 1 : 0) + (NJ > 2 && tap_used(MODE, TO, 2) ? 1 : 0) +
                            (NJ > 3 && tap_used(MODE, TO, 3) ? 1 : 0) + (NJ > 4 && tap_used(MODE, TO, 4) ? 1 : 0);
      constexpr int NPL = C::NPL;
      constexpr int NWL = (NUSED + (RES_OUT ? 1 : 0)) * NPL; // weight loads of one step
      constexpr int NMFMA = (C::F16 ? 3 : LDP_SPLIT_NPROD) * NPAIR * MB;
      constexpr int NREAD = MB * TI * NPL;
      constexpr int AHEAD = MB * (TI > 2 ? 2 : TI) * NPL;    // fragment reads issued before the first matrix instruction (two positions)
      auto step = [&](auto pc_tag, f32x4 (&wc)[NJ][WCH * WPL], f32x4 (&rcur)[RN * WPL], f32x4 (&wn)[NJ][WCH * WPL], f32x4 (&rnext)[RN * WPL]) {
        constexpr int pc = decltype(pc_tag)::value;
        constexpr bool PREF = pc + 1 < NSTEP || !LAST;      // a step follows this one (in this iteration or the next)
        if (PREF && !LDP_ABL(1024)) wload(pc + 1 < NSTEP ? it * NSTEP + pc + 1 : itn * NSTEP, wn, rnext);
        f32x4 asp[MB][TI][NPL];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
              asp[m][ti][pl] = *reinterpret_cast<const f32x4*>(xcur + (((((m * TI + ti) * (NC / 2) + ks * NSTEP + pc) * NPL + pl) * 64 + lane) * 4));
        }
        if constexpr (C::F16) {
          // three products per (position, tap): h l' and l' h into the low accumulator, h h into the main one
          constexpr int FA[3] = {1, 0, 0}, FB[3] = {0, 1, 0};
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) {
#pragma unroll
              for (int j = 0; j <= NJ; ++j) {
                if (j == NJ && !RES_OUT) continue;
                const int to = j == NJ ? (ti < TO ? ti : -1) : tap_dst(MODE, TO, ti, j < NJ ? j : 0);
                if (to < 0) continue;
                const f32x4* bp = j == NJ ? &rcur[0] : &wc[j < NJ ? j : 0][0];
                const f16x8_t bv = __builtin_bit_cast(f16x8_t, bp[FB[pi]]);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                  const f16x8_t av = __builtin_bit_cast(f16x8_t, asp[m][ti][FA[pi]]);
                  f32x4& dst = j == NJ ? (pi < 2 ? racc_lo[m][RES_OUT ? to : 0] : racc[m][RES_OUT ? to : 0]) : (pi < 2 ? acc_lo[m][to] : acc[m][to]);
                  dst = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, dst, 0, 0, 0);
                }
              }
            }
          }
        } else {
        // Product-major inside a position: the (tap, output position) pairs fed by position ti write DIFFERENT accumulators, so
        // consecutive matrix instructions are independent (a chain of six on one accumulator issues every ~28 cycles instead of
        // 16: the launch's time was proportional to the instruction count at 55 % pipe-busy).  Every accumulator still receives
        // its products in the same order -- small first (sconv.hpp), positions ascending: results are bit-identical to the
        // pair-major order.
        constexpr int NP = LDP_SPLIT_NPROD;
        // plane indices (0 = h, 1 = m, 2 = l) of product p, smallest magnitude first
        constexpr int PA9[9] = {2, 1, 2, 1, 2, 0, 1, 0, 0}, PB9[9] = {2, 2, 1, 1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
          for (int pi = 9 - NP; pi < 9; ++pi) {
#pragma unroll
            for (int j = 0; j <= NJ; ++j) {
              if (j == NJ && !RES_OUT) continue;                      // j = NJ: the block's 1x1 projection, position ti -> ti
              const int to = j == NJ ? (ti < TO ? ti : -1) : tap_dst(MODE, TO, ti, j < NJ ? j : 0);      // k = 5: to = ti + 2 - j; stride-2 / transposed: their tap sets
              if (to < 0) continue;
              const f32x4* bp = j == NJ ? &rcur[0] : &wc[j < NJ ? j : 0][0];
              const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, bp[PB9[pi]]);
#pragma unroll
              for (int m = 0; m < MB; ++m) {                          // the row blocks share the weight fragment
                const bf16x8_t av = __builtin_bit_cast(bf16x8_t, asp[m][ti][PA9[pi]]);
                if (j == NJ) racc[m][RES_OUT ? to : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, racc[m][RES_OUT ? to : 0], 0, 0, 0);
                else acc[m][to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[m][to], 0, 0, 0);
              }
            }
          }
        }
        }
        // order template of one step: AHEAD fragment reads, then the remaining reads spread evenly over the matrix instructions
        // (one read keeps about two positions ahead of its use) and the global loads (the next step's weight planes; in the first
        // step also the next iteration's staging loads) over the first LDP_S16_LF percent of them: position-major order needs taps
        // 0..2 at the very start of a step, so a load issued late in the previous step has no prefetch distance
        {
          constexpr int NLOADS = (PREF ? NWL : 0) + ((pc == 0 && !LAST) ? C::NLD : 0);
          constexpr int NRD = NREAD - AHEAD;
          constexpr int ML = NMFMA * LDP_S16_LF / 100 > 0 ? NMFMA * LDP_S16_LF / 100 : 1;       // matrix instructions the loads spread over
#pragma unroll
          for (int i = 0; i < AHEAD; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
          for (int m = 0; m < NMFMA; ++m) {
            if (NLOADS > 0 && m < ML) {
              const int nl = (m + 1) * NLOADS / ML - m * NLOADS / ML;
              if (nl > 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              if (nl > 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              if (nl > 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if (NRD > 0 && (m + 1) * NRD / NMFMA != m * NRD / NMFMA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      };
      // the buffers roll: even steps multiply from bc and fetch into bl, odd steps the other way round
      step(std::integral_constant<int, 0>{}, bc, rc, bl, rl);
      if constexpr (NSTEP > 1) step(std::integral_constant<int, 1>{}, bl, rl, bc, rc);
      if constexpr (NSTEP > 2) step(std::integral_constant<int, 2>{}, bc, rc, bl, rl);
      if constexpr (NSTEP > 3) step(std::integral_constant<int, 3>{}, bl, rl, bc, rc);
      static_assert(NSTEP <= 4, "at most four 32-channel steps per iteration");
      if (LDP_ABL(512)) return;
      if (!LAST) stage_store(xnext);
      __syncthreads();
      return;
    }
    if constexpr (S32) {
      // the planes of every (position, sub-chunk): lane-linear 16-byte units
      constexpr int NPL = C::NPL;
      f32x4 asp[TI][CPI][NPL];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int ci = 0; ci < CPI; ++ci)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            asp[ti][ci][pl] = *reinterpret_cast<const f32x4*>(xcur + ((((ti * NC + ks * CPI + ci) * NPL + pl) * 64 + lane) * 4));
      // one (position, tap) pair: six bf16 plane products small first (sconv.hpp) into one accumulator, or -- fp16 planes -- l' h and h l' into
      // the low accumulator and h h into the main one
      auto pair = [&](const f32x4* ap, const f32x4* bp, f32x16& c, f32x16& clo) {
        if constexpr (C::F16) {
          const f16x8_t ah = __builtin_bit_cast(f16x8_t, ap[0]), al = __builtin_bit_cast(f16x8_t, ap[1]);
          const f16x8_t bh = __builtin_bit_cast(f16x8_t, bp[0]), bl2 = __builtin_bit_cast(f16x8_t, bp[1]);
          clo = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, clo, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
          clo = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl2, clo, 0, 0, 0);
        } else {
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ap[0]), am = __builtin_bit_cast(bf16x8_t, ap[1]), al = __builtin_bit_cast(bf16x8_t, ap[NPL - 1]);
          const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, bp[0]), bm = __builtin_bit_cast(bf16x8_t, bp[1]), bl2 = __builtin_bit_cast(bf16x8_t, bp[NPL - 1]);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        }
      };
#pragma unroll
      for (int ci = 0; ci < CPI; ++ci)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            const int ti = tap_src(MODE, to, j);
            if (ti >= 0 && ti < TI) pair(&asp[ti][ci][0], &bc[j][ci * WPL], acc32[to], acc32_lo[C::F16 ? to : 0]);
          }
      if constexpr (RES_OUT) {                        // the block's 1x1 residual projection: the sixth "tap", position to -> to
#pragma unroll
        for (int ci = 0; ci < CPI; ++ci)
#pragma unroll
          for (int to = 0; to < TO; ++to) pair(&asp[to][ci][0], &rc[ci * WPL], racc32[to], racc32_lo[C::F16 ? to : 0]);
      }
      // order template as in the fp32 loop: fragment reads first, the next iteration's global loads (staging + weight
      // planes) spread evenly over the MFMA stream, the plane split and its LDS writes last
      {
        constexpr int NUSED = (tap_used(MODE, TO, 0) ? 1 : 0) + (NJ > 1 && tap_used(MODE, TO, 1) ? 1 : 0) + (NJ > 2 && tap_used(MODE, TO, 2) ? 1 : 0) +
                              (NJ > 3 && tap_used(MODE, TO, 3) ? 1 : 0) + (NJ > 4 && tap_used(MODE, TO, 4) ? 1 : 0);
        constexpr int NLOADS = LAST ? 0 : C::NLD + (NUSED + (RES_OUT ? 1 : 0)) * CPI * C::NPL;
        constexpr int NMFMA = CPI * (C::F16 ? 3 : 6) * (valid_pairs(MODE, TO) + (RES_OUT ? TO : 0));
        constexpr int MPL = NMFMA / (NLOADS > 0 ? NLOADS : 1) > 0 ? NMFMA / (NLOADS > 0 ? NLOADS : 1) : 1;
#pragma unroll
        for (int i = 0; i < TI * CPI * C::NPL; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
        }
        if (NMFMA - NLOADS * MPL > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NLOADS * MPL, 0);
      }
      if (!LAST) stage_store(xnext);
      __syncthreads();
      return;
    }
    f32x4 areg[MB][TI][CPI];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int ci = 0; ci < CPI; ++ci)
          areg[m][ti][ci] = *reinterpret_cast<const f32x4*>(
              xcur + (((m * TI + ti) * NC + ks * CPI + ci) * 16 + r) * 16 + swz(r, kq) * 4);

    // First conv of an evaluation (the T = 8 / 16 tiles with the projection): its 128-channel virtual chunk holds
    // 32 stored channels, so the K-slice waves behind them would multiply exact zeros.  They skip their MFMAs (their
    // accumulators stay +0, which is what the zero products add up to) and all waves share the epilogue.
    const bool dead_slice = SKIPZ && a.ca_real > 0 && (it * C::CH_IT + ks * CPI * 16) >= a.ca_real;
    if (!dead_slice) {
#pragma unroll
    for (int ci = 0; ci < CPI; ++ci) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int to = 0; to < TO; ++to) {
            const int ti = tap_src(MODE, to, j);
            if (ti >= 0 && ti < TI) {
#pragma unroll
              for (int m = 0; m < MB; ++m)
                acc[m][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[m][ti][ci][s], bc[j][ci][s],
                                                                  acc[m][to], 0, 0, 0);
            }
          }
        }
        if (RES_OUT) {
#pragma unroll
          for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int m = 0; m < MB; ++m)
              racc[m][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[m][to][ci][s], rc[ci][s],
                                                                 racc[m][to], 0, 0, 0);
        }
      }
    }
    }
    // Order template for the machine scheduler: fragment reads from LDS first, then one global
    // load issued every MPL MFMAs (a wave that issues all its loads up front sits in the memory
    // pipe's queue while the MFMA pipe idles), then the LDS writes of the next activation tile.
    if (LAST) {
#pragma unroll
      for (int i = 0; i < MB * TI * CPI; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    } else {
      constexpr int NUSED = (tap_used(MODE, TO, 0) ? 1 : 0) + (NJ > 1 && tap_used(MODE, TO, 1) ? 1 : 0) +
                            (NJ > 2 && tap_used(MODE, TO, 2) ? 1 : 0) + (NJ > 3 && tap_used(MODE, TO, 3) ? 1 : 0) +
                            (NJ > 4 && tap_used(MODE, TO, 4) ? 1 : 0);
      constexpr int NLOADS = C::NLD + (NUSED + (RES_OUT ? 1 : 0)) * CPI;
      constexpr int NMFMA = MB * CPI * 4 * (valid_pairs(MODE, TO) + (RES_OUT ? TO : 0));
      // loads are spread evenly over the whole MFMA stream: each is consumed at the same position of the
      // next iteration, i.e. every load gets exactly one iteration of prefetch distance (bunching them
      // into the first half or third of the stream measured 1.5 % / 4 % slower)
      constexpr int MPL = NMFMA / NLOADS > 0 ? NMFMA / NLOADS : 1;
#pragma unroll
      for (int i = 0; i < MB * TI * CPI; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int i = 0; i < NLOADS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
      }
      if (NMFMA - NLOADS * MPL > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NLOADS * MPL, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, C::NLD, 0);      // LDS writes last
    }
    if (!LAST) stage_store(xnext);
    __syncthreads();
  };

  // ---- prologue ---------------------------------------------------------------------------
  stage_load(it0);
  wload(S16 ? it0 * NSTEP : it0, wb0, rb0);
  stage_store(smem + (it0 & 1) * C::XT);
  __syncthreads();
  LDP_TL(1);

  // ---- main loop over input-channel chunks ---------------------------------------------------
  // Unrolled by two with the buffer roles swapped.  The counted loop holds whole pairs only and the one or two
  // closing iterations live behind it (round-3 findings, both read off the ISA):
  //  * the last iteration of the K range gets its own copy of the body without prefetch (LAST): the branch-free
  //    version re-requested its own chunk -- 64 KB of activations + 40 KB of weights per work-group at T=8 -- and
  //    the epilogue's operand loads queued behind those dead loads (+4 % plans/s at 256 plans);
  //  * with `if (it + 1 < nit) second half` inside the loop the backend's unified loop exit routes the odd exit
  //    through the latch, the first half's weight loads look pending on a path into the header and every trip
  //    opens with s_waitcnt vmcnt(0).  A straight-line pair body has no such path (the waits are the designed
  //    vmcnt(12..) again); the same loop with `break` exits, or all four copies inside one for(;;), cost 40-60
  //    more VGPRs and spilled in the two-row-block tiles.
  if (mode_2d(MODE) && !S16) {
    // The StableVAE's 3x3 convs keep the round-2 loop (branch-free, the last iteration re-requests its own chunk):
    // measured on one box at N = 256, encode 26.96 ms with it, 27.9 ms with the peeled last iteration, 29.3 ms
    // with zero padding as an address select (no wait behind the staging loads any more -- and slower), 30.2 ms
    // with the last k-step's MFMAs deferred behind the next iteration's LDS reads.  Long steady-state loops over
    // thousands of work-groups do not behave like the planner's launch-bound ones.
    for (int it = it0; it < nit; it += 2) {
      iteration(std::false_type{}, it, wb0, rb0, wb1, rb1);
      if (it + 1 < nit) iteration(std::false_type{}, it + 1, wb1, rb1, wb0, rb0);
    }
  } else {
    int it = it0;
    if constexpr (S16 && NSTEP % 2 == 0) {
      // an even number of steps leaves the next iteration's first step in the buffer this one started from: no role swap
      for (; it + 1 < nit; ++it) iteration(std::false_type{}, it, wb0, rb0, wb1, rb1);
      iteration(std::true_type{}, it, wb0, rb0, wb1, rb1);
    } else {
    for (; it + 2 < nit; it += 2) {
      iteration(std::false_type{}, it, wb0, rb0, wb1, rb1);
      iteration(std::false_type{}, it + 1, wb1, rb1, wb0, rb0);
    }
    if (it + 2 == nit) {
      iteration(std::false_type{}, it, wb0, rb0, wb1, rb1);
      iteration(std::true_type{}, it + 1, wb1, rb1, wb0, rb0);
    } else if (it + 1 == nit) {
      iteration(std::true_type{}, it, wb0, rb0, wb1, rb1);
    }
    }
  }

  if constexpr (C::F16 && S32) {
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc32[t][i] = acc32[t][i] + acc32_lo[t][i] * (1.0f / 2048.0f);
    if constexpr (RES_OUT) {
#pragma unroll
      for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) racc32[t][i] = racc32[t][i] + racc32_lo[t][i] * (1.0f / 2048.0f);
    }
  }
  if constexpr (C::F16 && !S32) {      // the low products carry the plane scale 2^11
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int t = 0; t < TO; ++t) acc[m][t] = acc[m][t] + acc_lo[m][t] * (1.0f / 2048.0f);
      if constexpr (RES_OUT) {
#pragma unroll
        for (int t = 0; t < TO; ++t) racc[m][t] = racc[m][t] + racc_lo[m][t] * (1.0f / 2048.0f);
      }
    }
  }
  // ---- epilogue: (optional second pass for the fused 1x1 residual conv) ----------------------
  LDP_TL(2);
  if LDP_ABL(16) return;
  // tile e[ks][to][row][col], row stride BNP
  const int ecol = wn * 16 + (lane & 15);
  const int erow0 = (lane >> 4) * 4;
  const int flags = (a.flags & mode_flag_mask(MODE)) | mode_flag_forced(MODE);
  constexpr int EPL = C::EPL;
  constexpr int NS = 16 * MB;                          // samples of this work-group
  constexpr int SPW = (NS + C::NW - 1) / C::NW;        // samples each wave finishes
  constexpr bool FULL = (NS % C::NW) == 0;             // every wave finishes exactly SPW samples
  // Everything the epilogue needs from global memory is requested here, before the LDS exchange of the accumulators,
  // so the L2 latencies overlap the barrier instead of serialising per sample.  Round 3: the optional operands sit
  // behind uniform branches and are kept RAW (FiLM's two addends are added where they are used): the former
  // branch-free version read dummy addresses for the operands a layer does not have, the compiler folded the
  // dummies into loads it already had and put a wait for those in the middle of the requests -- a full memory
  // latency between one batch of requests and the next (read off the ISA; tools/timeline.py phase "acc->lds").
  // Addressing: element (to, c) of sample b in a (B, TO, cout) tensor sits at  [uniform part: sample, e-th row
  // group / column half] + [per-lane part, the same for every sample and e].  The uniform part stays in SGPRs and
  // the loads/stores take the (scalar base + 32-bit lane offset) form: no per-access 64-bit VALU arithmetic in the
  // issue-bound epilogue, and no v_mad_u64_u32 whose dead upper half made the compiler wait for an unrelated load.
  const unsigned lane_to = BN <= 64 ? (unsigned)lane / BN : 0u;              // element el = lane + 64 e -> (to, col)
  const unsigned lane_col = BN <= 64 ? (unsigned)lane % BN : (unsigned)lane;
  auto e_to = [](int e) { return BN <= 64 ? e * (64 / BN) : e / (BN / 64); };
  auto e_col = [](int e) { return BN <= 64 ? 0 : 64 * (e % (BN / 64)); };
  const unsigned lane_chan = (unsigned)(cbk * BN) + lane_col;                // channel of this lane's elements (+ e_col)
  // 24-bit multiplies (lane_to < 64, widths < 2^24): a plain 32-bit a * b + c becomes v_mad_u64_u32 on gfx9, which
  // reads a 64-bit addend -- whatever sits in the upper register, e.g. a load still in flight
  const unsigned lane_eoff = __umul24(lane_to, (unsigned)a.cout) + lane_chan;      // offset inside a sample's (TO, cout) block
  const unsigned lane_uoff = __umul24(lane_to, (unsigned)a.d_real) + lane_chan;    // same for the unpadded (rows, D) tensors
  float p_bias[EPL], p_gs[EPL], p_gb[EPL], p_rb[EPL];
  float p_fts[SPW][EPL], p_fgs[SPW][EPL], p_ftb[SPW][EPL], p_fgb[SPW][EPL], p_add[SPW][EPL], p_nz[SPW][EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    p_bias[e] = (a.bias + e_col(e))[lane_chan];
    p_gs[e] = (flags & EP_GN) ? (a.gn_scale + e_col(e))[lane_chan] : 1.0f;
    p_gb[e] = (flags & EP_GN) ? (a.gn_bias + e_col(e))[lane_chan] : 0.0f;
    p_rb[e] = RES_OUT ? (a.bres + e_col(e))[lane_chan] : 0.0f;
  }
  const bool f_film = (flags & EP_FILM) != 0, f_res = (flags & EP_RESIN) != 0;
  const bool f_step = (flags & EP_STEP) != 0;
#pragma unroll
  for (int si = 0; si < SPW; ++si)
#pragma unroll
    for (int e = 0; e < EPL; ++e) p_fts[si][e] = p_fgs[si][e] = p_ftb[si][e] = p_fgb[si][e] = p_add[si][e] = p_nz[si][e] = 0.0f;
  // rows beyond B are clamped to the last sample (never stored)
  if (f_film) {
#pragma unroll
    for (int si = 0; si < SPW; ++si) {
      const int sr = wave + si * C::NW;
      const int b = (b0 + sr) < a.B ? (b0 + sr) : (a.B - 1);
      int kk = a.k;
      if (a.k_dev) kk = a.k_dev[b];
      const float* ft = a.film_t + (size_t)kk * a.film_stride;
      const float* fg = a.film_g + (size_t)b * a.film_stride;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        p_fts[si][e] = (ft + e_col(e))[lane_chan];
        p_fgs[si][e] = (fg + e_col(e))[lane_chan];
        p_ftb[si][e] = (ft + a.cout + e_col(e))[lane_chan];
        p_fgb[si][e] = (fg + a.cout + e_col(e))[lane_chan];
      }
    }
  }
  if (f_res || f_step) {
    const float* rp = f_res ? a.res_in : a.out;          // EP_STEP: x_t lives in out (read and written by this lane only)
#pragma unroll
    for (int si = 0; si < SPW; ++si) {
      const int sr = wave + si * C::NW;
      const int b = (b0 + sr) < a.B ? (b0 + sr) : (a.B - 1);
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        p_add[si][e] = (rp + ((size_t)b * TO + e_to(e)) * a.cout + e_col(e))[lane_eoff];
    }
  }
  if (f_step && a.noise && a.coef.sigma != 0.f) {
#pragma unroll
    for (int si = 0; si < SPW; ++si) {
      const int sr = wave + si * C::NW;
      const int b = (b0 + sr) < a.B ? (b0 + sr) : (a.B - 1);
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int to = (int)lane_to + e_to(e), c = (int)lane_chan + e_col(e);
        if (c < a.d_real && (b * TO + to) < a.rows_valid)
          p_nz[si][e] = (a.noise + ((size_t)b * TO + e_to(e)) * a.d_real + e_col(e))[lane_uoff];
      }
    }
  }

  {
    constexpr int pass = 0;
    if constexpr (S32) {
      // 32 x 32 tiles: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the 32 samples = (row block m, row)
#pragma unroll
      for (int to = 0; to < TO; ++to)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row32 = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          smem[((((row32 >> 4) * KS + ks) * TO + to) * 16 + (row32 & 15)) * BNP + wn * 32 + (lane & 31)] = acc32[to][i];
        }
    } else
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int to = 0; to < TO; ++to) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          smem[(((m * KS + ks) * TO + to) * 16 + erow0 + i) * BNP + ecol] = acc[m][to][i];
      }
    }
    __syncthreads();
    LDP_TL(3);

    // ---- K split over work-groups (ConvArgs::kw) --------------------------------------------------
    constexpr int TILE = NS * TO * BN;
    constexpr bool KW_OK = KWS && (MODE == MODE_K5 || MODE == MODE_DOWN || MODE == MODE_UP) && MB == 1 && TO * BN <= 512;      // mirrored by tconv_kw_ok()
    unsigned long long* kw_tile = nullptr;
    unsigned int ktag = 0;
    if (KW_OK && kw > 1) {
      const int tile_id = sb * (ngroups * cs) + cbk;               // (sample block, column block)
      kw_tile = a.kw_slab + (size_t)(tile_id * kw) * (2 * TILE);   // part p: + p * 2 * TILE; projection: + TILE
      // unique per (call, denoising step, launch of the evaluation): 12 bits of the call epoch (the host wipes the
      // slab every 2048 calls), 14 bits step + 1 (never 0), 6 bits launch slot
      ktag = (((unsigned int)a.ctl[2] & 0xfffu) << 20) | (((unsigned int)a.step + 1u) << 6) | (unsigned int)a.kw_slot;
      if (kpart != 0) {
        // a K-partial work-group: publish the KS-combined tile(s) and leave.  Agent-scope stores: written through,
        // so a consumer on any XCD finds them (the usual placement puts all parts of a tile on one XCD's L2).
        unsigned long long* mine = kw_tile + (size_t)kpart * (2 * TILE);
#pragma unroll
        for (int pass = 0; pass < (RES_OUT ? 2 : 1); ++pass) {
          if (pass == 1) {
            __syncthreads();
#pragma unroll
            for (int to = 0; to < TO; ++to)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                smem[((ks * TO + to) * 16 + erow0 + i) * BNP + ecol] = racc[0][RES_OUT ? to : 0][i];
            __syncthreads();
          }
          // a lane publishes two neighbouring elements with ONE 16-byte write-through store (every 8-byte sc1 store
          // is a fabric write of its own, and the launch cannot end before each is acknowledged); the consumer's
          // 8-byte polls do not care how the granules were written
          const auto krsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 0x7ffffff0, 0x00020000);
          constexpr int PAIRS = TO * BN / 2;                   // pairs per sample row: 32 .. 256
#pragma unroll
          for (int si = 0; si < SPW; ++si) {
            const int sr = wave + si * C::NW;
            if (!FULL && sr >= NS) continue;
#pragma unroll
            for (int pc = 0; pc < (PAIRS + 63) / 64; ++pc) {
              if (PAIRS < 64 && lane >= PAIRS) continue;
              const int el = 2 * (lane + 64 * pc);
              const int to = el / BN, col = el % BN;
              float x0 = 0.0f, x1 = 0.0f;
#pragma unroll
              for (int k2 = 0; k2 < KS; ++k2) {
                x0 += smem[((k2 * TO + to) * 16 + sr) * BNP + col];
                x1 += smem[((k2 * TO + to) * 16 + sr) * BNP + col + 1];
              }
              const unsigned long long g0 = granule_pack(ktag, x0), g1 = granule_pack(ktag, x1);
              const u32x4_t gv = {(unsigned int)g0, (unsigned int)(g0 >> 32), (unsigned int)g1, (unsigned int)(g1 >> 32)};
              __builtin_amdgcn_raw_buffer_store_b128(gv, krsrc, (unsigned int)((pass * TILE + sr * (TO * BN) + el) * 8), 0, AUX_SC1);
            }
          }
        }
        return;
      }
    }
    // the other parts' partial of element (sr, el), added in part order.  All granules of the element are requested
    // together; the wave re-requests until every lane holds current tags (bounded: a peer that never publishes ends
    // in the fault word like the statistics exchange's).
    auto kw_add = [&](float x, int pass, int sr, int el) {
      if (!(KW_OK && kw > 1)) return x;
      unsigned long long pv[KW_MAX - 1];
      int spin = 0;
      for (;;) {
        bool ok = true;
        // kw - 1 polls (kw is uniform: the guards are scalar branches), not KW_MAX - 1
#pragma unroll
        for (int q = 0; q < KW_MAX - 1; ++q) {
          pv[q] = 0;
          if (q + 1 < kw)
            pv[q] = __hip_atomic_load(kw_tile + (size_t)(q + 1) * (2 * TILE) + pass * TILE + sr * (TO * BN) + el,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // all polls landed before any is checked (see the statistics exchange)
#pragma unroll
        for (int q = 0; q < KW_MAX - 1; ++q) ok = ok && (q + 1 >= kw || granule_ok(pv[q], ktag));
        if (__all(ok)) break;
        if (++spin > (1 << 18) || ((spin & 1023) == 0 && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
          if (lane == 0) *a.fault = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < KW_MAX - 1; ++q) x += (q + 1 < kw) ? __uint_as_float((unsigned int)pv[q]) : 0.0f;
      return x;
    };

    // phase A: K-split partial sums -> values, per-sample (sum, sum of squares); with a column-split
    // group the half-sums are published to the peer work-group as {value, tag} granules: ONE
    // 8-byte agent-scope (write-through) store each, so a granule is never torn and needs no fence.
    const bool xch = (cs > 1) && (flags & EP_GN) && !LDP_ABL(32);
    // slab: [row block][group][part][16 samples][2] granules, addressed through a raw buffer descriptor (compiler-known
    // 16-byte sc1 accesses; offsets in bytes)
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(a.xchg), 0, 0x7ffffff0, 0x00020000);
    unsigned int xbase = 0;
    unsigned int tag = 0;
    if (xch) {
      xbase = (unsigned int)((sb * MB * ngroups + grp) * 4) * 256u;
      tag = ((unsigned int)a.ctl[2] << 12) + (unsigned int)a.step + 1u;     // unique per (call, step)
    }
    float vv[SPW][EPL];
    float s1a[SPW], s2a[SPW];
    bool range_bad = false;            // fp16-plane tiles: a sum of squares that is not finite (the range guard below)
#pragma unroll
    for (int si = 0; si < SPW; ++si) {
      const int sr = wave + si * C::NW;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int el = lane + 64 * e;
        const int to = el / BN, col = el % BN;
        float x = p_bias[e];
        if (FULL || sr < NS) {
#pragma unroll
          for (int k2 = 0; k2 < KS; ++k2)
            x += smem[((((sr >> 4) * KS + k2) * TO + to) * 16 + (sr & 15)) * BNP + col];
          x = kw_add(x, 0, sr, el);
        }
        vv[si][e] = x;
        s1 += x;
        s2 += x * x;
      }
      if (flags & EP_GN) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (xch && (FULL || sr < NS) && lane == 0) {
          // ONE 16-byte write-through store for the sample's two granules.  Two 8-byte stores were two fabric writes,
          // and a launch cannot complete before the memory side has acknowledged every one of them: behind the
          // weight-streaming layers the gap to the next launch was 1 us longer (round 3, tools/timeline.py).
          const unsigned long long g1 = granule_pack(tag, s1), g2 = granule_pack(tag, s2);
          const u32x4_t gv = {(unsigned int)g1, (unsigned int)(g1 >> 32), (unsigned int)g2, (unsigned int)(g2 >> 32)};
          __builtin_amdgcn_raw_buffer_store_b128(gv, xrsrc, xbase + (unsigned int)(((sr >> 4) * ngroups * 4 + half) * 256 + (sr & 15) * 16), 0, AUX_SC1);
        }
      }
      if constexpr (C::F16) {
        // Range guard of the fp16 planes (|x| < 65504; fp32 and the bf16 planes hold 3.4e38): an operand element beyond it is +-inf in its h plane,
        // every product with it is inf or NaN, and so is every output column of that sample at the positions its taps reach -- the sum of squares
        // this epilogue forms anyway (per sample and group behind the wave sum, per lane in the tiles without a GroupNorm) cannot stay finite.  A
        // non-finite sum raises word 1 of the pinned fault block; the host then recomputes the call on three bf16 planes (engine.hpp range_fallback).
        // One compare per sample, OR-ed into a flag; one conditional store behind the epilogue (never executed on in-range data).
        if (LDP_RANGE_GUARD) range_bad = range_bad || !(s2 <= 3.0e38f);
      }
      s1a[si] = s1;
      s2a[si] = s2;
    }

    LDP_TL(4);
    // The block's 1x1 residual projection (second accumulator set) goes through the same LDS tile
    // while the peer work-group's statistics granules are in flight.
    if (RES_OUT) {
      __syncthreads();                     // everyone finished reading the main tile
      if constexpr (S32) {
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row32 = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            smem[((((row32 >> 4) * KS + ks) * TO + to) * 16 + (row32 & 15)) * BNP + wn * 32 + (lane & 31)] = racc32[RES_OUT ? to : 0][i];
          }
      } else
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int to = 0; to < TO; ++to) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            smem[(((m * KS + ks) * TO + to) * 16 + erow0 + i) * BNP + ecol] = racc[m][RES_OUT ? to : 0][i];
        }
      }
      __syncthreads();
#pragma unroll
      for (int si = 0; si < SPW; ++si) {
        const int sr = wave + si * C::NW;
        const int b = b0 + sr;
        if ((!FULL && sr >= NS) || b >= a.B) continue;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const int el = lane + 64 * e;
          const int to = el / BN, col = el % BN;
          float x = p_rb[e];
#pragma unroll
          for (int k2 = 0; k2 < KS; ++k2)
            x += smem[((((sr >> 4) * KS + k2) * TO + to) * 16 + (sr & 15)) * BNP + col];
          x = kw_add(x, 1, sr, el);
          if (!LDP_ABL(128) || x == 12345.f) (a.res_out + ((size_t)b * TO + e_to(e)) * a.cout + e_col(e))[lane_eoff] = x;
        }
      }
    }

    LDP_TL(5);
    // phase B: statistics (own half + peer half, always summed as half0 + half1), normalise, store
    float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
    for (int si = 0; si < SPW; ++si) {
      const int sr = wave + si * C::NW;
      if (!FULL && sr >= NS) continue;
      const int b = b0 + sr;
      const bool live = b < a.B;
      float* v = vv[si];
      float mean = 0.f, rstd = 1.f;
      if (flags & EP_GN) {
        float s1 = s1a[si], s2 = s2a[si];
        if (xch) {
          // every part adds the cs partial sums in part order 0..cs-1 (its own from registers): all
          // work-groups of the group obtain bit-identical statistics
          // the cs - 1 peers' records are requested together (one L2 round trip, not one per peer) and re-requested
          // until every one carries this call's tag; bounded: a peer that never publishes ends in the fault word
          u32x4_t pg[3];
          int spin = 0;
          for (;;) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              pg[q] = u32x4_t{0u, 0u, 0u, 0u};
              if (q < cs - 1) {
                const int pp = q < half ? q : q + 1;
                pg[q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xbase + (unsigned int)(((sr >> 4) * ngroups * 4 + pp) * 256 + (sr & 15) * 16), 0, AUX_SC1);
              }
            }
            // Every poll has landed before any is looked at, and none is left in flight into the next round: agent-scope
            // loads that miss (peer on another XCD: two processes sharing the GPU) were seen to be overtaken by later
            // ones that hit, while the compiler's partial vmcnt(N) waits assume in-order returns.
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pg[0]), "+v"(pg[1]), "+v"(pg[2]) :: "memory");
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 3; ++q)
              ok = ok && (q >= cs - 1 || (((pg[q][0] ^ pg[q][1]) == tag) && ((pg[q][2] ^ pg[q][3]) == tag)));
            if (ok) break;
            if (++spin > (1 << 20) || ((spin & 1023) == 0 && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
              if (lane == 0) *a.fault = 1u;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
#ifdef LDP_TIMELINE
          if (a.tl && lane == 0) {      // exchanges and extra poll rounds of this launch (tools/timeline.py)
            atomicAdd(&a.tl[120000], 1ull);
            atomicAdd(&a.tl[120001], (unsigned long long)spin);
          }
#endif
          // every part adds the cs partial sums in part order 0..cs-1 (its own from registers)
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            if (pp >= cs) continue;
            const int q = pp < half ? pp : pp - 1;
            float p1 = s1, p2 = s2;
            if (pp != half) {
              p1 = __uint_as_float(q == 0 ? pg[0][0] : q == 1 ? pg[1][0] : pg[2][0]);
              p2 = __uint_as_float(q == 0 ? pg[0][2] : q == 1 ? pg[1][2] : pg[2][2]);
            }
            t1 = pp == 0 ? p1 : t1 + p1;
            t2 = pp == 0 ? p2 : t2 + p2;
          }
          s1 = t1;
          s2 = t2;
        }
        const float inv_n = 1.0f / (float)(TO * BN * cs);
        mean = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - mean * mean, 0.0f);
#if LDP_EPILOGUE_IEEE
        rstd = 1.0f / sqrtf(var + 1e-6f);
#else
        rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
#endif
      }
      if (!live) continue;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int el = lane + 64 * e;
        const int to = el / BN, col = el % BN;
        const int c = cbk * BN + col;
        float y = v[e];
        if (flags & EP_GN) {
          y = (y - mean) * rstd * p_gs[e] + p_gb[e];
          y = mish_f(y);
        }
        if (flags & EP_FILM) y = (p_fts[si][e] + p_fgs[si][e]) * y + (p_ftb[si][e] + p_fgb[si][e]);
        float* const optr = a.out + ((size_t)b * TO + e_to(e)) * a.cout + e_col(e);      // uniform; element at [lane_eoff]
        if (flags & EP_RESIN) y += p_add[si][e];
        if (flags & EP_RELU) y = fmaxf(y, 0.0f);
        if (flags & (EP_STEP | EP_EPSOUT)) {
          if (c < a.d_real && (b * TO + to) < a.rows_valid) {
            if (flags & EP_EPSOUT) (a.eps_out + ((size_t)b * TO + e_to(e)) * a.d_real + e_col(e))[lane_uoff] = y;
            if (flags & EP_STEP) {
              const float xt = p_add[si][e];
              float z = 0.f;
              if (a.coef.sigma != 0.f) {
                if (a.noise) z = p_nz[si][e];
                // element index = (first global row + local row) * padded width + channel: a draw depends on
                // the global row only, never on how rows are grouped into work-groups or shards
                else z = philox_normal(a.seed[0], (a.seed[1] + (uint64_t)(b * TO + to)) * (uint64_t)a.cout
                                       + (uint64_t)c, (uint32_t)a.step, 0u);
              }
              float x0 = (xt - a.coef.sqrt_1mab * y) * a.coef.inv_sqrt_ab;
              x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
              optr[lane_eoff] = a.coef.c_x0 * x0 + a.coef.c_x * xt + a.coef.c_eps * y + a.coef.sigma * z;
            }
          }
        } else if (!LDP_ABL(128) || y == 12345.f) {
          optr[lane_eoff] = y;
          if (C::STATS) { gs1 += y; gs2 += y * y; }          // BN == 64: this lane's elements are one column's TO pixels
        }
      }
    }
    if constexpr (C::F16) {
      if (LDP_RANGE_GUARD && range_bad && a.fault) a.fault[1] = 1u;
    }
    LDP_TL(6);
#ifdef LDP_TIMELINE
    if (a.tl) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); LDP_TL(7); }
#endif
    if (C::STATS) {
      if (a.stats_part) {                                // uniform: every wave of the work-group takes this path
        float* sp = smem + C::TILE_FLOATS;               // [2][NW][64], behind the tiles
        sp[wave * 64 + lane] = gs1;
        sp[(C::NW + wave) * 64 + lane] = gs2;
        __syncthreads();
        if (tid < 128) {
          const int which = tid >> 6, c = tid & 63;
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < C::NW; ++w) t += sp[(which * C::NW + w) * 64 + c];       // fixed order
          a.stats_part[((size_t)sb * a.cout + cbk * BN + c) * 2 + which] = t;
        }
      }
    }
  }
}

// shapes whose kernels carry the K-split-over-work-groups path (KW_OK in tconv_kernel)
__host__ __device__ constexpr bool tconv_kw_ok(int mode, int to, int nwn, int mb) {
  return (mode == MODE_K5 || mode == MODE_DOWN || mode == MODE_UP) && mb == 1 && to * 16 * nwn <= 512;
}

// host-side launchers: pick the instantiation named by the plan
struct ConvPlan {
  int mode, to, nwn, ks, cpi, res_out;
  int mb = 1;                                       // 16-sample row blocks per work-group
  int kws = 0;                                      // instantiation that carries the K-split path
  int split = 0;                                    // main loop on split bf16 operands (TConvCfg SPLIT): mb = 2, ConvArgs::w = the plane-packed weights
  int bn() const { return 16 * nwn; }
  int chunk() const { return 16 * ks * cpi; }       // input channels consumed per iteration
};
int tconv_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);   // -100: no such instantiation
int tconv_init_all();   // raises the dynamic-LDS limit of every instantiation (call before graph capture)
int tconv_init_k5();
int tconv_init_k5r();
int tconv_init_misc();
int tconv_init_2d();
int tconv_launch_2d(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_k5(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_k5r(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_misc(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_launch_split(const ConvPlan& p, const ConvArgs& a, hipStream_t stream);
int tconv_init_split();

}  // namespace ldp
