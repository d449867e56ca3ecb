// train.hip -- the reference's training step on the MI355X (the last row SURVEY 8 names).
//
//   agent/ldp_agent.py:113-180   plan_loss / idm_loss / loss            -> forward with saved activations + MSE
//   agent/ldp_agent.py:252       jax.grad(self.loss)                    -> hand-written backward (this file)
//   agent/ldp_agent.py:253       linear_algebra.global_norm(grads)      -> ldp_train_grad_norm
//   agent/ldp_agent.py:256,265   TrainState.apply_gradients (optax.adam)-> ldp_train_apply (one fused Adam kernel over a flat arena)
//
// Design (MI355X first, not a port of XLA's lowering).  Everything that is GEMM-shaped -- Dense layers, the k = 5 / stride-2 / transposed
// / 1x1 convolutions and ALL their gradients -- runs through ONE kernel family, `seg_gemm`: a batched GEMM on v_mfma_f32_16x16x4_f32
// (exact fp32, like the sampling path's driver line) whose K range is a list of SEGMENTS, each with its own operand offsets:
//     C_z[M, N] = sum over segments s of z:  A(s)[M, K] . B(s)[K, N]        (+ bias[N]) (+ add[M, N])
// With rows = samples (as in csrc/tconv.hpp's Toeplitz tiles) a convolution over T positions is, for output position t_out, the sum over
// its LIVE taps j of  X[:, t_in(t_out, j), :] @ W[j]  -- taps that fall on padding are simply not in the list -- and
//     forward   z = t_out,  segments = live taps:      A = X  (lda = T_in Cin),   B = W[j]        (Cin x Cout, row-major: the Flax leaf itself)
//     dgrad     z = t_in,   segments = taps hitting it: A = dY (lda = T_out Cout), B = W[j]^T      (read transposed: "NT")
//     wgrad     z = tap j,  segments = live t_out:      A = X[:, t_in]^T ("TN": K = the batch), B = dY[:, t_out],  C = dW[j]
// so parameters and their gradients stay in the reference's Flax layouts (kernel (k, Cin, Cout), Dense (in, out)) in one flat fp32 arena per
// module; the optimiser is a single launch over that arena, the global norm a two-stage reduction over it.
// Operand tiles are staged global -> LDS (64 x 32 / 32 x 64 floats, register-prefetched one K step ahead) in whichever of the two layouts
// matches their global contiguity, so all three forms (NN / NT / TN) read their MFMA fragments conflict-free with ds_read_b32.
// GroupNorm + Mish (+ FiLM, + residual), LayerNorm, ReLU and their backward passes are HBM-bound element-wise kernels (one wave per
// (sample, group) / per row, two-pass, fixed summation order).  Rows are padded to multiples of 32 with zero loss gradient, channel
// counts that are not multiples of 32 (D, A, the FiLM / IDM input widths) are zero-padded in the arena: padded entries receive exactly
// zero gradient and stay zero under Adam.  Nothing here uses atomics: every number is bit-reproducible run to run.
#include "engine.hpp"

#include <cmath>
#include <cstring>
#include <type_traits>

#pragma clang fp contract(off)

namespace ldp {
namespace {

constexpr int RP = 32;                 // row / channel padding granule

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// =====================================================================================================================
// segmented-K batched GEMM on the exact-fp32 MFMA
// =====================================================================================================================
struct GemmSeg { long long a_off, b_off; };
struct GemmBatch { long long c_off; int seg_begin, seg_end; long long bias_off = 0; };      // bias_off: this batch's bias row = GemmArgs::bias + bias_off

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias;            // (N) or nullptr
  const float* add;             // same layout as C (may alias C: accumulate) or nullptr
  const GemmSeg* segs;
  const GemmBatch* batches;
  int M, N, K;                  // K per segment (multiple of 32)
  int lda, ldb, ldc;
  // split K (seg_gemm_big only): the batch's K steps are dealt to `ksplit` work-groups; each writes its partial tile to part + split * c_extent
  // (+ the batch's c_off, row stride ldc) and reduce_parts_kernel adds them up in split order (+ bias, + add): deterministic
  int ksplit = 1;
  long long c_extent = 0;
  float* part = nullptr;
  // cnt != nullptr: no reduce launch -- every work-group of an output tile publishes its partial, takes a ticket, and the one that draws the last
  // ticket adds the tile's partials up (in split order, its own read back from memory like the others': the bits do not depend on who is last)
  unsigned int* cnt = nullptr;
  // second bases: a segment whose a_off carries GEMM_ALT reads A2 instead of A, a batch whose c_off carries it writes C2 instead of C (two
  // convolutions over the same input -- or into the same gradient -- as ONE launch: plan_proj)
  const float* A2 = nullptr;
  float* C2 = nullptr;
};
constexpr long long GEMM_ALT = 1ll << 62;

constexpr int BK = 32;
constexpr int LDS_KC = BK + 4;         // [row][k] tile: row stride (36 floats: ds_read_b32 of 16 rows x 4 k conflict-free)

// Split-K finish inside the GEMM launch (GemmArgs::cnt).  A work-group stores its partial tile in the accumulators' own order -- one 16-byte
// agent-scope (sc1: written through this XCD's L2) store per lane and 16 x 16 block, 1 KB per wave instruction -- into block (split, tile) of the
// workspace, waits for the stores (vmcnt 0), and takes a ticket of its tile.  The work-group that draws the last ticket reads the `ksplit` blocks
// back with sc1 loads (served past its own L2: the other splits ran on other XCDs) and adds them in split order, its own included -- the bits equal
// reduce_parts_kernel's.  No work-group waits for another (no co-residency requirement, unlike the sampling path's in-launch exchanges), and no
// L2 is written back or invalidated wholesale: the agent-scope fence pair (buffer_wbl2 / buffer_inv around the ticket) made the step 2.3 x slower.
struct SplitK {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned int tile, ntiles, block_bytes;
};
template <int BLOCK_FLOATS>
__device__ __forceinline__ SplitK splitk_open(const GemmArgs& g, int zb) {
  SplitK k;
  k.tile = (zb * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  k.ntiles = gridDim.x * gridDim.y * (gridDim.z / g.ksplit);
  k.block_bytes = BLOCK_FLOATS * 4;
  k.rsrc = __builtin_amdgcn_make_buffer_rsrc(g.part, 0, 0xfffffff0u, 0x00020000);
  return k;
}
__device__ __forceinline__ unsigned int splitk_off(const SplitK& k, int split, int slot) {     // byte offset of a lane's 16 bytes: block (split, tile), slot = 16 x 16 block of the tile
  return ((unsigned int)split * k.ntiles + k.tile) * k.block_bytes + (unsigned int)(slot * 64 + (threadIdx.x & 63)) * 16u;
}
__device__ __forceinline__ void splitk_put(const SplitK& k, int split, int slot, const f32x4& v) {
  const u32x4_t w = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(w, k.rsrc, splitk_off(k, split, slot), 0, AUX_SC1);
}
__device__ __forceinline__ f32x4 splitk_get(const SplitK& k, int split, int slot) {
  const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(k.rsrc, splitk_off(k, split, slot), 0, AUX_SC1);
  return f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
}
// after the puts: true in the work-group that adds the tile up
__device__ __forceinline__ bool splitk_last(const GemmArgs& g, const SplitK& k) {
  __shared__ unsigned int s_ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial has left for the fabric
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(g.cnt + k.tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const bool last = s_ticket == (unsigned int)g.ksplit - 1u;
  if (last && threadIdx.x == 0) __hip_atomic_store(g.cnt + k.tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  return last;
}

// A_KC: A is (M x K) row-major (k contiguous)   -- else A is (K x M) row-major (the TN form)
// B_KC: B is (N x K) row-major (the NT form)    -- else B is (K x N) row-major
// tile: BM = 32 MT (MT = 2: 64, MT = 1: 32) x BN = 64, 256 threads.  MT = 2: 2 x 2 waves of 32 x 32 (four independent accumulators per wave);
// MT = 1: 1 x 4 waves of 32 x 16 (two accumulators: what the 40-cycle dependent latency of the 32-cycle MFMA needs) -- twice the work-groups
// for the launches whose 64 x 64 tiling would leave half the chip idle (M = the batch = 256: four row tiles).
// Operand tiles travel global -> registers FOUR K steps ahead (the launches are small -- 8 to 64 K steps per work-group, one or two work-groups per CU -- so a
// K step that waits for its own loads costs a full L2 / HBM latency: with two steps in flight a step took ~2.9 us whatever its arithmetic),
// registers -> LDS one step ahead, double-buffered LDS, one barrier per step.
// KI = 2: the work-group is TWO such wave quartets (512 threads) on one output tile; each runs half of the work-group's K steps through its own LDS
// buffers and the second hands its accumulators over through LDS at the end -- a split of K that costs no partial block in memory, no ticket and no
// read-back (gemm_shape halves the split over work-groups for it).  Both quartets execute the same number of barriers.
template <bool A_KC, bool B_KC, int MT, int KI>
__global__ __launch_bounds__(256 * KI) void seg_gemm(const GemmArgs g) {
  constexpr int BM = 32 * MT, BN = 64;
  constexpr int LDS_KSA = BM + 16, LDS_KSB = BN + 16;      // [k][row] tiles: row stride = 16 mod 32 floats (conflict-free ds_read_b32 fragments)
  constexpr int NA = MT;                                   // float4 per thread of an A tile (BM x 32 floats / 256 threads / 4)
  __shared__ float As_[KI][2][A_KC ? BM * LDS_KC : BK * LDS_KSA];
  __shared__ float Bs_[KI][2][B_KC ? BN * LDS_KC : BK * LDS_KSB];
  const int half = KI == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;      // wave-uniform
  auto& As = As_[half];
  auto& Bs = Bs_[half];
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = MT == 2 ? wave >> 1 : 0, wn = MT == 2 ? wave & 1 : wave;
  constexpr int WN = MT == 2 ? 32 : 16, TN = WN / 16;      // columns / column blocks per wave
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int zb = blockIdx.z / g.ksplit, sp = blockIdx.z - zb * g.ksplit;
  const GemmBatch bt = g.batches[zb];
  const int nk = g.K / BK;
  const int all = (bt.seg_end - bt.seg_begin) * nk, per = (all + g.ksplit - 1) / g.ksplit;
  const int wg0 = sp * per, wg_total = max(min(all, wg0 + per) - wg0, 0);       // the work-group's K steps
  const int loop_n = (wg_total + KI - 1) / KI;                                   // steps (and barriers) every quartet walks
  const int mine = max(min(wg_total - half * loop_n, loop_n), 0);                // ... of which this quartet's are real
  const int it0 = mine > 0 ? wg0 + half * loop_n : wg0, total = max(mine, wg_total > 0 ? 1 : 0);      // (an idle quartet re-reads the first step and adds nothing)
  f32x4 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifndef LDP_TRAIN_PD
#define LDP_TRAIN_PD 4
#endif
  constexpr int PD = LDP_TRAIN_PD;                   // global -> register prefetch depth (K steps in flight per work-group; even)
  f32x4 ra[PD][NA], rb[PD][2];
  // The load stream walks the work-group's K steps in order with per-thread operand pointers that are BUMPED from step to step; the segment table is
  // read (one scalar load) and the pointers rebuilt only where a segment ends.  (Computing segment = it / nk, k0 = it % nk and every address from
  // scratch per step cost 45 scalar instructions and a scalar load waited for on the spot -- lgkmcnt(0), LDS reads included -- in every K step.)
  const float* pA[NA];
  const float* pB[2];
  int ld_it = 0, ld_s = bt.seg_begin + (total > 0 ? it0 / nk : 0), ld_k = total > 0 ? it0 % nk : 0;
  auto set_ptrs = [&]() {
    const GemmSeg sg = g.segs[ld_s];
    const float* Ap = ((sg.a_off & GEMM_ALT) ? g.A2 : g.A) + (sg.a_off & ~GEMM_ALT);
    const float* Bp = g.B + sg.b_off;
    const int k0 = ld_k * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (A_KC) {
        int row = m0 + (tid >> 3) + 32 * i;
        row = row < g.M ? row : g.M - 1;
        pA[i] = Ap + (size_t)row * g.lda + k0 + (tid & 7) * 4;
      } else {
        // (K x M) source: BK rows of BM floats; MT = 2: 16 float4 per row, two k rows per thread; MT = 1: 8 float4 per row, one k row per thread
        const int per = BM / 4, k = tid / per + (256 / per) * i;
        int col = m0 + (tid % per) * 4;
        col = col < g.M ? col : g.M - 4;
        pA[i] = Ap + (size_t)(k0 + k) * g.lda + col;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (B_KC) {
        int row = n0 + (tid >> 3) + 32 * i;
        row = row < g.N ? row : g.N - 1;
        pB[i] = Bp + (size_t)row * g.ldb + k0 + (tid & 7) * 4;
      } else {
        const int k = (tid >> 4) + 16 * i;
        int col = n0 + (tid & 15) * 4;
        col = col < g.N ? col : g.N - 4;
        pB[i] = Bp + (size_t)(k0 + k) * g.ldb + col;
      }
    }
  };
  const size_t stepA = A_KC ? (size_t)BK : (size_t)BK * g.lda, stepB = B_KC ? (size_t)BK : (size_t)BK * g.ldb;
  // loads the stream's next step into `slot` (past the end: the last step again -- the loads of the steady state are unconditional, see below)
  auto gload = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[slot][i] = *reinterpret_cast<const f32x4*>(pA[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[slot][i] = *reinterpret_cast<const f32x4*>(pB[i]);
    if (ld_it + 1 < total) {
      ++ld_it;
      if (++ld_k == nk) {
        ld_k = 0;
        ++ld_s;
        set_ptrs();
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) pA[i] += stepA;
#pragma unroll
        for (int i = 0; i < 2; ++i) pB[i] += stepB;
      }
    }
  };
  auto lstore = [&](int buf, int slot) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (A_KC) *reinterpret_cast<f32x4*>(&As[buf][((tid >> 3) + 32 * i) * LDS_KC + (tid & 7) * 4]) = ra[slot][i];
      else {
        const int per = BM / 4;
        *reinterpret_cast<f32x4*>(&As[buf][(tid / per + (256 / per) * i) * LDS_KSA + (tid % per) * 4]) = ra[slot][i];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (B_KC) *reinterpret_cast<f32x4*>(&Bs[buf][((tid >> 3) + 32 * i) * LDS_KC + (tid & 7) * 4]) = rb[slot][i];
      else *reinterpret_cast<f32x4*>(&Bs[buf][((tid >> 4) + 16 * i) * LDS_KSB + (tid & 15) * 4]) = rb[slot][i];
    }
  };
  const int fr = lane & 15, fk = lane >> 4;          // fragment row / k of the 16x16x4 MFMA operand maps
  if (total > 0) {
    set_ptrs();
#pragma unroll
    for (int p = 0; p < PD; ++p) gload(p);
    lstore(0, 0);
  }
  __syncthreads();
  // (the loop body is unrolled PD times so that register slots and LDS buffers are compile-time constants)
  // STEADY: no branch around the loads or the LDS stores, so the compiler can count: the store of step it + 1 waits for exactly the loads issued three
  // steps ago (s_waitcnt vmcnt(9 / 12)).  With `if (it + PD < total)` around the loads every such wait was vmcnt(0) -- the four-deep prefetch ran one deep.
  auto step = [&](int it, int slot, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const int buf = slot & 1;                        // it is a multiple of PD (even) + slot
    if (STEADY || it + PD < loop_n) gload(slot);                // slot `slot` held iteration `it`: already in LDS; the stream is at step it + PD
    // All fragments of the K step first, then its MFMAs back to back.  MFMA step e takes k = 4 e + (lane >> 4) from both operands: ds_read_b32 of
    // 16 rows x 4 k values, conflict-free in both tile layouts (row stride 36 floats for [row][k], 16 mod 64 for [k][row]).  (Handing a lane
    // its eight k values as two ds_read_b128 -- k = 8 (lane >> 4) + e -- was tried: the [k][row] operand then reads rows 8 apart, whose stride is 0 mod 64
    // banks: four-way conflicts, SQ_LDS_BANK_CONFLICT 0.4 of the LDS cycles, no gain.)
    float fa[2][8], fb[TN][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 32 + i * 16 + fr;
        fa[i][e] = A_KC ? As[buf][r * LDS_KC + e * 4 + fk] : As[buf][(e * 4 + fk) * LDS_KSA + r];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = wn * WN + j * 16 + fr;
        fb[j][e] = B_KC ? Bs[buf][c * LDS_KC + e * 4 + fk] : Bs[buf][(e * 4 + fk) * LDS_KSB + c];
      }
    }
    if (KI == 1 || it < mine) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (STEADY || it + 1 < loop_n) lstore(buf ^ 1, (slot + 1) % PD);
    __syncthreads();
  };
  int it = 0;
  for (; it + PD <= loop_n; it += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) step(it + u, u, std::true_type());
  }
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (it + u < loop_n) step(it + u, u, std::false_type());
  if (KI == 2) {                                     // the second quartet's accumulators -> the first (everybody is past its last LDS read: the loop ends in a barrier)
    f32x4* red = reinterpret_cast<f32x4*>(&As_[0][0][0]);
    if (half == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) red[((wave * 2 + i) * TN + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 v = red[((wave * 2 + i) * TN + j) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] += v[e];
      }
  }
  // epilogue: C/D map of the 16x16 MFMA: lane -> column lane & 15, rows 4 (lane >> 4) + e
  const bool partial = g.ksplit > 1;
  if (partial && g.cnt) {
    const SplitK sk = splitk_open<BM * BN>(g, zb);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) splitk_put(sk, sp, (wave * 2 + i) * TN + j, acc[i][j]);
    if (!splitk_last(g, sk)) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = splitk_get(sk, 0, (wave * 2 + i) * TN + j);
        for (int q = 1; q < g.ksplit; ++q) {
          const f32x4 v = splitk_get(sk, q, (wave * 2 + i) * TN + j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] += v[e];
        }
      }
  }
  const bool direct = !partial || g.cnt;                    // this work-group writes C itself (bias and add included)
  const long long coff = bt.c_off & ~GEMM_ALT;
  float* Cp = (direct ? ((bt.c_off & GEMM_ALT) ? g.C2 : g.C) : g.part + (size_t)sp * g.c_extent) + coff;
  const float* Dp = (direct && g.add) ? g.add + coff : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + fr;
    if (n >= g.N) continue;
    const float bv = (direct && g.bias) ? g.bias[bt.bias_off + n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 32 + i * 16 + fk * 4 + e;
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (Dp) v += Dp[(size_t)m * g.ldc + n];
        Cp[(size_t)m * g.ldc + n] = v;
      }
  }
}

// The same GEMM on 128 x 128 tiles, eight waves of 32 x 64 (2 x 4 accumulators): 32 FLOP per operand byte moved L2 -> LDS instead of 11 - 16 for the
// 32 / 64-row tiles, two waves per SIMD.  M = the batch (256) leaves such a tiling with a handful of work-groups, so the K steps of a batch are
// split over work-groups (GemmArgs::ksplit) and added up by reduce_parts_kernel.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) void seg_gemm_big(const GemmArgs g) {
  constexpr int BM = 128, BN = 128, LDS_KS = 128 + 16;
  __shared__ float As[2][A_KC ? BM * LDS_KC : BK * LDS_KS];
  __shared__ float Bs[2][B_KC ? BN * LDS_KC : BK * LDS_KS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int zb = blockIdx.z / g.ksplit, sp = blockIdx.z - zb * g.ksplit;
  const GemmBatch bt = g.batches[zb];
  const int nk = g.K / BK;
  const int all = (bt.seg_end - bt.seg_begin) * nk, per = (all + g.ksplit - 1) / g.ksplit;
  const int it0 = sp * per, it1 = min(all, it0 + per), total = max(it1 - it0, 0);
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int PD = 4;
  f32x4 ra[PD][2], rb[PD][2];
  auto gload = [&](int itr, int slot) {
    const int it = it0 + itr;
    const int s = bt.seg_begin + it / nk, k0 = (it % nk) * BK;
    const GemmSeg sg = g.segs[s];
    const float* Ap = ((sg.a_off & GEMM_ALT) ? g.A2 : g.A) + (sg.a_off & ~GEMM_ALT);
    const float* Bp = g.B + sg.b_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (A_KC) {
        int row = m0 + (tid >> 3) + 64 * i;
        row = row < g.M ? row : g.M - 1;
        ra[slot][i] = *reinterpret_cast<const f32x4*>(Ap + (size_t)row * g.lda + k0 + (tid & 7) * 4);
      } else {
        const int k = (tid >> 5) + 16 * i;
        int col = m0 + (tid & 31) * 4;
        col = col < g.M ? col : g.M - 4;
        ra[slot][i] = *reinterpret_cast<const f32x4*>(Ap + (size_t)(k0 + k) * g.lda + col);
      }
      if (B_KC) {
        int row = n0 + (tid >> 3) + 64 * i;
        row = row < g.N ? row : g.N - 1;
        rb[slot][i] = *reinterpret_cast<const f32x4*>(Bp + (size_t)row * g.ldb + k0 + (tid & 7) * 4);
      } else {
        const int k = (tid >> 5) + 16 * i;
        int col = n0 + (tid & 31) * 4;
        col = col < g.N ? col : g.N - 4;
        rb[slot][i] = *reinterpret_cast<const f32x4*>(Bp + (size_t)(k0 + k) * g.ldb + col);
      }
    }
  };
  auto lstore = [&](int buf, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (A_KC) *reinterpret_cast<f32x4*>(&As[buf][((tid >> 3) + 64 * i) * LDS_KC + (tid & 7) * 4]) = ra[slot][i];
      else *reinterpret_cast<f32x4*>(&As[buf][((tid >> 5) + 16 * i) * LDS_KS + (tid & 31) * 4]) = ra[slot][i];
      if (B_KC) *reinterpret_cast<f32x4*>(&Bs[buf][((tid >> 3) + 64 * i) * LDS_KC + (tid & 7) * 4]) = rb[slot][i];
      else *reinterpret_cast<f32x4*>(&Bs[buf][((tid >> 5) + 16 * i) * LDS_KS + (tid & 31) * 4]) = rb[slot][i];
    }
  };
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (p < total) gload(p, p);
  if (total > 0) lstore(0, 0);
  __syncthreads();
  auto step = [&](int it, int slot) {
    const int buf = slot & 1;
    if (it + PD < total) gload(it + PD, slot);
    float fa[2][8], fb[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wm * 32 + i * 16 + fr;
        fa[i][e] = A_KC ? As[buf][r * LDS_KC + e * 4 + fk] : As[buf][(e * 4 + fk) * LDS_KS + r];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = wn * 64 + j * 16 + fr;
        fb[j][e] = B_KC ? Bs[buf][c * LDS_KC + e * 4 + fk] : Bs[buf][(e * 4 + fk) * LDS_KS + c];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    if (it + 1 < total) lstore(buf ^ 1, (slot + 1) % PD);
    __syncthreads();
  };
  for (int it = 0; it < total; it += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (it + u < total) step(it + u, u);
  }
  const bool partial = g.ksplit > 1;
  if (partial && g.cnt) {
    const SplitK sk = splitk_open<BM * BN>(g, zb);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) splitk_put(sk, sp, (wave * 2 + i) * 4 + j, acc[i][j]);
    if (!splitk_last(g, sk)) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = splitk_get(sk, 0, (wave * 2 + i) * 4 + j);
        for (int q = 1; q < g.ksplit; ++q) {
          const f32x4 v = splitk_get(sk, q, (wave * 2 + i) * 4 + j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] += v[e];
        }
      }
  }
  const bool direct = !partial || g.cnt;
  const long long coff = bt.c_off & ~GEMM_ALT;
  float* Cp = (direct ? ((bt.c_off & GEMM_ALT) ? g.C2 : g.C) : g.part + (size_t)sp * g.c_extent) + coff;
  const float* Dp = (direct && g.add) ? g.add + coff : nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fr;
    if (n >= g.N) continue;
    const float bv = (direct && g.bias) ? g.bias[bt.bias_off + n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 32 + i * 16 + fk * 4 + e;
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (Dp) v += Dp[(size_t)m * g.ldc + n];
        Cp[(size_t)m * g.ldc + n] = v;
      }
  }
}
// C = sum over the splits (in split order) + bias + add
__global__ void reduce_parts_kernel(const GemmArgs g, int nbatch) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)g.M * g.N;
  if (i >= per * nbatch) return;
  const int z = (int)(i / per);
  const long long r = i - (long long)z * per;
  const int m = (int)(r / g.N), n = (int)(r - (long long)m * g.N);
  const size_t off = (size_t)(g.batches[z].c_off & ~GEMM_ALT) + (size_t)m * g.ldc + n;
  float v = g.part[off];
  for (int sp = 1; sp < g.ksplit; ++sp) v += g.part[(size_t)sp * g.c_extent + off];
  if (g.bias) v += g.bias[g.batches[z].bias_off + n];
  if (g.add) v += g.add[off];
  g.C[off] = v;
}

inline dim3 g1(long long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

enum GemmForm { G_NN = 0, G_NT = 1, G_TN = 2 };

struct GemmTune { int small_wg = 1 << 20, big = 0, split = 1, wg_target = 384, fuse = 1, intra = 0; };

// Launch shape.  The GEMMs of a 256-sample step are small (0.1 .. 2 GFLOP) and often deep (K up to 4096) with few output tiles: what fills the chip
// is splitting K.  128 x 128 tiles (32 FLOP per byte moved into LDS) when both M and N reach 128, else 32 / 64-row x 64 tiles; the K steps of a batch
// are dealt to `ks` work-groups until ~192 of them exist or a split would get fewer than two K steps.
struct GemmShape { bool big; bool small32; int ks; int ki; };      // ks: K split over work-groups; ki: 2 = each work-group is two wave quartets splitting its K steps (seg_gemm KI)
GemmShape gemm_shape(const GemmArgs& g, int nbatch, int min_steps, const GemmTune& tn, bool can_split) {
  GemmShape sh{false, false, 1, 1};
  long long tiles;
  if (tn.big && g.M >= 128 && g.N >= 128) {
    sh.big = true;
    tiles = (long long)((g.N + 127) / 128) * ((g.M + 127) / 128) * nbatch;
  } else {
    const long long wg64 = (long long)((g.N + 63) / 64) * ((g.M + 63) / 64) * nbatch;
    sh.small32 = wg64 < tn.small_wg && g.M % 32 == 0;
    tiles = (long long)((g.N + 63) / 64) * (sh.small32 ? (g.M + 31) / 32 : (g.M + 63) / 64) * nbatch;
  }
  if (can_split && tn.split)
    while (sh.ks < 32 && tiles * sh.ks < tn.wg_target && min_steps / (sh.ks * 2) >= 2) sh.ks *= 2;
  if (tn.intra && !sh.big && sh.ks >= 2) { sh.ki = 2; sh.ks /= 2; }      // the first factor of two inside the work-group: no partial block, no ticket
  return sh;
}

// part / c_extent: workspace for split-K partials (nullptr: never split); min_steps = the fewest K steps (segments x K / 32) any batch of the launch has
constexpr long long CNT_TILES = 1 << 16;     // tickets of one launch (tiles x batches); larger launches fall back to the separate reduce
// workspace of the in-launch finish: ks blocks of one tile per output tile (0: the launch cannot use it -- too many tiles, or offsets beyond 2^31)
long long fused_part_bytes(const GemmArgs& g, int nbatch, const GemmShape& sh) {
  const long long tiles = sh.big ? (long long)((g.N + 127) / 128) * ((g.M + 127) / 128) * nbatch
                                 : (long long)((g.N + 63) / 64) * (sh.small32 ? (g.M + 31) / 32 : (g.M + 63) / 64) * nbatch;
  const long long bytes = tiles * sh.ks * (sh.big ? 128 * 128 : sh.small32 ? 32 * 64 : 64 * 64) * 4;
  return (tiles <= CNT_TILES && bytes < (1ll << 31)) ? bytes : 0;
}
int gemm_launch(GemmForm f, GemmArgs g, int nbatch, hipStream_t s, const GemmTune& tn = GemmTune(), int min_steps = 0, float* part = nullptr,
                long long c_extent = 0, unsigned int* cnt = nullptr) {
  if (nbatch <= 0 || g.M <= 0 || g.N <= 0) return LDP_OK;
  if (g.K % BK || g.M % 4 || g.N % 4 || g.lda % 4 || g.ldb % 4)
    return fail(LDP_EINVAL, "seg_gemm: K = %d must be a multiple of %d and M, N, lda, ldb multiples of 4 (%d, %d, %d, %d)", g.K, BK, g.M, g.N, g.lda, g.ldb);
  const GemmShape sh = gemm_shape(g, nbatch, min_steps, tn, part != nullptr);
  g.ksplit = sh.ks; g.part = part; g.c_extent = c_extent;
  g.cnt = (tn.fuse && sh.ks > 1 && fused_part_bytes(g, nbatch, sh) > 0) ? cnt : nullptr;
  if (sh.ks > 1 && !g.cnt && c_extent == 0) return fail(LDP_EINVAL, "seg_gemm: a launch without a C-layout workspace cannot split K outside the in-launch finish");
  if (sh.big) {
    dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, nbatch * sh.ks);
    if (f == G_NN) hipLaunchKernelGGL((seg_gemm_big<true, false>), grid, dim3(512), 0, s, g);
    else if (f == G_NT) hipLaunchKernelGGL((seg_gemm_big<true, true>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((seg_gemm_big<false, false>), grid, dim3(512), 0, s, g);
  } else {
    dim3 grid((g.N + 63) / 64, sh.small32 ? (g.M + 31) / 32 : (g.M + 63) / 64, nbatch * sh.ks);
#define LDP_SEG_LAUNCH(A, B, MT_)                                                                                                     \
  do {                                                                                                                                \
    if (sh.ki == 2) hipLaunchKernelGGL((seg_gemm<A, B, MT_, 2>), grid, dim3(512), 0, s, g);                                           \
    else hipLaunchKernelGGL((seg_gemm<A, B, MT_, 1>), grid, dim3(256), 0, s, g);                                                      \
  } while (0)
    if (sh.small32) {
      if (f == G_NN) LDP_SEG_LAUNCH(true, false, 1);
      else if (f == G_NT) LDP_SEG_LAUNCH(true, true, 1);
      else LDP_SEG_LAUNCH(false, false, 1);
    } else {
      if (f == G_NN) LDP_SEG_LAUNCH(true, false, 2);
      else if (f == G_NT) LDP_SEG_LAUNCH(true, true, 2);
      else LDP_SEG_LAUNCH(false, false, 2);
    }
#undef LDP_SEG_LAUNCH
  }
  if (sh.ks > 1 && !g.cnt) hipLaunchKernelGGL(reduce_parts_kernel, g1((long long)g.M * g.N * nbatch), dim3(256), 0, s, g, nbatch);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// =====================================================================================================================
// element-wise kernels (HBM-bound; IEEE exp / divide: these are not the issue-bound MFMA epilogues of the sampling path)
// =====================================================================================================================
__device__ __forceinline__ float mish_x(float x) {          // x tanh(softplus(x)) = x n / (n + 2), n = e^x (e^x + 2)
  const float e = expf(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  return x * (n / (n + 2.0f));
}
__device__ __forceinline__ float mish_dx(float x) {         // d/dx: w + x w',  w = n / (n + 2),  w' = 4 e (e + 1) / (n + 2)^2
  const float e = expf(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  const float d = n + 2.0f;
  const float w = n / d;
  return x > 20.0f ? 1.0f : w + x * (4.0f * e * (e + 1.0f)) / (d * d);
}
__device__ __forceinline__ float wsum(float v) {            // wave sum in every lane, fixed order (csrc/tconv.hpp wave_sum's DPP chain)
  return wave_sum(v);
}

// out[r][c] = act(in[r][c])   (act 1 mish, 2 relu); rows x cols with row strides
__global__ void act_fwd_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, int rows, int cols, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  const float x = in[(size_t)r * ldi + c];
  out[(size_t)r * ldo + c] = act == 1 ? mish_x(x) : fmaxf(x, 0.0f);
}
// din[r][c] = dout[r][c] * act'(pre[r][c])
__global__ void act_bwd_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ pre, int ldp, float* __restrict__ din, int ldi,
                               int rows, int cols, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  const float x = pre[(size_t)r * ldp + c];
  const float d = dout[(size_t)r * ldd + c];
  din[(size_t)r * ldi + c] = act == 1 ? d * mish_dx(x) : (x > 0.0f ? d : 0.0f);
}
// dst[r][c0 + c] = src[r][c]   (column-block copy between strided matrices; src == nullptr: zero fill)
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows, int cols) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  dst[(size_t)r * ldd + c] = src ? src[(size_t)r * lds + c] : 0.0f;
}
// dst = a + b (same dense layout)
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = a[i] + b[i];
}
// emb[r][:] = table[t[r]][:]  (rows >= n_rows: zero)
__global__ void gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ t, float* __restrict__ out, int ldo, int n_rows, int rows_p,
                                   int width) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows_p * width) return;
  const int r = (int)(i / width), c = (int)(i - (long long)r * width);
  out[(size_t)r * ldo + c] = r < n_rows ? table[(size_t)t[r] * width + c] : 0.0f;
}
// FlaxDDPMScheduler.add_noise into a padded buffer: out[r][c] = sqrt(abar[t]) x0 + sqrt(1 - abar[t]) noise for c < width, 0 in the padding;
// `per` = rows sharing one timestep (T for the planner, 1 for the IDM).  Also copies the noise into its padded twin.
__global__ void add_noise_pad_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int* __restrict__ t, AbarTable tab,
                                     float* __restrict__ out, float* __restrict__ noise_p, int ldo, int rows, int rows_p, int width, int width_p, int per) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows_p * width_p) return;
  const int r = (int)(i / width_p), c = (int)(i - (long long)r * width_p);
  float o = 0.0f, z = 0.0f;
  if (r < rows && c < width) {
    const float a = tab.v[t[r / per]];
    z = noise[(size_t)r * width + c];
    o = sqrtf(a) * x0[(size_t)r * width + c] + sqrtf(1.0f - a) * z;
  }
  out[(size_t)r * ldo + c] = o;
  if (noise_p) noise_p[(size_t)r * width_p + c] = z;
}
// MSE loss and its gradient: dpred[r][c] = scale * (pred - noise) on the real entries, 0 in the padding; partial sums of squares per block
__global__ void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ noise_p, float* __restrict__ dpred, float* __restrict__ part,
                                int rows, int rows_p, int width, int width_p, float scale) {
  __shared__ float red[4];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float sq = 0.0f;
  if (i < (long long)rows_p * width_p) {
    const int r = (int)(i / width_p), c = (int)(i - (long long)r * width_p);
    float d = 0.0f;
    if (r < rows && c < width) {
      d = pred[i] - noise_p[i];
      sq = d * d;
    }
    dpred[i] = scale * d;
  }
  sq = wsum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[0] = alpha * (sum of part[0..n)) / count      (one wave, fixed order)
__global__ void finish_loss_kernel(const float* __restrict__ part, int n, float alpha, float count, float* __restrict__ out) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wsum(s);
  if (threadIdx.x == 0) out[0] = alpha * (s / count);
}

// ---- GroupNorm (+ Mish, FiLM, residual) over (B, T, C) channels-last: one wave per (sample, group), lane = channel of the group ----------
// stats[(b * G + g) * 2] = {mean, rstd};  y = mish(gn(c) * gamma + beta) [* emb[b][ch] + emb[b][C + ch]] [+ res]
__global__ __launch_bounds__(256) void gn_fwd_kernel(const float* __restrict__ c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ emb, const float* __restrict__ res, float* __restrict__ y,
                                                     float* __restrict__ stats, int Bp, int T, int C, int G, int lde) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= Bp * G) return;
  const int b = w / G, g = w - b * G, cg = C / G;
  const float* cb = c + (size_t)b * T * C + g * cg;
  float s1 = 0.0f, s2 = 0.0f;
  for (int t = 0; t < T; ++t)
    for (int ch = lane; ch < cg; ch += 64) {
      const float v = cb[(size_t)t * C + ch];
      s1 += v;
      s2 += v * v;
    }
  s1 = wsum(s1);
  s2 = wsum(s2);
  const float inv = 1.0f / (float)(T * cg);
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.0f);          // flax GroupNorm: use_fast_variance
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  if (lane == 0) {
    stats[(size_t)w * 2] = mean;
    stats[(size_t)w * 2 + 1] = rstd;
  }
  float* yb = y + (size_t)b * T * C + g * cg;
  const float* rb = res ? res + (size_t)b * T * C + g * cg : nullptr;
  for (int ch = lane; ch < cg; ch += 64) {
    const float ga = gamma[g * cg + ch], be = beta[g * cg + ch];
    float sc = 1.0f, sh = 0.0f;
    if (emb) {
      sc = emb[(size_t)b * lde + g * cg + ch];
      sh = emb[(size_t)b * lde + C + g * cg + ch];
    }
    for (int t = 0; t < T; ++t) {
      const float n = (cb[(size_t)t * C + ch] - mean) * rstd * ga + be;
      float o = mish_x(n);
      if (emb) o = sc * o + sh;
      if (rb) o += rb[(size_t)t * C + ch];
      yb[(size_t)t * C + ch] = o;
    }
  }
}
// backward of the above w.r.t. c (conv output incl. bias), gamma, beta, emb:
//   dm = dy * scale;  dn = dm * mish'(n);  dxh = dn * gamma;  dc = rstd (dxh - mean_g(dxh) - xh mean_g(dxh xh))
//   per-sample partials: pg[b][ch] = sum_t dn xh, pb[b][ch] = sum_t dn, pc[b][ch] = sum_t dc;  demb[b][ch] = sum_t dy m, demb[b][C + ch] = sum_t dy
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ c, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ emb,
                                                     float* __restrict__ dc, float* __restrict__ part, float* __restrict__ demb, int Bp, int T, int C,
                                                     int G, int lde) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= Bp * G) return;
  const int b = w / G, g = w - b * G, cg = C / G;
  const size_t base = (size_t)b * T * C + g * cg;
  const float mean = stats[(size_t)w * 2], rstd = stats[(size_t)w * 2 + 1];
  float a1 = 0.0f, a2 = 0.0f;
  for (int ch = lane; ch < cg; ch += 64) {
    const float ga = gamma[g * cg + ch], be = beta[g * cg + ch];
    const float sc = emb ? emb[(size_t)b * lde + g * cg + ch] : 1.0f;
    float sg = 0.0f, sb = 0.0f, se = 0.0f, sd = 0.0f;
    for (int t = 0; t < T; ++t) {
      const float xh = (c[base + (size_t)t * C + ch] - mean) * rstd;
      const float n = xh * ga + be;
      const float d = dy[base + (size_t)t * C + ch];
      const float dn = d * sc * mish_dx(n);
      const float dxh = dn * ga;
      a1 += dxh;
      a2 += dxh * xh;
      sg += dn * xh;
      sb += dn;
      if (emb) {
        se += d * mish_x(n);
        sd += d;
      }
    }
    part[(size_t)b * 3 * C + g * cg + ch] = sg;
    part[(size_t)b * 3 * C + C + g * cg + ch] = sb;
    if (emb) {
      demb[(size_t)b * lde + g * cg + ch] = se;
      demb[(size_t)b * lde + C + g * cg + ch] = sd;
    }
  }
  a1 = wsum(a1);
  a2 = wsum(a2);
  const float inv = 1.0f / (float)(T * cg);
  const float m1 = a1 * inv, m2 = a2 * inv;
  for (int ch = lane; ch < cg; ch += 64) {
    const float ga = gamma[g * cg + ch], be = beta[g * cg + ch];
    const float sc = emb ? emb[(size_t)b * lde + g * cg + ch] : 1.0f;
    float sdc = 0.0f;
    for (int t = 0; t < T; ++t) {
      const float xh = (c[base + (size_t)t * C + ch] - mean) * rstd;
      const float n = xh * ga + be;
      const float dxh = dy[base + (size_t)t * C + ch] * sc * mish_dx(n) * ga;
      const float v = rstd * ((dxh - m1) - xh * m2);
      dc[base + (size_t)t * C + ch] = v;
      sdc += v;
    }
    part[(size_t)b * 3 * C + 2 * C + g * cg + ch] = sdc;
  }
}

// Fast paths of the two GroupNorm kernels for (sample, group) blocks of exactly 256 values -- every level of the planner U-Net: 8 x 32, 4 x 64, 2 x 128
// (positions x channels per group).  Element e = lane + 64 i (i < 4) of the block: position e / CG, channel e % CG -- all 64 lanes busy at every width
// (the generic kernels walk channels by lane: half a wave at CG = 32), every operand requested up front, ONE pass over the tensor (the values stay
// in registers between the statistics and the output), per-channel sums inside the lane (CG = 64: its four values are one channel; CG = 128: two
// channels) or with one 32-lane exchange (CG = 32).
template <int CG>
__global__ __launch_bounds__(256) void gn_fwd4_kernel(const float* __restrict__ c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ emb, const float* __restrict__ res, float* __restrict__ y,
                                                      float* __restrict__ stats, int Bp, int C, int G, int lde) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= Bp * G) return;
  const int b = w / G, g = w - b * G;
  constexpr int T = 256 / CG;
  const size_t base = (size_t)b * T * C + (size_t)g * CG;
  float v[4], ga[4], be[4], sc[4], sh[4], r[4];
  size_t off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = lane + 64 * i, t = e / CG, ch = e % CG;
    off[i] = base + (size_t)t * C + ch;
    v[i] = c[off[i]];
    ga[i] = gamma[g * CG + ch];
    be[i] = beta[g * CG + ch];
    sc[i] = emb ? emb[(size_t)b * lde + g * CG + ch] : 1.0f;
    sh[i] = emb ? emb[(size_t)b * lde + C + g * CG + ch] : 0.0f;
    r[i] = res ? res[off[i]] : 0.0f;
  }
  float s1 = (v[0] + v[1]) + (v[2] + v[3]);
  float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  s1 = wsum(s1);
  s2 = wsum(s2);
  const float inv = 1.0f / 256.0f;
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.0f);          // flax GroupNorm: use_fast_variance
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  if (lane == 0) {
    stats[(size_t)w * 2] = mean;
    stats[(size_t)w * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float n = (v[i] - mean) * rstd * ga[i] + be[i];
    float o = mish_x(n);
    if (emb) o = sc[i] * o + sh[i];
    if (res) o += r[i];
    y[off[i]] = o;
  }
}
template <int CG>
__global__ __launch_bounds__(256) void gn_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ c, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ emb,
                                                      float* __restrict__ dc, float* __restrict__ part, float* __restrict__ demb, int Bp, int C, int G,
                                                      int lde) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= Bp * G) return;
  const int b = w / G, g = w - b * G;
  constexpr int T = 256 / CG;
  const size_t base = (size_t)b * T * C + (size_t)g * CG;
  const float mean = stats[(size_t)w * 2], rstd = stats[(size_t)w * 2 + 1];
  float xh[4], d[4], ga[4], be[4], sc[4];
  size_t off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = lane + 64 * i, t = e / CG, ch = e % CG;
    off[i] = base + (size_t)t * C + ch;
    xh[i] = c[off[i]];
    d[i] = dy[off[i]];
    ga[i] = gamma[g * CG + ch];
    be[i] = beta[g * CG + ch];
    sc[i] = emb ? emb[(size_t)b * lde + g * CG + ch] : 1.0f;
  }
  float dxh[4], tg[4], tb[4], te[4], td[4];
  float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xh[i] = (xh[i] - mean) * rstd;
    const float n = xh[i] * ga[i] + be[i];
    const float dn = d[i] * sc[i] * mish_dx(n);
    dxh[i] = dn * ga[i];
    a1 += dxh[i];
    a2 += dxh[i] * xh[i];
    tg[i] = dn * xh[i];
    tb[i] = dn;
    te[i] = emb ? d[i] * mish_x(n) : 0.0f;
    td[i] = d[i];
  }
  a1 = wsum(a1);
  a2 = wsum(a2);
  const float m1 = a1 * (1.0f / 256.0f), m2 = a2 * (1.0f / 256.0f);
  float tc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    tc[i] = rstd * ((dxh[i] - m1) - xh[i] * m2);
    dc[off[i]] = tc[i];
  }
  // per-channel sums over the positions: the lane's own values of the channel, then (CG = 32) the lane 32 away
  auto put = [&](const float (&q)[4], float* dst0, size_t row, int colbase) {
    if (CG == 128) {                                         // channels lane (i = 0, 2) and lane + 64 (i = 1, 3)
      dst0[row + colbase + lane] = q[0] + q[2];
      dst0[row + colbase + lane + 64] = q[1] + q[3];
    } else {
      float sum = (q[0] + q[1]) + (q[2] + q[3]);
      if (CG == 32) {
        sum += __shfl_xor(sum, 32);
        if (lane < 32) dst0[row + colbase + lane] = sum;
      } else {
        dst0[row + colbase + lane] = sum;
      }
    }
  };
  const size_t prow = (size_t)b * 3 * C;
  put(tg, part, prow, g * CG);
  put(tb, part, prow, C + g * CG);
  put(tc, part, prow, 2 * C + g * CG);
  if (emb) {
    const size_t erow = (size_t)b * lde;
    put(te, demb, erow, g * CG);
    put(td, demb, erow, C + g * CG);
  }
}

// ---- LayerNorm over rows of width H (one wave per row, 4 rows per work-group) ------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ stats, int rows, int H) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * H;
  float s1 = 0.0f, s2 = 0.0f;
  for (int c = lane; c < H; c += 64) {
    const float v = xr[c];
    s1 += v;
    s2 += v * v;
  }
  s1 = wsum(s1);
  s2 = wsum(s2);
  const float mean = s1 / (float)H;
  const float var = fmaxf(s2 / (float)H - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + 1e-6f);
  if (lane == 0) {
    stats[(size_t)r * 2] = mean;
    stats[(size_t)r * 2 + 1] = rstd;
  }
  for (int c = lane; c < H; c += 64) y[(size_t)r * H + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}
// dx = rstd (dxh - mean(dxh) - xh mean(dxh xh)) (+ dres: the residual branch's gradient);  partials part[r][c] = dy xh, part[r][H + c] = dy
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ part, int rows, int H) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float mean = stats[(size_t)r * 2], rstd = stats[(size_t)r * 2 + 1];
  float a1 = 0.0f, a2 = 0.0f;
  for (int c = lane; c < H; c += 64) {
    const float xh = (x[(size_t)r * H + c] - mean) * rstd;
    const float d = dy[(size_t)r * H + c];
    const float dxh = d * gamma[c];
    a1 += dxh;
    a2 += dxh * xh;
    part[(size_t)r * 2 * H + c] = d * xh;
    part[(size_t)r * 2 * H + H + c] = d;
  }
  a1 = wsum(a1);
  a2 = wsum(a2);
  const float m1 = a1 / (float)H, m2 = a2 / (float)H;
  for (int c = lane; c < H; c += 64) {
    const float xh = (x[(size_t)r * H + c] - mean) * rstd;
    const float dxh = dy[(size_t)r * H + c] * gamma[c];
    float v = rstd * ((dxh - m1) - xh * m2);
    if (dres) v += dres[(size_t)r * H + c];
    dx[(size_t)r * H + c] = v;
  }
}

// ---- column sums (bias / scale gradients): out[c] = sum_r x[r][c], fixed order ------------------------------------------------------
// The columns may be scattered over up to three gradient leaves (a GroupNorm backward leaves d gamma | d beta | d bias side by side):
// column c goes to o[c / seg][c % seg].
struct ColOut { float* o[3]; int seg; };
// Column sums (bias / GroupNorm / LayerNorm parameter gradients) are DEFERRED: a tape only records them (source, shape, up to three output leaves) and
// one pair of launches at its end sums them all -- 60 jobs of 2 - 3 MB each are 8000 short work-groups in one launch instead of 100 launches of 48
// work-groups (8 us each, latency-bound: 12 % of the step's kernel time).  Stage 1: work-group = (job, 64 columns, 64 rows), four row lanes per
// column -> tmp[job][chunk][column]; stage 2: the chunks in order -> the leaves.  Fixed order, no atomics.
struct ColJob {
  const float* x;
  float* o[3];
  long long tmp_off;               // floats into the workspace
  int ld, rows, cols, seg;
  int wg0, ncb;                    // first stage-1 work-group of the job, its column blocks
  int c0;                          // first stage-2 thread of the job
  int pad;
};
constexpr int COLJOBS = 48, COL_CHUNK = 64;
struct ColJobs { ColJob j[COLJOBS]; int n, pad; };
__device__ __forceinline__ int coljob_of(const ColJobs& js, int id, bool stage2) {
  int k = 0;
  for (int i = 1; i < js.n; ++i) k = (stage2 ? js.j[i].c0 : js.j[i].wg0) <= id ? i : k;      // (uniform in stage 1; n <= 48)
  return k;
}
__global__ __launch_bounds__(256) void colsum_jobs1_kernel(const ColJobs js, float* __restrict__ tmp) {
  __shared__ float red[4][64];
  const int k = coljob_of(js, blockIdx.x, false);
  const ColJob& jb = js.j[k];
  const int local = blockIdx.x - jb.wg0, cb = local % jb.ncb, chunk = local / jb.ncb;
  const int c = cb * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int r0 = chunk * COL_CHUNK, r1 = min(jb.rows, r0 + COL_CHUNK);
  float s = 0.0f;
  if (c < jb.cols)
    for (int r = r0 + q; r < r1; r += 4) s += jb.x[(size_t)r * jb.ld + c];
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < jb.cols) tmp[jb.tmp_off + (size_t)chunk * jb.cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_jobs2_kernel(const ColJobs js, const float* __restrict__ tmp, int total) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int k = coljob_of(js, id, true);
  const ColJob& jb = js.j[k];
  const int c = id - jb.c0;
  const int S = (jb.rows + COL_CHUNK - 1) / COL_CHUNK;
  float s = 0.0f;
  for (int i = 0; i < S; ++i) s += tmp[jb.tmp_off + (size_t)i * jb.cols + c];
  jb.o[c / jb.seg][c % jb.seg] = s;
}

// ---- optimiser ------------------------------------------------------------------------------------------------------------------------
// optax.adam (scale_by_adam, eps_root = 0, then -lr):  mu = (1 - b1) g + b1 mu;  nu = (1 - b2) g^2 + b2 nu;
//   p += -lr * (mu / bc1) / (sqrt(nu / bc2) + eps),  bc = 1 - b^count   (oracle/train.py adam_apply)
// One work-group per 1024-element stripe (sumsq1_kernel's stripes and reduction order): besides the update it leaves the stripe's sum of squared
// gradients in part[block], so that the global norm of the step (a metric: agent/ldp_agent.py:253, nothing is clipped) costs no second pass over the arena.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu, float* __restrict__ nu, long long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2, float* __restrict__ part) {
  __shared__ float red[4];
  const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = i0 + u * 256;
    if (i < n) {
      const float gi = g[i];
      s += gi * gi;
      const float m = (1.0f - b1) * gi + b1 * mu[i];
      const float v = (1.0f - b2) * (gi * gi) + b2 * nu[i];
      mu[i] = m;
      nu[i] = v;
      p[i] = p[i] + (-lr) * ((m / bc1) / (sqrtf(v / bc2) + eps));
    }
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// sum of squares: stage 1 one partial per block (fixed 1024-element stripes), stage 2 one wave
__global__ __launch_bounds__(256) void sumsq1_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = i0 + u * 256;
    if (i < n) s += x[i] * x[i];
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[0] = sqrt(sum over the listed partial arrays)   (double accumulation over at most a few hundred thousand partials, one work-group, fixed order)
__global__ __launch_bounds__(1024) void sumsq2_kernel(const float* __restrict__ pa, long long na, const float* __restrict__ pb, long long nb,
                                                      float* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (long long i = threadIdx.x; i < na; i += 1024) s += (double)pa[i];
  for (long long i = threadIdx.x; i < nb; i += 1024) s += (double)pb[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}


// =====================================================================================================================
// host side: parameter arena, launch tables, the two tapes
// =====================================================================================================================
struct Leaf {
  std::string path;               // "<flax path>/<leaf>" inside the module
  std::vector<int64_t> shape;     // Flax shape
  int taps = 1, rows = 1, cols = 1, rows_p = 1, cols_p = 1;      // view [taps][rows][cols], padded to [taps][rows_p][cols_p]
  size_t off = 0;                 // float offset in the arena
  size_t size_p() const { return (size_t)taps * rows_p * cols_p; }
};

struct Module {
  std::vector<Leaf> leaves;
  std::map<std::string, int> index;
  size_t total = 0;               // floats (each leaf padded to a multiple of 64)
  DevBuf P, G, M, V, gpart;       // params, grads, Adam moments, sum-of-squares partials
  bool gpart_fresh = false;       // gpart holds the stripes' sums of squares of the CURRENT gradients (left by ldp_train_apply)
  long long step = 0;
  bool ready = false;
  int add(const std::string& path, std::vector<int64_t> shape, int rows_p = -1, int cols_p = -1) {
    Leaf l;
    l.path = path;
    l.shape = shape;
    if (shape.size() == 1) { l.cols = (int)shape[0]; }
    else if (shape.size() == 2) { l.rows = (int)shape[0]; l.cols = (int)shape[1]; }
    else { l.taps = (int)shape[0]; l.rows = (int)shape[1]; l.cols = (int)shape[2]; }
    l.rows_p = rows_p < 0 ? l.rows : rows_p;
    l.cols_p = cols_p < 0 ? l.cols : cols_p;
    l.off = total;
    total += (l.size_p() + 63) / 64 * 64;
    index[path] = (int)leaves.size();
    leaves.push_back(l);
    return (int)leaves.size() - 1;
  }
  const Leaf& leaf(const std::string& p) const { return leaves[index.at(p)]; }
  float* p(const std::string& path) const { return P.f() + leaf(path).off; }
  float* g(const std::string& path) const { return G.f() + leaf(path).off; }
};

struct ConvPlan {                  // launch tables of one convolution (device indices into Trainer::d_segs / d_batches)
  int mode = MODE_K5, Tin = 0, Tout = 0, cin = 0, cout = 0, ntaps = 0;
  int f_b0 = 0, f_nb = 0, d_b0 = 0, d_nb = 0, w_b0 = 0, w_nb = 0;      // first batch / batch count of the forward, dgrad, wgrad launches
  int f_minseg = 0, d_minseg = 0, w_minseg = 0;                       // fewest segments any batch of the forward / dgrad / wgrad launch has (split-K sizing)
};

// What one module's tape writes while it runs: the bump-allocated activations, the split-K / column-sum workspaces of its main and of its side stream
// (fork / join below), the side stream itself.  One per module, so that the planner's and the IDM's tapes can be in flight together.
struct Lane {
  static constexpr int NS = 3;     // side streams (opt.train_sides of them are used)
  DevBuf ws;                       // bump-allocated activations
  size_t ws_floats = 0, ws_used = 0;
  DevBuf colsum_tmp, gemm_part, gemm_cnt;      // gemm_cnt: CNT_TILES zeroed tickets (every launch leaves them zero)
  DevBuf colsum_tmp2[NS], gemm_part2[NS], gemm_cnt2[NS];   // the side streams'
  size_t colsum_need = 0, part_need = 0;
  std::vector<ColJob> coljobs;     // the tape's deferred column sums (flush_colsums)
  hipStream_t s2[NS] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> events;
  size_t ev_next = 0, side_next = 0;
  Lane() = default;
  Lane(const Lane&) = delete;
  Lane& operator=(const Lane&) = delete;
  ~Lane() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    for (hipStream_t q : s2)
      if (q) (void)hipStreamDestroy(q);
  }
};

struct ProjPlan { int f_b0 = 0, f_nb = 0, f_minseg = 0, d_b0 = 0, d_nb = 0, d_minseg = 0; };      // plan_proj below

struct Trainer {
  int D = 0, DP = 0, A = 0, AP = 0, G = 0, T = 0, L = 0, E = 0, CP = 0;       // CP = padded width of [temb | cond]
  int IH = 0, INP = 0, NB = 0, TD = 0;                                           // IDM hidden, padded input width, blocks, time dim
  std::vector<int> dims, Tl;
  Module pl, idm;
  // launch tables
  std::vector<GemmSeg> h_segs;
  std::vector<GemmBatch> h_batches;
  DevBuf d_segs, d_batches;
  int plan_B = 0;                  // the batch the conv tables were built for (offsets do not depend on B: built once)
  std::map<std::string, ConvPlan> convs;
  std::map<int, ProjPlan> projs;          // block -> the grouped launches of its two input convolutions (plan_proj)
  int dense_batch = 0;             // a one-segment batch with zero offsets (plain GEMMs)
  // The FiLM Dense layers of all residual blocks of one width read the same conditioning vector: one batched launch per width for the forward
  // (batch = block: weights / bias at the leaf's arena offset, output columns slot * 2C of a (Bp, nb * 2C) group buffer), one for the weight
  // gradients, and ONE segmented-K launch for the data gradient (segment = block: sum over the blocks inside the accumulator).
  struct FilmGroup { int C2 = 0; std::vector<int> blocks; int f_b0 = 0, w_b0 = 0, d_b0 = 0; };
  std::vector<FilmGroup> film;
  std::vector<std::pair<int, int>> film_of;      // block -> (group, slot)
  // tables and workspaces
  DevBuf sintab_p, sintab_i;       // (n_train, E) sin|cos and (n_train, TD) cos|sin
  Lane lane[2];                    // [0] the planner's tape, [1] the IDM's: nothing mutable is shared, the two may be enqueued on different streams
};

Trainer* trainer(ldp_handle* h) { return static_cast<Trainer*>(h->train); }

// ---- tap sets (csrc/tconv.hpp's, restated for the launch tables) -------------------------------------------------------------------
int ntaps_of(int mode) { return mode == MODE_K5 ? 5 : mode == MODE_DOWN ? 3 : mode == MODE_UP ? 4 : 1; }
// input position of tap j at output position to, or -1 when the tap does not contribute there
int tap_in(int mode, int to, int j) {
  switch (mode) {
    case MODE_K5: return to + j - 2;
    case MODE_DOWN: return 2 * to + j;                                       // XLA SAME on an even length: pads (0, 1)
    case MODE_UP: {                                                          // out[2q] = x[q-1] K0 + x[q] K2; out[2q+1] = x[q] K1 + x[q+1] K3
      const int q = to >> 1;
      if ((to & 1) == 0) return j == 0 ? q - 1 : j == 2 ? q : -1;
      return j == 1 ? q : j == 3 ? q + 1 : -1;
    }
    default: return to;
  }
}

ConvPlan plan_conv(Trainer& t, int mode, int Tin, int Tout, int cin, int cout) {
  ConvPlan c;
  c.mode = mode; c.Tin = Tin; c.Tout = Tout; c.cin = cin; c.cout = cout; c.ntaps = ntaps_of(mode);
  const long long wtap = (long long)cin * cout;
  // forward: one batch per output position
  c.f_b0 = (int)t.h_batches.size();
  for (int to = 0; to < Tout; ++to) {
    GemmBatch b{(long long)to * cout, (int)t.h_segs.size(), 0};
    for (int j = 0; j < c.ntaps; ++j) {
      const int ti = tap_in(mode, to, j);
      if (ti >= 0 && ti < Tin) t.h_segs.push_back(GemmSeg{(long long)ti * cin, j * wtap});
    }
    b.seg_end = (int)t.h_segs.size();
    c.f_minseg = to == 0 ? b.seg_end - b.seg_begin : std::min(c.f_minseg, b.seg_end - b.seg_begin);
    t.h_batches.push_back(b);
  }
  c.f_nb = Tout;
  // dgrad: one batch per input position; A = dY at the output positions that read it, B = W[j] read transposed
  c.d_b0 = (int)t.h_batches.size();
  for (int ti = 0; ti < Tin; ++ti) {
    GemmBatch b{(long long)ti * cin, (int)t.h_segs.size(), 0};
    for (int to = 0; to < Tout; ++to)
      for (int j = 0; j < c.ntaps; ++j)
        if (tap_in(mode, to, j) == ti) t.h_segs.push_back(GemmSeg{(long long)to * cout, j * wtap});
    b.seg_end = (int)t.h_segs.size();
    c.d_minseg = ti == 0 ? b.seg_end - b.seg_begin : std::min(c.d_minseg, b.seg_end - b.seg_begin);
    t.h_batches.push_back(b);
  }
  c.d_nb = Tin;
  // wgrad: one batch per tap that is live somewhere; A = X at t_in (transposed read), B = dY at t_out
  c.w_b0 = (int)t.h_batches.size();
  for (int j = 0; j < c.ntaps; ++j) {
    GemmBatch b{j * wtap, (int)t.h_segs.size(), 0};
    for (int to = 0; to < Tout; ++to) {
      const int ti = tap_in(mode, to, j);
      if (ti >= 0 && ti < Tin) t.h_segs.push_back(GemmSeg{(long long)ti * cin, (long long)to * cout});
    }
    b.seg_end = (int)t.h_segs.size();
    if (b.seg_end > b.seg_begin) {
      c.w_minseg = c.w_nb == 0 ? b.seg_end - b.seg_begin : std::min(c.w_minseg, b.seg_end - b.seg_begin);
      t.h_batches.push_back(b);
      ++c.w_nb;
    }
    else t.h_segs.resize(b.seg_begin);
  }
  return c;
}

// A residual block with a 1 x 1 projection runs TWO convolutions over its input (k = 5 -> GroupNorm, 1 x 1 -> the residual) and sums two data
// gradients into it.  One launch each: forward = the k = 5 batches plus one batch per position for the projection (second output base); data gradient
// = the k = 5 segments of an input position plus one segment that reads the projection's dY (second A base).  Offsets into the weights are arena offsets.
ProjPlan plan_proj(Trainer& t, int T, int cin, int cout, long long w5, long long b5, long long w1, long long b1) {
  ProjPlan p;
  const long long wtap = (long long)cin * cout;
  p.f_b0 = (int)t.h_batches.size();
  p.f_minseg = 1;
  for (int to = 0; to < T; ++to) {
    GemmBatch b{(long long)to * cout, (int)t.h_segs.size(), 0, b5};
    for (int j = 0; j < 5; ++j) {
      const int ti = tap_in(MODE_K5, to, j);
      if (ti >= 0 && ti < T) t.h_segs.push_back(GemmSeg{(long long)ti * cin, w5 + j * wtap});
    }
    b.seg_end = (int)t.h_segs.size();
    t.h_batches.push_back(b);
  }
  for (int to = 0; to < T; ++to) {
    GemmBatch b{GEMM_ALT | ((long long)to * cout), (int)t.h_segs.size(), (int)t.h_segs.size() + 1, b1};
    t.h_segs.push_back(GemmSeg{(long long)to * cin, w1});
    t.h_batches.push_back(b);
  }
  p.f_nb = 2 * T;
  p.d_b0 = (int)t.h_batches.size();
  for (int ti = 0; ti < T; ++ti) {
    GemmBatch b{(long long)ti * cin, (int)t.h_segs.size(), 0, 0};
    for (int to = 0; to < T; ++to)
      for (int j = 0; j < 5; ++j)
        if (tap_in(MODE_K5, to, j) == ti) t.h_segs.push_back(GemmSeg{(long long)to * cout, w5 + j * wtap});
    t.h_segs.push_back(GemmSeg{GEMM_ALT | ((long long)ti * cout), w1});
    b.seg_end = (int)t.h_segs.size();
    p.d_minseg = ti == 0 ? b.seg_end - b.seg_begin : std::min(p.d_minseg, b.seg_end - b.seg_begin);
    t.h_batches.push_back(b);
  }
  p.d_nb = T;
  return p;
}

struct Ctx {                        // one enqueue; dry = walk the tape only to size the workspace (nothing is launched); side = on the trainer's side stream
  ldp_handle* h; Trainer* t; Lane* L; hipStream_t s; bool dry; int side = 0;      // side: 0 the caller's stream, k > 0 the lane's side stream k - 1
  const GemmSeg* segs() const { return t->d_segs.as<GemmSeg>(); }
  const GemmBatch* batches() const { return t->d_batches.as<GemmBatch>(); }
};

// Weight-gradient work -- wgrad GEMMs, the column sums behind bias / GroupNorm / LayerNorm parameter gradients -- has no consumer before the optimiser.
// fork(): a context on the trainer's side stream, ordered behind everything the main stream has been given so far (so the dY it reads exists); the
// main stream goes on with the data-gradient chain and the two overlap on the chip.  join(): the main stream waits for the side stream (end of a tape).
// Buffers the side stream reads are never rewritten inside a tape (bump allocation, no reuse); its split-K / column-sum workspaces are its own.
int fork(const Ctx& c, Ctx* out, int bit = 1, int fixed = 0) {          // bit: the feature of train_streams this fork belongs to; fixed > 0: that side stream (else round robin)
  *out = c;
  if (c.dry || !(c.h->opt.train_streams & bit) || c.side) return LDP_OK;
  Lane& t = *c.L;
  const int ns = std::min(std::max(c.h->opt.train_sides, 1), (int)Lane::NS);
  const int k = fixed > 0 ? std::min(fixed, ns) : 1 + (int)(t.side_next++ % ns);
  hipEvent_t ev = t.events[t.ev_next++ % t.events.size()];
  LDP_HIP(hipEventRecord(ev, c.s));
  LDP_HIP(hipStreamWaitEvent(t.s2[k - 1], ev, 0));
  out->s = t.s2[k - 1];
  out->side = k;
  return LDP_OK;
}
// finer than join(): mark() leaves an event behind what the side context has been given so far, wait_for() makes the main stream wait for it
int mark(const Ctx& w, hipEvent_t* ev) {
  *ev = nullptr;
  if (w.dry || !w.side) return LDP_OK;
  Lane& t = *w.L;
  *ev = t.events[t.ev_next++ % t.events.size()];
  LDP_HIP(hipEventRecord(*ev, w.s));
  return LDP_OK;
}
int wait_for(const Ctx& c, hipEvent_t ev) {
  if (ev) LDP_HIP(hipStreamWaitEvent(c.s, ev, 0));
  return LDP_OK;
}
int join(const Ctx& c) {
  if (c.dry || !c.h->opt.train_streams) return LDP_OK;
  Lane& t = *c.L;
  for (hipStream_t q : t.s2) {
    hipEvent_t ev = t.events[t.ev_next++ % t.events.size()];
    LDP_HIP(hipEventRecord(ev, q));
    LDP_HIP(hipStreamWaitEvent(c.s, ev, 0));
  }
  t.side_next = 0;
  return LDP_OK;
}

// c_extent = 0: the launch has no C-layout partial workspace (its outputs are scattered over an arena): K is split only with the in-launch finish
int run_gemm(const Ctx& c, GemmForm f, const GemmArgs& g, int nbatch, int min_steps, long long c_extent) {
  GemmTune tn;
  tn.small_wg = c.h->opt.train_small_wg;
  tn.big = c.h->opt.train_big;
  tn.split = c.h->opt.train_split;
  tn.wg_target = c.h->opt.train_wg_target;
  tn.fuse = c.h->opt.train_fuse_reduce;
  tn.intra = c.h->opt.train_intra_split;
  if (c_extent == 0 && !tn.fuse) tn.split = 0;
  if (c.dry) {
    const GemmShape sh = gemm_shape(g, nbatch, min_steps, tn, true);
    if (sh.ks > 1) c.L->part_need = std::max({c.L->part_need, (size_t)sh.ks * (size_t)c_extent * 4, (size_t)fused_part_bytes(g, nbatch, sh)});
    return LDP_OK;
  }
  static const bool trace = getenv("LDP_TRAIN_TRACE") != nullptr;       // one line per GEMM launch (tools/r6/gemm_table.py joins them with a kernel trace)
  if (trace) {
    const GemmShape sh = gemm_shape(g, nbatch, min_steps, tn, true);
    const long long b0 = g.batches - c.batches();
    long long steps = 0;
    for (int b = 0; b < nbatch; ++b) steps += (long long)(c.t->h_batches[b0 + b].seg_end - c.t->h_batches[b0 + b].seg_begin) * (g.K / BK);
    fprintf(stderr, "LDP_GEMM form=%s M=%d N=%d K=%d nb=%d steps=%lld ks=%d tile=%s gflop=%.4f\n", f == G_NN ? "NN" : f == G_NT ? "NT" : "TN", g.M, g.N, g.K, nbatch,
            steps, sh.ks * sh.ki, sh.big ? "128x128" : sh.small32 ? "32x64" : "64x64", 2.0 * g.M * g.N * BK * steps / 1e9);
  }
  return gemm_launch(f, g, nbatch, c.s, tn, min_steps, (c.side ? c.L->gemm_part2[c.side - 1] : c.L->gemm_part).f(), c_extent,
                     (c.side ? c.L->gemm_cnt2[c.side - 1] : c.L->gemm_cnt).as<unsigned int>());
}
// y (Bp, Tout, cout) = conv(x (Bp, Tin, cin)) + bias
int conv_fwd(const Ctx& c, const ConvPlan& p, const float* x, const float* w, const float* bias, float* y, int Bp) {
  GemmArgs g{x, w, y, bias, nullptr, c.segs(), c.batches() + p.f_b0, Bp, p.cout, p.cin, p.Tin * p.cin, p.cout, p.Tout * p.cout};
  return run_gemm(c, G_NN, g, p.f_nb, p.f_minseg * (p.cin / BK), (long long)Bp * p.Tout * p.cout);
}
// dx (Bp, Tin, cin) = conv^T(dy) (+ add)
int conv_dgrad(const Ctx& c, const ConvPlan& p, const float* dy, const float* w, const float* add, float* dx, int Bp) {
  GemmArgs g{dy, w, dx, nullptr, add, c.segs(), c.batches() + p.d_b0, Bp, p.cin, p.cout, p.Tout * p.cout, p.cout, p.Tin * p.cin};
  return run_gemm(c, G_NT, g, p.d_nb, p.d_minseg * (p.cout / BK), (long long)Bp * p.Tin * p.cin);
}
// dw (taps, cin, cout) = sum over samples and positions of x^T dy   (taps that are dead everywhere keep their zero gradient)
int conv_wgrad(const Ctx& c, const ConvPlan& p, const float* x, const float* dy, float* dw, int Bp) {
  GemmArgs g{x, dy, dw, nullptr, nullptr, c.segs(), c.batches() + p.w_b0, p.cin, p.cout, Bp, p.Tin * p.cin, p.Tout * p.cout, p.cout};
  return run_gemm(c, G_TN, g, p.w_nb, p.w_minseg * (Bp / BK), (long long)p.ntaps * p.cin * p.cout);
}
// plain GEMMs over strided matrices: Y (M, N) = X (M, K) @ W (K, N) + bias (+ add)
int dense_fwd(const Ctx& c, const float* x, int ldx, const float* w, int ldw, const float* bias, const float* add, float* y, int ldy, int M, int K, int N) {
  GemmArgs g{x, w, y, bias, add, c.segs(), c.batches() + c.t->dense_batch, M, N, K, ldx, ldw, ldy};
  return run_gemm(c, G_NN, g, 1, K / BK, (long long)M * ldy);
}
// dX (M, K) = dY (M, N) @ W^T (+ add)      (W (K, N) row-major)
int dense_dgrad(const Ctx& c, const float* dy, int ldy, const float* w, int ldw, const float* add, float* dx, int ldx, int M, int K, int N) {
  GemmArgs g{dy, w, dx, nullptr, add, c.segs(), c.batches() + c.t->dense_batch, M, K, N, ldy, ldw, ldx};
  return run_gemm(c, G_NT, g, 1, N / BK, (long long)M * ldx);
}
// dW (K, N) = X^T (K x M) dY (M, N)
int dense_wgrad(const Ctx& c, const float* x, int ldx, const float* dy, int ldy, float* dw, int ldw, int M, int K, int N) {
  GemmArgs g{x, dy, dw, nullptr, nullptr, c.segs(), c.batches() + c.t->dense_batch, K, N, M, ldx, ldy, ldw};
  return run_gemm(c, G_TN, g, 1, M / BK, (long long)K * ldw);
}
int colsum_to(const Ctx& c, const float* x, int ld, int rows, int cols, const ColOut& out) {
  const size_t floats = (size_t)((rows + COL_CHUNK - 1) / COL_CHUNK) * cols;
  if (c.dry) { c.L->colsum_need += floats * 4; return LDP_OK; }
  ColJob jb{};
  jb.x = x; jb.o[0] = out.o[0]; jb.o[1] = out.o[1]; jb.o[2] = out.o[2];
  jb.ld = ld; jb.rows = rows; jb.cols = cols; jb.seg = out.seg;
  c.L->coljobs.push_back(jb);
  return LDP_OK;
}
// every column sum the tape recorded, in two launches per 48 jobs, on the context's stream (the side stream that ran the tape's tail, or the caller's)
int flush_colsums(const Ctx& c) {
  if (c.dry) return LDP_OK;
  std::vector<ColJob>& all = c.L->coljobs;
  float* tmp = (c.side ? c.L->colsum_tmp2[c.side - 1] : c.L->colsum_tmp).f();
  long long off = 0;
  for (size_t b = 0; b < all.size(); b += COLJOBS) {
    ColJobs js{};
    js.n = (int)std::min<size_t>(COLJOBS, all.size() - b);
    int wg = 0, cols = 0;
    for (int i = 0; i < js.n; ++i) {
      ColJob jb = all[b + i];
      jb.tmp_off = off;
      jb.ncb = (jb.cols + 63) / 64;
      jb.wg0 = wg;
      jb.c0 = cols;
      const int S = (jb.rows + COL_CHUNK - 1) / COL_CHUNK;
      wg += jb.ncb * S;
      cols += jb.cols;
      off += (long long)S * jb.cols;
      js.j[i] = jb;
    }
    hipLaunchKernelGGL(colsum_jobs1_kernel, dim3(wg), dim3(256), 0, c.s, js, tmp);
    hipLaunchKernelGGL(colsum_jobs2_kernel, dim3((cols + 255) / 256), dim3(256), 0, c.s, js, (const float*)tmp, cols);
  }
  all.clear();
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}
int colsum(const Ctx& c, const float* x, int ld, int rows, int cols, float* out) {
  return colsum_to(c, x, ld, rows, cols, ColOut{{out, nullptr, nullptr}, cols});
}

float* ws_take(Lane& t, size_t floats) {
  floats = (floats + 63) / 64 * 64;
  float* p = t.ws.f() + t.ws_used;
  t.ws_used += floats;
  return p;
}

// ---- module descriptions (Flax trees: latent_diffusion_planning_amd/weights.py planner_shapes / idm_shapes) --------------------------------
struct BlockDesc { int cin, cout, T; bool proj; };

void planner_blocks(const Trainer& t, std::vector<BlockDesc>& out) {
  out.clear();
  int cin = t.D;
  for (int l = 0; l < t.L; ++l) {
    out.push_back({cin, t.dims[l], t.Tl[l], true});
    out.push_back({t.dims[l], t.dims[l], t.Tl[l], false});
    cin = t.dims[l];
  }
  out.push_back({cin, cin, t.Tl[t.L - 1], false});
  out.push_back({cin, cin, t.Tl[t.L - 1], false});
  for (int i = 0; i < t.L - 1; ++i) {
    const int c = t.dims[t.L - 2 - i], T = t.Tl[t.L - 1 - i];
    out.push_back({2 * cin, c, T, true});
    out.push_back({c, c, T, false});
    cin = c;
  }
}

void describe_planner(Trainer& t) {
  Module& m = t.pl;
  const int E = t.E, cd = E + t.G;
  m.add("Dense_0/kernel", {E, 4 * E});
  m.add("Dense_0/bias", {4 * E});
  m.add("Dense_1/kernel", {4 * E, E});
  m.add("Dense_1/bias", {E});
  std::vector<BlockDesc> bs;
  planner_blocks(t, bs);
  for (size_t i = 0; i < bs.size(); ++i) {
    const std::string p = "ConditionalResidualBlock1D_" + std::to_string(i);
    const int cin = bs[i].cin, cout = bs[i].cout, cin_p = rup(cin, RP);
    m.add(p + "/Conv1dBlock_0/Conv_0/kernel", {5, cin, cout}, cin_p, cout);
    m.add(p + "/Conv1dBlock_0/Conv_0/bias", {cout});
    m.add(p + "/Conv1dBlock_0/GroupNorm_0/scale", {cout});
    m.add(p + "/Conv1dBlock_0/GroupNorm_0/bias", {cout});
    m.add(p + "/Conv1dBlock_1/Conv_0/kernel", {5, cout, cout});
    m.add(p + "/Conv1dBlock_1/Conv_0/bias", {cout});
    m.add(p + "/Conv1dBlock_1/GroupNorm_0/scale", {cout});
    m.add(p + "/Conv1dBlock_1/GroupNorm_0/bias", {cout});
    m.add(p + "/Dense_0/kernel", {cd, 2 * cout}, t.CP, 2 * cout);
    m.add(p + "/Dense_0/bias", {2 * cout});
    if (bs[i].proj) {
      m.add(p + "/Conv_0/kernel", {1, cin, cout}, cin_p, cout);
      m.add(p + "/Conv_0/bias", {cout});
    }
  }
  for (int l = 0; l + 1 < t.L; ++l) {
    const std::string p = "Downsample1d_" + std::to_string(l) + "/Conv_0";
    m.add(p + "/kernel", {3, t.dims[l], t.dims[l]});
    m.add(p + "/bias", {t.dims[l]});
  }
  for (int i = 0; i + 1 < t.L; ++i) {
    const int c = t.dims[t.L - 2 - i];
    const std::string p = "Upsample1d_" + std::to_string(i) + "/ConvTranspose_0";
    m.add(p + "/kernel", {4, c, c});
    m.add(p + "/bias", {c});
  }
  const int c0 = t.dims[0];
  m.add("Conv1dBlock_0/Conv_0/kernel", {5, c0, c0});
  m.add("Conv1dBlock_0/Conv_0/bias", {c0});
  m.add("Conv1dBlock_0/GroupNorm_0/scale", {c0});
  m.add("Conv1dBlock_0/GroupNorm_0/bias", {c0});
  m.add("Conv_0/kernel", {1, c0, t.D}, c0, t.DP);
  m.add("Conv_0/bias", {t.D}, 1, t.DP);
}

void describe_idm(Trainer& t) {
  Module& m = t.idm;
  const int H = t.IH;
  m.add("MLP_0/Dense_0/kernel", {t.TD, H});
  m.add("MLP_0/Dense_0/bias", {H});
  m.add("MLP_0/Dense_1/kernel", {H, H});
  m.add("MLP_0/Dense_1/bias", {H});
  m.add("MLPResNet_0/Dense_0/kernel", {t.A + 2 * t.D + H, H}, t.INP, H);
  m.add("MLPResNet_0/Dense_0/bias", {H});
  for (int b = 0; b < t.NB; ++b) {
    const std::string p = "MLPResNet_0/MLPResNetBlock_" + std::to_string(b);
    m.add(p + "/LayerNorm_0/scale", {H});
    m.add(p + "/LayerNorm_0/bias", {H});
    m.add(p + "/Dense_0/kernel", {H, 4 * H});
    m.add(p + "/Dense_0/bias", {4 * H});
    m.add(p + "/Dense_1/kernel", {4 * H, H});
    m.add(p + "/Dense_1/bias", {H});
  }
  m.add("MLPResNet_0/Dense_1/kernel", {H, t.A}, H, t.AP);
  m.add("MLPResNet_0/Dense_1/bias", {t.A}, 1, t.AP);
}

int ensure_trainer(ldp_handle* h) {
  if (h->train) return LDP_OK;
  const ldp_config& c = h->cfg;
  Trainer* t = new (std::nothrow) Trainer();
  if (!t) return fail(LDP_ENOMEM, "out of host memory");
  t->D = c.obs_dim; t->DP = rup(c.obs_dim, RP); t->A = c.action_dim; t->AP = rup(c.action_dim, RP);
  t->G = c.global_cond_dim; t->T = c.pred_horizon; t->L = c.n_levels; t->E = c.step_embed_dim;
  t->CP = rup(t->E + t->G, RP);
  for (int l = 0; l < t->L; ++l) { t->dims.push_back(c.down_dims[l]); t->Tl.push_back(c.pred_horizon >> l); }
  t->IH = c.idm_hidden; t->NB = c.idm_blocks; t->TD = c.idm_time_dim;
  t->INP = rup(t->A + 2 * t->D + t->IH, RP);
  describe_planner(*t);
  describe_idm(*t);
  // launch tables: the plain-GEMM batch first, then every convolution of the U-Net
  t->h_segs.push_back(GemmSeg{0, 0});
  t->h_batches.push_back(GemmBatch{0, 0, 1});
  t->dense_batch = 0;
  std::vector<BlockDesc> bs;
  planner_blocks(*t, bs);
  for (size_t i = 0; i < bs.size(); ++i) {
    const std::string p = "b" + std::to_string(i);
    t->convs[p + "c0"] = plan_conv(*t, MODE_K5, bs[i].T, bs[i].T, rup(bs[i].cin, RP), bs[i].cout);
    t->convs[p + "c1"] = plan_conv(*t, MODE_K5, bs[i].T, bs[i].T, bs[i].cout, bs[i].cout);
    if (bs[i].proj) t->convs[p + "r"] = plan_conv(*t, MODE_P1, bs[i].T, bs[i].T, rup(bs[i].cin, RP), bs[i].cout);
  }
  for (int l = 0; l + 1 < t->L; ++l) t->convs["down" + std::to_string(l)] = plan_conv(*t, MODE_DOWN, t->Tl[l], t->Tl[l + 1], t->dims[l], t->dims[l]);
  for (int i = 0; i + 1 < t->L; ++i) {
    const int lv = t->L - 1 - i;
    t->convs["up" + std::to_string(i)] = plan_conv(*t, MODE_UP, t->Tl[lv], t->Tl[lv - 1], t->dims[lv - 1], t->dims[lv - 1]);
  }
  t->convs["fin"] = plan_conv(*t, MODE_K5, t->T, t->T, t->dims[0], t->dims[0]);
  t->convs["out"] = plan_conv(*t, MODE_P1, t->T, t->T, t->dims[0], t->DP);
  for (size_t i = 0; i < bs.size(); ++i) {
    if (!bs[i].proj) continue;
    const std::string p = "ConditionalResidualBlock1D_" + std::to_string(i);
    t->projs[(int)i] = plan_proj(*t, bs[i].T, rup(bs[i].cin, RP), bs[i].cout, (long long)t->pl.leaf(p + "/Conv1dBlock_0/Conv_0/kernel").off,
                                 (long long)t->pl.leaf(p + "/Conv1dBlock_0/Conv_0/bias").off, (long long)t->pl.leaf(p + "/Conv_0/kernel").off,
                                 (long long)t->pl.leaf(p + "/Conv_0/bias").off);
  }
  t->film_of.assign(bs.size(), {0, 0});
  for (size_t i = 0; i < bs.size(); ++i) {
    size_t g = 0;
    while (g < t->film.size() && t->film[g].C2 != 2 * bs[i].cout) ++g;
    if (g == t->film.size()) { t->film.emplace_back(); t->film[g].C2 = 2 * bs[i].cout; }
    t->film_of[i] = {(int)g, (int)t->film[g].blocks.size()};
    t->film[g].blocks.push_back((int)i);
  }
  for (Trainer::FilmGroup& fg : t->film) {
    const int nb = (int)fg.blocks.size();
    auto woff = [&](int slot) { return (long long)t->pl.leaf("ConditionalResidualBlock1D_" + std::to_string(fg.blocks[slot]) + "/Dense_0/kernel").off; };
    auto boff = [&](int slot) { return (long long)t->pl.leaf("ConditionalResidualBlock1D_" + std::to_string(fg.blocks[slot]) + "/Dense_0/bias").off; };
    fg.f_b0 = (int)t->h_batches.size();
    for (int k = 0; k < nb; ++k) {
      GemmBatch b{(long long)k * fg.C2, (int)t->h_segs.size(), (int)t->h_segs.size() + 1, boff(k)};
      t->h_segs.push_back(GemmSeg{0, woff(k)});
      t->h_batches.push_back(b);
    }
    fg.w_b0 = (int)t->h_batches.size();
    for (int k = 0; k < nb; ++k) {
      GemmBatch b{woff(k), (int)t->h_segs.size(), (int)t->h_segs.size() + 1, 0};
      t->h_segs.push_back(GemmSeg{0, (long long)k * fg.C2});
      t->h_batches.push_back(b);
    }
    fg.d_b0 = (int)t->h_batches.size();
    GemmBatch b{0, (int)t->h_segs.size(), (int)t->h_segs.size() + nb, 0};
    for (int k = 0; k < nb; ++k) t->h_segs.push_back(GemmSeg{(long long)k * fg.C2, woff(k)});
    t->h_batches.push_back(b);
  }
  int r = upload(t->d_segs, t->h_segs.data(), t->h_segs.size() * sizeof(GemmSeg), nullptr);
  if (r == LDP_OK) r = upload(t->d_batches, t->h_batches.data(), t->h_batches.size() * sizeof(GemmBatch), nullptr);
  std::vector<float> tab;
  if (r == LDP_OK) { sinusoid_table(c.planner_train_steps, t->E, false, tab); r = upload(t->sintab_p, tab.data(), tab.size() * 4, nullptr); }
  if (r == LDP_OK) { sinusoid_table(c.idm_train_steps, t->TD, true, tab); r = upload(t->sintab_i, tab.data(), tab.size() * 4, nullptr); }
  if (r != LDP_OK) { delete t; return r; }
  h->train = t;
  return LDP_OK;
}

Module* module_of(ldp_handle* h, int32_t module, const char** prefix) {
  Trainer* t = trainer(h);
  if (module == 1) { if (prefix) *prefix = "planner/"; return &t->pl; }
  if (module == 2) { if (prefix) *prefix = "idm/"; return &t->idm; }
  return nullptr;
}

// Flax leaf (host) <-> padded arena image (host)
void pack_leaf(const Leaf& l, const float* src, float* dst) {
  for (int j = 0; j < l.taps; ++j)
    for (int r = 0; r < l.rows; ++r)
      std::memcpy(dst + ((size_t)j * l.rows_p + r) * l.cols_p, src + ((size_t)j * l.rows + r) * l.cols, (size_t)l.cols * 4);
}
void unpack_leaf(const Leaf& l, const float* src, float* dst) {
  for (int j = 0; j < l.taps; ++j)
    for (int r = 0; r < l.rows; ++r)
      std::memcpy(dst + ((size_t)j * l.rows + r) * l.cols, src + ((size_t)j * l.rows_p + r) * l.cols_p, (size_t)l.cols * 4);
}


#define TK(kernel, grid, block, ...) do { if (!c.dry) hipLaunchKernelGGL(kernel, grid, block, 0, c.s, __VA_ARGS__); } while (0)

AbarTable abar_of(int n_train) {
  std::vector<float> betas, alphas, acp;
  betas_squaredcos(n_train, betas, alphas, acp);
  AbarTable tab{};
  std::copy(acp.begin(), acp.end(), tab.v);
  return tab;
}

int act_fwd(const Ctx& c, const float* in, int ldi, float* out, int ldo, int rows, int cols, int act) {
  TK(act_fwd_kernel, g1((long long)rows * cols), dim3(256), in, ldi, out, ldo, rows, cols, act);
  return LDP_OK;
}
int act_bwd(const Ctx& c, const float* dout, int ldd, const float* pre, int ldp, float* din, int ldi, int rows, int cols, int act) {
  TK(act_bwd_kernel, g1((long long)rows * cols), dim3(256), dout, ldd, pre, ldp, din, ldi, rows, cols, act);
  return LDP_OK;
}
int copy_cols(const Ctx& c, const float* src, int lds, float* dst, int ldd, int rows, int cols) {
  TK(copy_cols_kernel, g1((long long)rows * cols), dim3(256), src, lds, dst, ldd, rows, cols);
  return LDP_OK;
}
// the three per-sample partial sums a GroupNorm backward leaves (part (Bp, 3C): d gamma | d beta | d conv-bias) -> the three gradient leaves
int gn_param_grads(const Ctx& c, const float* part, int Bp, int C, float* dgamma, float* dbeta, float* dbias) {
  return colsum_to(c, part, 3 * C, Bp, 3 * C, ColOut{{dgamma, dbeta, dbias}, C});
}

// GroupNorm forward / backward launches: the 256-value fast paths where a (sample, group) block has exactly 256 values, else the generic kernels
int gn_fwd(const Ctx& c, const float* x, const float* gamma, const float* beta, const float* emb, const float* res, float* y, float* stats, int Bp, int T, int C,
           int G, int lde) {
  if (c.dry) return LDP_OK;
  const int cg = C / G;
  const dim3 grid((Bp * G + 3) / 4), blk(256);
  const bool fast = c.h->opt.train_gn4 && T * cg == 256 && C % G == 0;
  if (fast && cg == 32) hipLaunchKernelGGL(gn_fwd4_kernel<32>, grid, blk, 0, c.s, x, gamma, beta, emb, res, y, stats, Bp, C, G, lde);
  else if (fast && cg == 64) hipLaunchKernelGGL(gn_fwd4_kernel<64>, grid, blk, 0, c.s, x, gamma, beta, emb, res, y, stats, Bp, C, G, lde);
  else if (fast && cg == 128) hipLaunchKernelGGL(gn_fwd4_kernel<128>, grid, blk, 0, c.s, x, gamma, beta, emb, res, y, stats, Bp, C, G, lde);
  else hipLaunchKernelGGL(gn_fwd_kernel, grid, blk, 0, c.s, x, gamma, beta, emb, res, y, stats, Bp, T, C, G, lde);
  return LDP_OK;
}
int gn_bwd(const Ctx& c, const float* dy, const float* x, const float* stats, const float* gamma, const float* beta, const float* emb, float* dc, float* part,
           float* demb, int Bp, int T, int C, int G, int lde) {
  if (c.dry) return LDP_OK;
  const int cg = C / G;
  const dim3 grid((Bp * G + 3) / 4), blk(256);
  const bool fast = c.h->opt.train_gn4 && T * cg == 256 && C % G == 0;
  if (fast && cg == 32) hipLaunchKernelGGL(gn_bwd4_kernel<32>, grid, blk, 0, c.s, dy, x, stats, gamma, beta, emb, dc, part, demb, Bp, C, G, lde);
  else if (fast && cg == 64) hipLaunchKernelGGL(gn_bwd4_kernel<64>, grid, blk, 0, c.s, dy, x, stats, gamma, beta, emb, dc, part, demb, Bp, C, G, lde);
  else if (fast && cg == 128) hipLaunchKernelGGL(gn_bwd4_kernel<128>, grid, blk, 0, c.s, dy, x, stats, gamma, beta, emb, dc, part, demb, Bp, C, G, lde);
  else hipLaunchKernelGGL(gn_bwd_kernel, grid, blk, 0, c.s, dy, x, stats, gamma, beta, emb, dc, part, demb, Bp, T, C, G, lde);
  return LDP_OK;
}

struct BlockSave {                 // what a ConditionalResidualBlock1D keeps for its backward
  const float* x = nullptr;        // (Bp, T, cin_p) block input
  float *c0 = nullptr, *f = nullptr, *c1 = nullptr, *res = nullptr, *out = nullptr, *emb = nullptr, *demb = nullptr, *st0 = nullptr, *st1 = nullptr;
};

// ---- planner: loss + gradients (agent/ldp_agent.py:113-127; networks/diffusion_nets_v2.py:66-169) ----------------------------------
int planner_tape(Ctx& c, const float* x0, const float* noise, const int* tdev, const float* cond, float alpha, float* loss_out, int B) {
  Trainer& t = *c.t;
  Module& m = t.pl;
  const int Bp = rup(B, RP), T = t.T, DP = t.DP, E = t.E, CP = t.CP, NG = c.h->cfg.n_groups;
  std::vector<BlockDesc> bs;
  planner_blocks(t, bs);
  const int nblk = (int)bs.size();
  c.L->ws_used = 0;
  auto P = [&](const std::string& path) { return m.P.f() + m.leaf(path).off; };
  auto Gd = [&](const std::string& path) { return m.G.f() + m.leaf(path).off; };
  auto take = [&](size_t n) { return ws_take(*c.L, n); };

  // ---- forward --------------------------------------------------------------------------------------------------------------------
  float* xn = take((size_t)Bp * T * DP);
  float* nz = take((size_t)Bp * T * DP);
  TK(add_noise_pad_kernel, g1((long long)Bp * T * DP), dim3(256), x0, noise, tdev, abar_of(c.h->cfg.planner_train_steps), xn, nz, DP, B * T, Bp * T, t.D, DP, T);
  float* semb = take((size_t)Bp * E);
  TK(gather_rows_kernel, g1((long long)Bp * E), dim3(256), t.sintab_p.f(), tdev, semb, E, B, Bp, E);
  float* d0 = take((size_t)Bp * 4 * E);
  float* md0 = take((size_t)Bp * 4 * E);
  LDP_TRY(dense_fwd(c, semb, E, P("Dense_0/kernel"), 4 * E, P("Dense_0/bias"), nullptr, d0, 4 * E, Bp, E, 4 * E));
  LDP_TRY(act_fwd(c, d0, 4 * E, md0, 4 * E, Bp, 4 * E, 1));
  float* gbuf = take((size_t)Bp * CP);
  float* gm = take((size_t)Bp * CP);
  LDP_TRY(copy_cols(c, nullptr, 0, gbuf, CP, Bp, CP));
  LDP_TRY(dense_fwd(c, md0, 4 * E, P("Dense_1/kernel"), E, P("Dense_1/bias"), nullptr, gbuf, CP, Bp, 4 * E, E));
  if (t.G > 0) LDP_TRY(copy_cols(c, cond, t.G, gbuf + E, CP, B, t.G));
  LDP_TRY(act_fwd(c, gbuf, CP, gm, CP, Bp, CP, 1));

  std::vector<BlockSave> sv(nblk);
  // FiLM parameters of every block: they depend on the conditioning vector only.  One batched launch per block width, on the side stream, ahead of
  // the first convolutions; the main stream waits for a width's launch at the first block that needs it.
  const int nfg = (int)t.film.size();
  std::vector<float*> embG(nfg), dembG(nfg);
  std::vector<int> ldE(nfg);
  std::vector<hipEvent_t> film_ready(nfg, nullptr);
  std::vector<char> film_waited(nfg, 0);
  {
    Ctx w;
    LDP_TRY(fork(c, &w, 2));
    for (int g = 0; g < nfg; ++g) {
      const Trainer::FilmGroup& fg = t.film[g];
      const int nb = (int)fg.blocks.size();
      ldE[g] = nb * fg.C2;
      embG[g] = take((size_t)Bp * ldE[g]); dembG[g] = take((size_t)Bp * ldE[g]);
      GemmArgs ga{gm, m.P.f(), embG[g], m.P.f(), nullptr, c.segs(), c.batches() + fg.f_b0, Bp, fg.C2, CP, CP, fg.C2, ldE[g]};
      LDP_TRY(run_gemm(w, G_NN, ga, nb, CP / BK, (long long)Bp * ldE[g]));
      LDP_TRY(mark(w, &film_ready[g]));
    }
    for (int i = 0; i < nblk; ++i) {
      const int g = t.film_of[i].first, slot = t.film_of[i].second;
      sv[i].emb = embG[g] + (size_t)slot * t.film[g].C2;
      sv[i].demb = dembG[g] + (size_t)slot * t.film[g].C2;
    }
  }
  auto block_fwd = [&](int i, const float* x) -> int {
    const BlockDesc& b = bs[i];
    const std::string p = "ConditionalResidualBlock1D_" + std::to_string(i), k = "b" + std::to_string(i);
    const size_t ny = (size_t)Bp * b.T * b.cout;
    BlockSave& S = sv[i];
    S.x = x;
    S.c0 = take(ny); S.f = take(ny); S.c1 = take(ny); S.out = take(ny);
    S.st0 = take((size_t)Bp * NG * 2); S.st1 = take((size_t)Bp * NG * 2);
    const float* res = x;
    if (b.proj && c.h->opt.train_group_proj) {              // the block's two convolutions over x as one launch (plan_proj)
      S.res = take(ny);
      const ProjPlan& pp = t.projs[i];
      const int cin_p = rup(b.cin, RP);
      GemmArgs ga{x, m.P.f(), S.c0, m.P.f(), nullptr, c.segs(), c.batches() + pp.f_b0, Bp, b.cout, cin_p, b.T * cin_p, b.cout, b.T * b.cout};
      ga.C2 = S.res;
      LDP_TRY(run_gemm(c, G_NN, ga, pp.f_nb, pp.f_minseg * (cin_p / BK), 0));
      res = S.res;
    } else {
      if (b.proj) {
        S.res = take(ny);
        LDP_TRY(conv_fwd(c, t.convs[k + "r"], x, P(p + "/Conv_0/kernel"), P(p + "/Conv_0/bias"), S.res, Bp));
        res = S.res;
      }
      LDP_TRY(conv_fwd(c, t.convs[k + "c0"], x, P(p + "/Conv1dBlock_0/Conv_0/kernel"), P(p + "/Conv1dBlock_0/Conv_0/bias"), S.c0, Bp));
    }
    const int fgi = t.film_of[i].first;
    if (!film_waited[fgi]) { LDP_TRY(wait_for(c, film_ready[fgi])); film_waited[fgi] = 1; }
    LDP_TRY(gn_fwd(c, S.c0, P(p + "/Conv1dBlock_0/GroupNorm_0/scale"), P(p + "/Conv1dBlock_0/GroupNorm_0/bias"),
       S.emb, (const float*)nullptr, S.f, S.st0, Bp, b.T, b.cout, NG, ldE[fgi]));
    LDP_TRY(conv_fwd(c, t.convs[k + "c1"], S.f, P(p + "/Conv1dBlock_1/Conv_0/kernel"), P(p + "/Conv1dBlock_1/Conv_0/bias"), S.c1, Bp));
    LDP_TRY(gn_fwd(c, S.c1, P(p + "/Conv1dBlock_1/GroupNorm_0/scale"), P(p + "/Conv1dBlock_1/GroupNorm_0/bias"),
       (const float*)nullptr, res, S.out, S.st1, Bp, b.T, b.cout, NG, 0));
    return LDP_OK;
  };

  const float* x = xn;
  std::vector<const float*> skips, down_in(t.L, nullptr), up_in(t.L, nullptr);
  int idx = 0;
  for (int l = 0; l < t.L; ++l) {
    LDP_TRY(block_fwd(idx, x)); x = sv[idx++].out;
    LDP_TRY(block_fwd(idx, x)); x = sv[idx++].out;
    skips.push_back(x);
    if (l + 1 < t.L) {
      const std::string p = "Downsample1d_" + std::to_string(l) + "/Conv_0";
      float* xd = take((size_t)Bp * t.Tl[l + 1] * t.dims[l]);
      down_in[l] = x;
      LDP_TRY(conv_fwd(c, t.convs["down" + std::to_string(l)], x, P(p + "/kernel"), P(p + "/bias"), xd, Bp));
      x = xd;
    }
  }
  LDP_TRY(block_fwd(idx, x)); x = sv[idx++].out;
  LDP_TRY(block_fwd(idx, x)); x = sv[idx++].out;
  for (int i = 0; i + 1 < t.L; ++i) {
    const int lv = t.L - 1 - i, cx = bs[idx].cin / 2, Tl = t.Tl[lv];
    const float* skip = skips.back();
    skips.pop_back();
    float* cat = take((size_t)Bp * Tl * 2 * cx);
    LDP_TRY(copy_cols(c, x, cx, cat, 2 * cx, Bp * Tl, cx));
    LDP_TRY(copy_cols(c, skip, cx, cat + cx, 2 * cx, Bp * Tl, cx));
    LDP_TRY(block_fwd(idx, cat)); x = sv[idx++].out;
    LDP_TRY(block_fwd(idx, x)); x = sv[idx++].out;
    const std::string p = "Upsample1d_" + std::to_string(i) + "/ConvTranspose_0";
    const int cu = t.dims[lv - 1];
    float* xu = take((size_t)Bp * t.Tl[lv - 1] * cu);
    up_in[i] = x;
    LDP_TRY(conv_fwd(c, t.convs["up" + std::to_string(i)], x, P(p + "/kernel"), P(p + "/bias"), xu, Bp));
    x = xu;
  }
  const int c0 = t.dims[0];
  const float* fin_in = x;
  float* cF = take((size_t)Bp * T * c0);
  float* yF = take((size_t)Bp * T * c0);
  float* stF = take((size_t)Bp * 8 * 2);
  LDP_TRY(conv_fwd(c, t.convs["fin"], x, P("Conv1dBlock_0/Conv_0/kernel"), P("Conv1dBlock_0/Conv_0/bias"), cF, Bp));
  LDP_TRY(gn_fwd(c, cF, P("Conv1dBlock_0/GroupNorm_0/scale"), P("Conv1dBlock_0/GroupNorm_0/bias"), (const float*)nullptr,
     (const float*)nullptr, yF, stF, Bp, T, c0, 8, 0));                 // the final Conv1dBlock keeps flax's default of 8 groups (networks/diffusion_nets_v2.py:162-165)
  float* pred = take((size_t)Bp * T * DP);
  LDP_TRY(conv_fwd(c, t.convs["out"], yF, P("Conv_0/kernel"), P("Conv_0/bias"), pred, Bp));

  // ---- loss (agent/ldp_agent.py:124) and its gradient ---------------------------------------------------------------------------------
  float* dpred = take((size_t)Bp * T * DP);
  const long long npred = (long long)Bp * T * DP;
  const int nlb = (int)((npred + 255) / 256);
  float* lpart = take((size_t)nlb);
  const float count = (float)((double)B * T * t.D);
  TK(mse_grad_kernel, dim3(nlb), dim3(256), pred, nz, dpred, lpart, B * T, Bp * T, t.D, DP, alpha * 2.0f / count);
  TK(finish_loss_kernel, dim3(1), dim3(64), lpart, nlb, alpha, count, loss_out);

  // ---- backward ---------------------------------------------------------------------------------------------------------------------
  float* dgm = take((size_t)Bp * CP);
  bool dgm_live = false;
  // weight and data gradient of a width's FiLM layers, on the side context of the LAST block of the width to finish its backward
  std::vector<int> film_left(nfg);
  for (int g = 0; g < nfg; ++g) film_left[g] = (int)t.film[g].blocks.size();
  auto film_bwd = [&](int i) -> int {
    const int g = t.film_of[i].first;
    if (--film_left[g] > 0) return LDP_OK;
    Ctx w;
    LDP_TRY(fork(c, &w, 1, 1));                            // side stream 1 owns dgm (the widths finish in a fixed order, the tail follows them there)
    const Trainer::FilmGroup& fg = t.film[g];
    const int nb = (int)fg.blocks.size();
    GemmArgs gw{gm, dembG[g], m.G.f(), nullptr, nullptr, c.segs(), c.batches() + fg.w_b0, CP, fg.C2, Bp, CP, ldE[g], fg.C2};
    LDP_TRY(run_gemm(w, G_TN, gw, nb, Bp / BK, 0));
    GemmArgs gd{dembG[g], m.P.f(), dgm, nullptr, dgm_live ? dgm : nullptr, c.segs(), c.batches() + fg.d_b0, Bp, CP, fg.C2, ldE[g], fg.C2, CP};
    LDP_TRY(run_gemm(w, G_NT, gd, 1, nb * (fg.C2 / BK), (long long)Bp * CP));
    dgm_live = true;
    return LDP_OK;
  };
  auto block_bwd = [&](int i, const float* dout, bool need_dx, float** dx_out) -> int {
    const BlockDesc& b = bs[i];
    const std::string p = "ConditionalResidualBlock1D_" + std::to_string(i), k = "b" + std::to_string(i);
    const size_t ny = (size_t)Bp * b.T * b.cout;
    const int C = b.cout, cin_p = rup(b.cin, RP);
    BlockSave& S = sv[i];
    float* dc1 = take(ny);
    float* part1 = take((size_t)Bp * 3 * C);              // (per use: the side stream reads it while the main stream runs on)
    LDP_TRY(gn_bwd(c, dout, S.c1, S.st1, P(p + "/Conv1dBlock_1/GroupNorm_0/scale"), P(p + "/Conv1dBlock_1/GroupNorm_0/bias"),
       (const float*)nullptr, dc1, part1, (float*)nullptr, Bp, b.T, C, NG, 0));
    Ctx w;
    LDP_TRY(fork(c, &w));                                  // dout, dc1, part1 exist
    LDP_TRY(gn_param_grads(w, part1, Bp, C, Gd(p + "/Conv1dBlock_1/GroupNorm_0/scale"), Gd(p + "/Conv1dBlock_1/GroupNorm_0/bias"), Gd(p + "/Conv1dBlock_1/Conv_0/bias")));
    LDP_TRY(conv_wgrad(w, t.convs[k + "c1"], S.f, dc1, Gd(p + "/Conv1dBlock_1/Conv_0/kernel"), Bp));
    if (b.proj) {
      LDP_TRY(conv_wgrad(w, t.convs[k + "r"], S.x, dout, Gd(p + "/Conv_0/kernel"), Bp));
      LDP_TRY(colsum(w, dout, C, Bp * b.T, C, Gd(p + "/Conv_0/bias")));
    }
    float* df = take(ny);
    LDP_TRY(conv_dgrad(c, t.convs[k + "c1"], dc1, P(p + "/Conv1dBlock_1/Conv_0/kernel"), nullptr, df, Bp));
    float* dc0 = take(ny);
    float* part0 = take((size_t)Bp * 3 * C);
    LDP_TRY(gn_bwd(c, df, S.c0, S.st0, P(p + "/Conv1dBlock_0/GroupNorm_0/scale"), P(p + "/Conv1dBlock_0/GroupNorm_0/bias"),
       S.emb, dc0, part0, S.demb, Bp, b.T, C, NG, ldE[t.film_of[i].first]));
    LDP_TRY(fork(c, &w));                                  // dc0, part0, S.demb exist
    LDP_TRY(gn_param_grads(w, part0, Bp, C, Gd(p + "/Conv1dBlock_0/GroupNorm_0/scale"), Gd(p + "/Conv1dBlock_0/GroupNorm_0/bias"), Gd(p + "/Conv1dBlock_0/Conv_0/bias")));
    LDP_TRY(conv_wgrad(w, t.convs[k + "c0"], S.x, dc0, Gd(p + "/Conv1dBlock_0/Conv_0/kernel"), Bp));
    // FiLM Dense: emb = gm @ Wf + bf.  Bias gradient here; weight and data gradients once per block width, when its last block is through (film_bwd)
    LDP_TRY(colsum(w, S.demb, ldE[t.film_of[i].first], Bp, 2 * C, Gd(p + "/Dense_0/bias")));
    LDP_TRY(film_bwd(i));
    if (need_dx) {
      float* dx = take((size_t)Bp * b.T * cin_p);
      if (b.proj && c.h->opt.train_group_proj) {            // both data gradients into x as one segmented launch (plan_proj)
        const ProjPlan& pp = t.projs[i];
        GemmArgs gd{dc0, m.P.f(), dx, nullptr, nullptr, c.segs(), c.batches() + pp.d_b0, Bp, cin_p, C, b.T * C, C, b.T * cin_p};
        gd.A2 = dout;
        LDP_TRY(run_gemm(c, G_NT, gd, pp.d_nb, pp.d_minseg * (C / BK), (long long)Bp * b.T * cin_p));
      } else if (b.proj) {
        LDP_TRY(conv_dgrad(c, t.convs[k + "c0"], dc0, P(p + "/Conv1dBlock_0/Conv_0/kernel"), nullptr, dx, Bp));
        LDP_TRY(conv_dgrad(c, t.convs[k + "r"], dout, P(p + "/Conv_0/kernel"), dx, dx, Bp));
      } else {
        LDP_TRY(conv_dgrad(c, t.convs[k + "c0"], dc0, P(p + "/Conv1dBlock_0/Conv_0/kernel"), dout, dx, Bp));
      }
      *dx_out = dx;
    }
    return LDP_OK;
  };

  // output 1x1 conv and the final Conv1dBlock
  Ctx w;
  LDP_TRY(fork(c, &w));
  LDP_TRY(conv_wgrad(w, t.convs["out"], yF, dpred, Gd("Conv_0/kernel"), Bp));
  LDP_TRY(colsum(w, dpred, DP, Bp * T, DP, Gd("Conv_0/bias")));
  float* dyF = take((size_t)Bp * T * c0);
  LDP_TRY(conv_dgrad(c, t.convs["out"], dpred, P("Conv_0/kernel"), nullptr, dyF, Bp));
  float* dcF = take((size_t)Bp * T * c0);
  float* partF = take((size_t)Bp * 3 * c0);
  LDP_TRY(gn_bwd(c, dyF, cF, stF, P("Conv1dBlock_0/GroupNorm_0/scale"), P("Conv1dBlock_0/GroupNorm_0/bias"), (const float*)nullptr,
     dcF, partF, (float*)nullptr, Bp, T, c0, 8, 0));
  LDP_TRY(fork(c, &w));
  LDP_TRY(gn_param_grads(w, partF, Bp, c0, Gd("Conv1dBlock_0/GroupNorm_0/scale"), Gd("Conv1dBlock_0/GroupNorm_0/bias"), Gd("Conv1dBlock_0/Conv_0/bias")));
  LDP_TRY(conv_wgrad(w, t.convs["fin"], fin_in, dcF, Gd("Conv1dBlock_0/Conv_0/kernel"), Bp));
  float* d = take((size_t)Bp * T * c0);
  LDP_TRY(conv_dgrad(c, t.convs["fin"], dcF, P("Conv1dBlock_0/Conv_0/kernel"), nullptr, d, Bp));

  // up path, last stage first
  std::vector<float*> dskip(t.L, nullptr);
  idx = nblk - 1;
  for (int i = t.L - 2; i >= 0; --i) {
    const int lv = t.L - 1 - i, cu = t.dims[lv - 1], Tl = t.Tl[lv];
    const std::string p = "Upsample1d_" + std::to_string(i) + "/ConvTranspose_0";
    const ConvPlan& up = t.convs["up" + std::to_string(i)];
    LDP_TRY(fork(c, &w));
    LDP_TRY(conv_wgrad(w, up, up_in[i], d, Gd(p + "/kernel"), Bp));
    LDP_TRY(colsum(w, d, cu, Bp * up.Tout, cu, Gd(p + "/bias")));
    float* du = take((size_t)Bp * Tl * cu);
    LDP_TRY(conv_dgrad(c, up, d, P(p + "/kernel"), nullptr, du, Bp));
    float* dx = nullptr;
    LDP_TRY(block_bwd(idx--, du, true, &dx));
    float* dcat = nullptr;
    LDP_TRY(block_bwd(idx--, dx, true, &dcat));
    const int cx = bs[idx + 1].cin / 2;
    float* dmain = take((size_t)Bp * Tl * cx);
    dskip[lv] = take((size_t)Bp * Tl * cx);
    LDP_TRY(copy_cols(c, dcat, 2 * cx, dmain, cx, Bp * Tl, cx));
    LDP_TRY(copy_cols(c, dcat + cx, 2 * cx, dskip[lv], cx, Bp * Tl, cx));
    d = dmain;
  }
  // mid blocks
  {
    float* dx = nullptr;
    LDP_TRY(block_bwd(idx--, d, true, &dx));
    LDP_TRY(block_bwd(idx--, dx, true, &d));
  }
  // down path, deepest level first: the level's output also fed its skip connection (when the up path consumed it)
  for (int l = t.L - 1; l >= 0; --l) {
    if (dskip[l]) {
      const long long n = (long long)Bp * t.Tl[l] * t.dims[l];
      float* sum = take((size_t)n);
      TK(add2_kernel, g1(n), dim3(256), d, dskip[l], sum, n);
      d = sum;
    }
    float* dx = nullptr;
    LDP_TRY(block_bwd(idx--, d, true, &dx));
    float* dxin = nullptr;
    LDP_TRY(block_bwd(idx--, dx, l > 0, &dxin));
    if (l > 0) {
      const std::string p = "Downsample1d_" + std::to_string(l - 1) + "/Conv_0";
      const ConvPlan& dn = t.convs["down" + std::to_string(l - 1)];
      LDP_TRY(fork(c, &w));
      LDP_TRY(conv_wgrad(w, dn, down_in[l - 1], dxin, Gd(p + "/kernel"), Bp));
      LDP_TRY(colsum(w, dxin, dn.cout, Bp * dn.Tout, dn.cout, Gd(p + "/bias")));
      float* dd = take((size_t)Bp * dn.Tin * dn.cin);
      LDP_TRY(conv_dgrad(c, dn, dxin, P(p + "/kernel"), nullptr, dd, Bp));
      d = dd;
    }
  }
  // the conditioning vector: g = [temb | cond], gm = mish(g); only the time-embedding half has parameters behind it
  // (the whole branch hangs off dgm, which the side stream accumulated: it stays there, behind the last block's FiLM data gradient)
  float* dg = take((size_t)Bp * CP);
  LDP_TRY(fork(c, &w, 1, 1));
  LDP_TRY(act_bwd(w, dgm, CP, gbuf, CP, dg, CP, Bp, CP, 1));
  LDP_TRY(dense_wgrad(w, md0, 4 * E, dg, CP, Gd("Dense_1/kernel"), E, Bp, 4 * E, E));
  LDP_TRY(colsum(w, dg, CP, Bp, E, Gd("Dense_1/bias")));
  float* dmd0 = take((size_t)Bp * 4 * E);
  float* dd0 = take((size_t)Bp * 4 * E);
  LDP_TRY(dense_dgrad(w, dg, CP, P("Dense_1/kernel"), E, nullptr, dmd0, 4 * E, Bp, 4 * E, E));
  LDP_TRY(act_bwd(w, dmd0, 4 * E, d0, 4 * E, dd0, 4 * E, Bp, 4 * E, 1));
  LDP_TRY(dense_wgrad(w, semb, E, dd0, 4 * E, Gd("Dense_0/kernel"), 4 * E, Bp, E, 4 * E));
  LDP_TRY(colsum(w, dd0, 4 * E, Bp, 4 * E, Gd("Dense_0/bias")));
  LDP_TRY(flush_colsums(w));                               // (w: forked after the main stream's last launch, on the side stream that ran the tail)
  if (!c.dry) LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// ---- IDM: loss + gradients (agent/ldp_agent.py:129-140; networks/mlp_diffusion_nets.py:8-68, networks/mlp_nets.py:49-97) -----------------
int idm_tape(Ctx& c, const float* s_in, const float* a0, const float* noise, const int* tdev, float alpha, float* loss_out, int R) {
  Trainer& t = *c.t;
  Module& m = t.idm;
  const int Rp = rup(R, RP), H = t.IH, A = t.A, AP = t.AP, INP = t.INP, TD = t.TD, S2 = 2 * t.D, CO = A + S2;
  c.L->ws_used = 0;
  auto P = [&](const std::string& path) { return m.P.f() + m.leaf(path).off; };
  auto Gd = [&](const std::string& path) { return m.G.f() + m.leaf(path).off; };
  auto take = [&](size_t n) { return ws_take(*c.L, n); };
  // ---- forward ------------------------------------------------------------------------------------------------------------------
  float* noisy = take((size_t)Rp * AP);
  float* nz = take((size_t)Rp * AP);
  TK(add_noise_pad_kernel, g1((long long)Rp * AP), dim3(256), a0, noise, tdev, abar_of(c.h->cfg.idm_train_steps), noisy, nz, AP, R, Rp, A, AP, 1);
  float* inb = take((size_t)Rp * INP);                                  // [a | s | cond | 0]: networks/mlp_diffusion_nets.py:66 concat order
  LDP_TRY(copy_cols(c, nullptr, 0, inb, INP, Rp, INP));
  LDP_TRY(copy_cols(c, noisy, AP, inb, INP, Rp, A));
  LDP_TRY(copy_cols(c, s_in, S2, inb + A, INP, R, S2));
  float* semb = take((size_t)Rp * TD);
  TK(gather_rows_kernel, g1((long long)Rp * TD), dim3(256), t.sintab_i.f(), tdev, semb, TD, R, Rp, TD);
  float* c1 = take((size_t)Rp * H);
  float* mc1 = take((size_t)Rp * H);
  LDP_TRY(dense_fwd(c, semb, TD, P("MLP_0/Dense_0/kernel"), H, P("MLP_0/Dense_0/bias"), nullptr, c1, H, Rp, TD, H));
  LDP_TRY(act_fwd(c, c1, H, mc1, H, Rp, H, 1));
  LDP_TRY(dense_fwd(c, mc1, H, P("MLP_0/Dense_1/kernel"), H, P("MLP_0/Dense_1/bias"), nullptr, inb + CO, INP, Rp, H, H));
  float* hcur = take((size_t)Rp * H);
  LDP_TRY(dense_fwd(c, inb, INP, P("MLPResNet_0/Dense_0/kernel"), H, P("MLPResNet_0/Dense_0/bias"), nullptr, hcur, H, Rp, INP, H));
  struct BS { float *h, *y, *st, *u0, *u; };
  std::vector<BS> sv(t.NB);
  for (int b = 0; b < t.NB; ++b) {
    const std::string p = "MLPResNet_0/MLPResNetBlock_" + std::to_string(b);
    BS& S = sv[b];
    S.h = hcur;
    S.y = take((size_t)Rp * H); S.st = take((size_t)Rp * 2); S.u0 = take((size_t)Rp * 4 * H); S.u = take((size_t)Rp * 4 * H);
    TK(ln_fwd_kernel, dim3((Rp + 3) / 4), dim3(256), S.h, P(p + "/LayerNorm_0/scale"), P(p + "/LayerNorm_0/bias"), S.y, S.st, Rp, H);
    LDP_TRY(dense_fwd(c, S.y, H, P(p + "/Dense_0/kernel"), 4 * H, P(p + "/Dense_0/bias"), nullptr, S.u0, 4 * H, Rp, H, 4 * H));
    LDP_TRY(act_fwd(c, S.u0, 4 * H, S.u, 4 * H, Rp, 4 * H, 2));
    float* hn = take((size_t)Rp * H);
    LDP_TRY(dense_fwd(c, S.u, 4 * H, P(p + "/Dense_1/kernel"), H, P(p + "/Dense_1/bias"), S.h, hn, H, Rp, 4 * H, H));
    hcur = hn;
  }
  float* hr = take((size_t)Rp * H);
  LDP_TRY(act_fwd(c, hcur, H, hr, H, Rp, H, 2));
  float* pred = take((size_t)Rp * AP);
  LDP_TRY(dense_fwd(c, hr, H, P("MLPResNet_0/Dense_1/kernel"), AP, P("MLPResNet_0/Dense_1/bias"), nullptr, pred, AP, Rp, H, AP));
  // ---- loss ---------------------------------------------------------------------------------------------------------------------
  float* dpred = take((size_t)Rp * AP);
  const long long npred = (long long)Rp * AP;
  const int nlb = (int)((npred + 255) / 256);
  float* lpart = take((size_t)nlb);
  const float count = (float)((double)R * A);
  TK(mse_grad_kernel, dim3(nlb), dim3(256), pred, nz, dpred, lpart, R, Rp, A, AP, alpha * 2.0f / count);
  TK(finish_loss_kernel, dim3(1), dim3(64), lpart, nlb, alpha, count, loss_out);
  // ---- backward -----------------------------------------------------------------------------------------------------------------
  Ctx w;
  LDP_TRY(fork(c, &w));
  LDP_TRY(dense_wgrad(w, hr, H, dpred, AP, Gd("MLPResNet_0/Dense_1/kernel"), AP, Rp, H, AP));
  LDP_TRY(colsum(w, dpred, AP, Rp, AP, Gd("MLPResNet_0/Dense_1/bias")));
  float* dhr = take((size_t)Rp * H);
  LDP_TRY(dense_dgrad(c, dpred, AP, P("MLPResNet_0/Dense_1/kernel"), AP, nullptr, dhr, H, Rp, H, AP));
  float* dh = take((size_t)Rp * H);
  LDP_TRY(act_bwd(c, dhr, H, hcur, H, dh, H, Rp, H, 2));
  for (int b = t.NB - 1; b >= 0; --b) {
    const std::string p = "MLPResNet_0/MLPResNetBlock_" + std::to_string(b);
    BS& S = sv[b];
    float* part = take((size_t)Rp * 2 * H);
    LDP_TRY(fork(c, &w));                                                // dh (and the previous block's LayerNorm partial sums) exist
    LDP_TRY(dense_wgrad(w, S.u, 4 * H, dh, H, Gd(p + "/Dense_1/kernel"), H, Rp, 4 * H, H));
    LDP_TRY(colsum(w, dh, H, Rp, H, Gd(p + "/Dense_1/bias")));
    float* du = take((size_t)Rp * 4 * H);
    float* du0 = take((size_t)Rp * 4 * H);
    LDP_TRY(dense_dgrad(c, dh, H, P(p + "/Dense_1/kernel"), H, nullptr, du, 4 * H, Rp, 4 * H, H));
    LDP_TRY(act_bwd(c, du, 4 * H, S.u0, 4 * H, du0, 4 * H, Rp, 4 * H, 2));
    LDP_TRY(fork(c, &w));
    LDP_TRY(dense_wgrad(w, S.y, H, du0, 4 * H, Gd(p + "/Dense_0/kernel"), 4 * H, Rp, H, 4 * H));
    LDP_TRY(colsum(w, du0, 4 * H, Rp, 4 * H, Gd(p + "/Dense_0/bias")));
    float* dy = take((size_t)Rp * H);
    LDP_TRY(dense_dgrad(c, du0, 4 * H, P(p + "/Dense_0/kernel"), 4 * H, nullptr, dy, H, Rp, H, 4 * H));
    float* dhp = take((size_t)Rp * H);
    TK(ln_bwd_kernel, dim3((Rp + 3) / 4), dim3(256), dy, S.h, S.st, P(p + "/LayerNorm_0/scale"), dh, dhp, part, Rp, H);
    LDP_TRY(fork(c, &w));
    LDP_TRY(colsum_to(w, part, 2 * H, Rp, 2 * H, ColOut{{Gd(p + "/LayerNorm_0/scale"), Gd(p + "/LayerNorm_0/bias"), nullptr}, H}));
    dh = dhp;
  }
  LDP_TRY(fork(c, &w));
  LDP_TRY(dense_wgrad(w, inb, INP, dh, H, Gd("MLPResNet_0/Dense_0/kernel"), H, Rp, INP, H));
  LDP_TRY(colsum(w, dh, H, Rp, H, Gd("MLPResNet_0/Dense_0/bias")));
  float* dcv = take((size_t)Rp * H);                                    // gradient w.r.t. the cond-encoder output: rows A + 2D .. of the input Dense
  LDP_TRY(dense_dgrad(c, dh, H, P("MLPResNet_0/Dense_0/kernel") + (size_t)CO * H, H, nullptr, dcv, H, Rp, H, H));
  LDP_TRY(fork(c, &w));
  LDP_TRY(dense_wgrad(w, mc1, H, dcv, H, Gd("MLP_0/Dense_1/kernel"), H, Rp, H, H));
  LDP_TRY(colsum(w, dcv, H, Rp, H, Gd("MLP_0/Dense_1/bias")));
  float* dmc1 = take((size_t)Rp * H);
  float* dc1 = take((size_t)Rp * H);
  LDP_TRY(dense_dgrad(c, dcv, H, P("MLP_0/Dense_1/kernel"), H, nullptr, dmc1, H, Rp, H, H));
  LDP_TRY(act_bwd(c, dmc1, H, c1, H, dc1, H, Rp, H, 1));
  LDP_TRY(fork(c, &w, 1, 1));
  LDP_TRY(dense_wgrad(w, semb, TD, dc1, H, Gd("MLP_0/Dense_0/kernel"), H, Rp, TD, H));
  LDP_TRY(colsum(w, dc1, H, Rp, H, Gd("MLP_0/Dense_0/bias")));
  LDP_TRY(flush_colsums(w));
  if (!c.dry) LDP_HIP(hipGetLastError());
  return LDP_OK;
}

// size the workspace with a dry walk of the tape, then enqueue it
template <class F>
int run_tape(ldp_handle* h, int lane, hipStream_t s, F&& tape) {
  Lane& t = trainer(h)->lane[lane];
  Ctx c{h, trainer(h), &t, s, true};
  t.colsum_need = 0;
  t.part_need = 0;
  t.coljobs.clear();
  LDP_TRY(tape(c));
  if (t.ws_used > t.ws_floats || t.colsum_need > t.colsum_tmp.bytes || t.part_need > t.gemm_part.bytes) {
    LDP_HIP(hipStreamSynchronize(s));                          // (an earlier step may still be reading the old workspace; its side stream was joined into s)
    if (t.ws_used > t.ws_floats) {
      LDP_TRY(t.ws.alloc(t.ws_used * 4));
      t.ws_floats = t.ws_used;
    }
    LDP_TRY(t.colsum_tmp.alloc(t.colsum_need));
    LDP_TRY(t.gemm_part.alloc(t.part_need));
    for (int k = 0; k < Lane::NS; ++k) {
      LDP_TRY(t.colsum_tmp2[k].alloc(t.colsum_need));
      LDP_TRY(t.gemm_part2[k].alloc(t.part_need));
    }
  }
  c.dry = false;
  LDP_TRY(tape(c));
  return join(c);
}

int need_module(ldp_handle* h, int32_t module, Module** out) {
  if (!h) return fail(LDP_EINVAL, "null handle");
  if (!h->train) return fail(LDP_ESTATE, "ldp_train_init was not called");
  Module* m = module_of(h, module, nullptr);
  if (!m) return fail(LDP_EINVAL, "module must be 1 (planner) or 2 (idm), got %d", module);
  if (!m->ready) return fail(LDP_ESTATE, "ldp_train_init was not called for module %d", module);
  *out = m;
  return LDP_OK;
}

}  // namespace

void train_destroy(ldp_handle* h) {
  delete trainer(h);
  h->train = nullptr;
}

}  // namespace ldp

using namespace ldp;

extern "C" {

int ldp_train_init(ldp_handle* h, int32_t modules, void* stream) {
  if (!h) return fail(LDP_EINVAL, "null handle");
  if (!(modules & 3) || (modules & ~3)) return fail(LDP_EINVAL, "modules must be a mask of 1 (planner) and 2 (idm)");
  LDP_HIP(hipSetDevice(h->cfg.device));
  LDP_TRY(ensure_trainer(h));
  LDP_HIP(hipDeviceSynchronize());                                      // (the module's lane may have work on other streams than `stream`)
  for (int bit = 1; bit <= 2; bit <<= 1) {
    if (!(modules & bit)) continue;
    {
      Lane& t = trainer(h)->lane[bit - 1];
      LDP_TRY(t.gemm_cnt.alloc(CNT_TILES * 4));
      LDP_HIP(hipMemset(t.gemm_cnt.p, 0, CNT_TILES * 4));
      for (int k = 0; k < Lane::NS; ++k) {
        LDP_TRY(t.gemm_cnt2[k].alloc(CNT_TILES * 4));
        LDP_HIP(hipMemset(t.gemm_cnt2[k].p, 0, CNT_TILES * 4));
        if (!t.s2[k]) LDP_HIP(hipStreamCreateWithFlags(&t.s2[k], hipStreamNonBlocking));
      }
      while (t.events.size() < 256) {
        hipEvent_t e;
        LDP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        t.events.push_back(e);
      }
    }
    const char* prefix = nullptr;
    Module* m = module_of(h, bit, &prefix);
    std::vector<float> img(m->total, 0.0f);
    for (const Leaf& l : m->leaves) {
      auto it = h->weights.find(std::string(prefix) + l.path);
      if (it == h->weights.end()) return fail(LDP_ESTATE, "weight '%s%s' was never set", prefix, l.path.c_str());
      if (it->second.numel() != (int64_t)l.taps * l.rows * l.cols) {      // (numel, not the shape vector: idm_finalize re-labels Dense kernels as one-tap convolutions)
        std::string got, want;
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        for (auto v : l.shape) want += std::to_string(v) + ",";
        return fail(LDP_EINVAL, "weight '%s%s' has shape (%s), the module expects (%s)", prefix, l.path.c_str(), got.c_str(), want.c_str());
      }
      pack_leaf(l, it->second.data.data(), img.data() + l.off);
    }
    LDP_TRY(m->P.alloc(m->total * 4));
    LDP_TRY(m->G.alloc(m->total * 4));
    LDP_TRY(m->M.alloc(m->total * 4));
    LDP_TRY(m->V.alloc(m->total * 4));
    LDP_TRY(m->gpart.alloc(((m->total + 1023) / 1024) * 4));
    LDP_HIP(hipMemcpy(m->P.p, img.data(), m->total * 4, hipMemcpyHostToDevice));
    LDP_HIP(hipMemset(m->G.p, 0, m->total * 4));
    LDP_HIP(hipMemset(m->M.p, 0, m->total * 4));
    LDP_HIP(hipMemset(m->V.p, 0, m->total * 4));
    m->step = 0;
    m->gpart_fresh = false;
    m->ready = true;
  }
  return LDP_OK;
}

int ldp_train_planner_grad(ldp_handle* h, const float* x0, const float* noise, const int32_t* t_dev, const float* cond, float alpha,
                           float* loss_out, int32_t B, void* stream) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, 1, &m));
  if (!x0 || !noise || !t_dev || !loss_out || B <= 0 || (h->cfg.global_cond_dim > 0 && !cond)) return fail(LDP_EINVAL, "bad argument");
  LDP_HIP(hipSetDevice(h->cfg.device));
  m->gpart_fresh = false;
  return run_tape(h, 0, (hipStream_t)stream, [&](Ctx& c) { return planner_tape(c, x0, noise, t_dev, cond, alpha, loss_out, B); });
}

int ldp_train_idm_grad(ldp_handle* h, const float* s, const float* a0, const float* noise, const int32_t* t_dev, float alpha, float* loss_out,
                       int32_t R, void* stream) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, 2, &m));
  if (!s || !a0 || !noise || !t_dev || !loss_out || R <= 0) return fail(LDP_EINVAL, "bad argument");
  LDP_HIP(hipSetDevice(h->cfg.device));
  m->gpart_fresh = false;
  return run_tape(h, 1, (hipStream_t)stream, [&](Ctx& c) { return idm_tape(c, s, a0, noise, t_dev, alpha, loss_out, R); });
}

int ldp_train_grad_norm(ldp_handle* h, int32_t modules, float* out, void* stream) {
  if (!h || !out) return fail(LDP_EINVAL, "bad argument");
  if (!(modules & 3) || (modules & ~3)) return fail(LDP_EINVAL, "modules must be a mask of 1 (planner) and 2 (idm)");
  hipStream_t s = (hipStream_t)stream;
  const float* pa[2] = {nullptr, nullptr};
  long long na[2] = {0, 0};
  int k = 0;
  for (int bit = 1; bit <= 2; bit <<= 1) {
    if (!(modules & bit)) continue;
    Module* m = nullptr;
    LDP_TRY(need_module(h, bit, &m));
    const long long nb = (long long)((m->total + 1023) / 1024);
    if (!m->gpart_fresh) hipLaunchKernelGGL(sumsq1_kernel, dim3((unsigned)nb), dim3(256), 0, s, m->G.f(), (long long)m->total, m->gpart.f());
    pa[k] = m->gpart.f();
    na[k++] = nb;
  }
  hipLaunchKernelGGL(sumsq2_kernel, dim3(1), dim3(1024), 0, s, pa[0], na[0], pa[1], na[1], out);
  LDP_HIP(hipGetLastError());
  return LDP_OK;
}

int ldp_train_apply(ldp_handle* h, int32_t module, float lr, float b1, float b2, float eps, void* stream) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, module, &m));
  const long long count = m->step + 1;
  const float bc1 = (float)(1.0 - std::pow((double)b1, (double)count)), bc2 = (float)(1.0 - std::pow((double)b2, (double)count));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((m->total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, m->P.f(), m->G.f(), m->M.f(), m->V.f(),
                     (long long)m->total, lr, b1, b2, eps, bc1, bc2, m->gpart.f());
  LDP_HIP(hipGetLastError());
  m->gpart_fresh = true;
  m->step = count;
  return LDP_OK;
}

int ldp_train_step_count(ldp_handle* h, int32_t module, int64_t set_to, int64_t* out) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, module, &m));
  if (set_to >= 0) m->step = set_to;
  if (out) *out = m->step;
  return LDP_OK;
}

static int leaf_io(ldp_handle* h, int32_t module, int32_t which, const char* path, float* host, int64_t numel, bool write, void* stream) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, module, &m));
  if (!path || !host) return fail(LDP_EINVAL, "bad argument");
  if (which < 0 || which > 3) return fail(LDP_EINVAL, "which must be 0 (params), 1 (grads), 2 (mu) or 3 (nu)");
  auto it = m->index.find(path);
  if (it == m->index.end()) return fail(LDP_EKEY, "module %d has no leaf '%s'", module, path);
  const Leaf& l = m->leaves[it->second];
  if (numel != (int64_t)l.taps * l.rows * l.cols) return fail(LDP_EINVAL, "leaf '%s' has %lld elements, caller passed %lld", path, (long long)l.taps * l.rows * l.cols, (long long)numel);
  DevBuf& buf = which == 0 ? m->P : which == 1 ? m->G : which == 2 ? m->M : m->V;
  LDP_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::vector<float> img(l.size_p(), 0.0f);
  if (write) {
    if (which == 1) m->gpart_fresh = false;
    pack_leaf(l, host, img.data());
    LDP_HIP(hipMemcpy(buf.f() + l.off, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  } else {
    LDP_HIP(hipMemcpy(img.data(), buf.f() + l.off, img.size() * 4, hipMemcpyDeviceToHost));
    unpack_leaf(l, img.data(), host);
  }
  return LDP_OK;
}

int ldp_train_read(ldp_handle* h, int32_t module, int32_t which, const char* path, float* host_out, int64_t numel, void* stream) {
  return leaf_io(h, module, which, path, host_out, numel, false, stream);
}

int ldp_train_write(ldp_handle* h, int32_t module, int32_t which, const char* path, const float* host_in, int64_t numel, void* stream) {
  return leaf_io(h, module, which, path, const_cast<float*>(host_in), numel, true, stream);
}

int ldp_train_arena(ldp_handle* h, int32_t module, int32_t which, float** dev_out, int64_t* numel_out) {
  Module* m = nullptr;
  LDP_TRY(need_module(h, module, &m));
  if (!dev_out || !numel_out) return fail(LDP_EINVAL, "bad argument");
  if (which < 0 || which > 3) return fail(LDP_EINVAL, "which must be 0 (params), 1 (grads), 2 (mu) or 3 (nu)");
  DevBuf& buf = which == 0 ? m->P : which == 1 ? m->G : which == 2 ? m->M : m->V;
  if (which == 1) m->gpart_fresh = false;                    // (the caller may write the gradients: the all-reduce of a data-parallel step)
  *dev_out = buf.f();
  *numel_out = (int64_t)m->total;
  return LDP_OK;
}

int ldp_train_publish(ldp_handle* h, int32_t modules, void* stream) {
  if (!h) return fail(LDP_EINVAL, "null handle");
  if (!(modules & 3) || (modules & ~3)) return fail(LDP_EINVAL, "modules must be a mask of 1 (planner) and 2 (idm)");
  hipStream_t s = (hipStream_t)stream;
  LDP_HIP(hipSetDevice(h->cfg.device));
  LDP_HIP(hipStreamSynchronize(s));
  for (int bit = 1; bit <= 2; bit <<= 1) {
    if (!(modules & bit)) continue;
    Module* m = nullptr;
    LDP_TRY(need_module(h, bit, &m));
    const char* prefix = bit == 1 ? "planner/" : "idm/";
    std::vector<float> img(m->total);
    LDP_HIP(hipMemcpy(img.data(), m->P.p, m->total * 4, hipMemcpyDeviceToHost));
    for (const Leaf& l : m->leaves) {
      HostTensor t;
      t.shape = l.shape;
      t.data.resize((size_t)l.taps * l.rows * l.cols);
      unpack_leaf(l, img.data() + l.off, t.data.data());
      h->weights[std::string(prefix) + l.path] = std::move(t);
    }
  }
  drop_graphs(h);
  if (modules & 1) LDP_TRY(planner_finalize(h, s));
  if (modules & 2) LDP_TRY(idm_finalize(h, s));
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}

}  // extern "C"

