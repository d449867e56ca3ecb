// sconv.hip -- 3x3 convolution (stride 1, pad 1, NHWC) of fp32 tensors on v_mfma_f32_32x32x16_bf16 through
// three-plane split operands (sconv.hpp).  The StableVAE's ResnetBlock2D convolutions at 64 / 32 / 16 pixels
// (diffusers FlaxAutoencoderKL, model/stable_vae_model.yaml:4-16): 93 % of an encode's FLOPs.
//
// Work-group = 256 output pixels (whole image rows) x 128 output channels, 8 waves = 4 (64 pixels) x 2 (64 columns),
// each wave 2 x 2 MFMA tiles of 32 x 32, six plane products per tile and k-step.  K loop = (16-channel chunk, image
// row offset dh); per iteration the three dw taps x 2 x 2 tiles x 6 products = 72 MFMAs per wave.
//   * Input planes arrive as 16-byte units (8 channels of one pixel of one plane).  The work-group's halo tile
//     ((R + 2) x (W + 2) pixels) of a chunk is staged ONCE for all nine taps: LDS image [plane][k half][halo pixel],
//     so the A fragment of tap (dh, dw) is the same ds_read_b128 at a shifted address (conflict-free: 16 lanes of a
//     read group always cover 16 consecutive 16-byte slots).  Zero padding = units fetched from a zero page.
//   * Weights are packed at finalize as the LDS image of each (chunk, dh) iteration: [dw][plane][k half][128 columns].
//   * Both operands move global -> LDS by DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), double buffered,
//     one barrier per iteration; 150 KB of LDS, one work-group (two waves per SIMD) per CU.
//   * Accumulation: the hh products in one accumulator, the five small products (mm, lh, hl, mh, hm: small to large)
//     in a second one, added at the end -- the small terms are not rounded at the big accumulator's ulp
//     (tools/split_bf16_probe.hip: 0.16 fp32 ulps rms at K = 5120 against 0.48 for the fp32 MFMA chain).
//   * Epilogue: + bias (+ residual), fp32 NHWC store, per-tile column sums for the GroupNorm that follows.
#include "sconv.hpp"

#include <cstring>
#include <type_traits>

#include "common.hpp"

#pragma clang fp contract(off)

#ifndef LDP_RANGE_GUARD
#define LDP_RANGE_GUARD 1
#endif

namespace ldp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// NPL = 3: bf16 planes (h, m, l), six products; NPL = 2: fp16 planes (h, l' = (x - h) * 2^11), three products -- h h into accB, h l' + l' h into
// accS, added with weight 2^-11 in the epilogue (tconv.hpp TConvCfg::F16)
template <int W, int NPL = 3>
struct SCfg {
  static constexpr int R = 256 / W;                        // image rows per work-group tile
  static constexpr int HALO_W = W + 2, HALO_R = R + 2;
  static constexpr int NPIX = HALO_R * HALO_W;
  static constexpr int NPIXP = (NPIX + 31) / 32 * 32;      // 6 * NPIXP is then a whole number of 64-lane DMA instructions
  static constexpr int A_UNITS = 2 * NPL * NPIXP;          // [plane][k half][halo pixel]
  static constexpr int A_INSTR = A_UNITS / 64;
  static constexpr int A_PER_WAVE = (A_INSTR + 7) / 8;
  static constexpr int B_UNITS = 3 * NPL * 2 * 128;        // [dw][plane][k half][column]
  static constexpr int B_INSTR = B_UNITS / 64;             // 36 (24)
  static constexpr int B_PER_WAVE = (B_INSTR + 7) / 8;
  static constexpr int LDS_UNITS = 2 * A_UNITS + 2 * B_UNITS;
  static constexpr int LDS_BYTES = LDS_UNITS * 16;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(256 % W == 0 && (W == 64 || W == 32 || W == 16), "tile shapes");
};

struct SConvK {
  const u32x4* xp;
  const u32x4* wp;
  const float* bias;
  const float* res_in;
  float* out;
  float* stats_part;
  const u32x4* zero;
  int N, H, cin, cout;
  unsigned int plane_units;
  int dbg;               // -DLDP_ABLATE builds only (tools/), bits 16.. (the low bits are tconv's): 0x10000 no DMA inside the loop, 0x20000 no MFMAs,
                         // 0x40000 no epilogue, 0x80000 no fragment reads
};
#ifdef LDP_ABLATE
#define LDP_DBG(bit) ((a.dbg >> 16) & (bit))
#else
#define LDP_DBG(bit) 0
#endif

__device__ __forceinline__ void dma16(const u32x4* src, u32x4* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

#define LDP_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0)
#define LDP_MH(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0)

template <int W, bool DUAL, bool PIPE, int NPL = 3>
__global__ __launch_bounds__(512, 2) void sconv3_kernel(const SConvK a) {
  using C = SCfg<W, NPL>;
  static_assert(NPL == 3 || (NPL == 2 && DUAL), "fp16 planes: the low products have their own accumulator");
  constexpr int R = C::R, HALO_W = C::HALO_W, NPIXP = C::NPIXP;
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // block -> (pixel tile, column tile): the column tiles of a pixel tile run back to back on one XCD (block id % 8,
  // observed dispatch; speed only), so the halo tile crosses the fabric once
  const int nct = a.cout >> 7;
  const int tpi = a.H / R;                                 // tiles per image
  const int ntiles = a.N * tpi;
  int tile, ct;
  {
    const int id = blockIdx.x;
    if ((ntiles & 7) == 0) { const int j = id >> 3; tile = (j / nct) * 8 + (id & 7); ct = j % nct; }
    else { tile = id / nct; ct = id % nct; }
  }
  const int n = tile / tpi, tr = tile % tpi, h0 = tr * R;
  const int C8 = a.cin >> 3;
  const int nchunk = a.cin >> 4, nit = nchunk * 3;
  const unsigned int chunk_step = 2u * a.H * W;            // units between 16-channel chunks

  // ---- DMA plan of the halo tile (fixed across chunks) ----
  unsigned int a_idx[C::A_PER_WAVE];
  bool a_ok[C::A_PER_WAVE];
#pragma unroll
  for (int i = 0; i < C::A_PER_WAVE; ++i) {
    const int s = (wave + 8 * i) * 64 + lane;
    const int pc = s / NPIXP, hp = s % NPIXP;
    const int plane = pc >> 1, kh = pc & 1;
    const int hr = hp / HALO_W, hc = hp % HALO_W;
    const int hh = h0 + hr - 1, ww = hc - 1;
    a_ok[i] = hp < C::NPIX && hh >= 0 && hh < a.H && ww >= 0 && ww < W;
    a_idx[i] = (unsigned int)plane * a.plane_units + (unsigned int)(((n * C8 + kh) * a.H + hh) * W + ww);
  }
  auto issue_a = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < C::A_PER_WAVE; ++i) {
      const int j = wave + 8 * i;
      if (i < C::A_INSTR / 8 || wave < C::A_INSTR % 8) {     // compile-time true except for the last round
        const u32x4* src = a_ok[i] ? a.xp + (a_idx[i] + (unsigned int)chunk * chunk_step) : a.zero;
        dma16(src, lds + buf * C::A_UNITS + j * 64);
      }
    }
  };
  const u32x4* wbase = a.wp + (size_t)ct * nit * C::B_UNITS + lane;
  auto issue_b = [&](int it, int buf) {
#pragma unroll
    for (int i = 0; i < C::B_PER_WAVE; ++i) {
      const int j = wave + 8 * i;
      if (i < C::B_INSTR / 8 || wave < C::B_INSTR % 8)
        dma16(wbase + (size_t)it * C::B_UNITS + j * 64, lds + 2 * C::A_UNITS + buf * C::B_UNITS + j * 64);
    }
  };

  // ---- fragment addresses (bytes from the start of LDS) ----
  const int kh_lane = lane >> 5, l31 = lane & 31;
  int a_off[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = wm * 64 + mt * 32 + l31;                 // pixel of the tile (row-major over its R rows)
    a_off[mt] = (kh_lane * NPIXP + (q / W) * HALO_W + (q % W)) * 16;
  }
  const int b_off = (2 * C::A_UNITS + kh_lane * 128 + wn * 64 + l31) * 16;
  const char* ldsb = reinterpret_cast<const char*>(lds);

  f32x16 accB[2][2], accS[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accB[mt][nt][r] = 0.f; accS[mt][nt][r] = 0.f; }

  issue_a(0, 0);
  issue_b(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if constexpr (!PIPE) {
  for (int chunk = 0; chunk < nchunk; ++chunk) {
      const int abuf = chunk & 1;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const int it = chunk * 3 + dh;
        const int bbuf = it & 1;
        if (!LDP_DBG(1)) {
          if (it + 1 < nit) issue_b(it + 1, bbuf ^ 1);
          if (dh == 0 && chunk + 1 < nchunk) issue_a(chunk + 1, abuf ^ 1);
        }
        const char* ab = ldsb + abuf * (C::A_UNITS * 16);
        const char* bb = ldsb + b_off + bbuf * (C::B_UNITS * 16);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          bf16x8 fa[2][3], fb[2][3];
          static_assert(NPL == 3 || PIPE, "fp16 planes: pipelined loop only");
          if (LDP_DBG(8)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) { fa[i][pl] = __builtin_bit_cast(bf16x8, lds[0]); fb[i][pl] = fa[i][pl]; }
            asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]));
            asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]));
          } else {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              fa[mt][pl] = *reinterpret_cast<const bf16x8*>(ab + a_off[mt] + (pl * 2 * NPIXP + dh * HALO_W + dw) * 16);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              fb[nt][pl] = *reinterpret_cast<const bf16x8*>(bb + ((dw * 3 + pl) * 256 + nt * 32) * 16);
          }
          if (LDP_DBG(2)) {
            asm volatile("" ::"v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]));
            asm volatile("" ::"v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[0][2]), "v"(fb[1][0]), "v"(fb[1][1]), "v"(fb[1][2]));
          } else
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              f32x16& cs = DUAL ? accS[mt][nt] : accB[mt][nt];
              LDP_MF(fa[mt][1], fb[nt][1], cs);               // m m
              LDP_MF(fa[mt][2], fb[nt][0], cs);               // l h
              LDP_MF(fa[mt][0], fb[nt][2], cs);               // h l
              LDP_MF(fa[mt][1], fb[nt][0], cs);               // m h
              LDP_MF(fa[mt][0], fb[nt][1], cs);               // h m
              LDP_MF(fa[mt][0], fb[nt][0], accB[mt][nt]);     // h h
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  } else {
    // Software pipeline over the 9 (dh, dw) steps of a chunk: the fragments of step s + 1 are requested before the
    // MFMAs of step s (two register sets, alternating), and the iteration's wait-for-DMA + barrier sits in front of its
    // LAST step: what the step prefetches there are the first fragments of the next iteration, so no wave starts an
    // iteration with an empty matrix pipe.  Every read of an iteration's weight buffer is issued before that barrier
    // (its dw = 2 fragments were prefetched during dw = 1), the DMA that overwrites the buffer after it.
    bf16x8 fa[2][2][NPL], fb[2][2][NPL];                    // [set][tile][plane] (fp16 planes travel as the same 16-byte units)
    auto read_frags = [&](auto set_, const char* ab, const char* bb, auto dh_, auto dw_) {
      constexpr int set = decltype(set_)::value, dh = decltype(dh_)::value, dw = decltype(dw_)::value;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fa[set][mt][pl] = *reinterpret_cast<const bf16x8*>(ab + a_off[mt] + (pl * 2 * NPIXP + dh * HALO_W + dw) * 16);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          fb[set][nt][pl] = *reinterpret_cast<const bf16x8*>(bb + ((dw * NPL + pl) * 256 + nt * 32) * 16);
    };
    auto mfmas = [&](auto set_) {
      constexpr int set = decltype(set_)::value;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if constexpr (NPL == 2) {
            LDP_MH(fa[set][mt][1], fb[set][nt][0], accS[mt][nt]);   // l' h
            LDP_MH(fa[set][mt][0], fb[set][nt][0], accB[mt][nt]);   // h h
            LDP_MH(fa[set][mt][0], fb[set][nt][1], accS[mt][nt]);   // h l'
          } else {
          f32x16& cs = DUAL ? accS[mt][nt] : accB[mt][nt];
          LDP_MF(fa[set][mt][1], fb[set][nt][1], cs);       // m m
          LDP_MF(fa[set][mt][NPL - 1], fb[set][nt][0], cs); // l h
          LDP_MF(fa[set][mt][0], fb[set][nt][NPL - 1], cs); // h l
          LDP_MF(fa[set][mt][1], fb[set][nt][0], cs);       // m h
          LDP_MF(fa[set][mt][0], fb[set][nt][1], cs);       // h m
          LDP_MF(fa[set][mt][0], fb[set][nt][0], accB[mt][nt]);   // h h
          }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto abase = [&](int buf) { return ldsb + buf * (C::A_UNITS * 16); };
    auto bbase = [&](int buf) { return ldsb + b_off + buf * (C::B_UNITS * 16); };
    // one (dh, dw) step of chunk `chunk`; PAR = parity of the register set holding this step's fragments
    auto step = [&](auto par_, auto dh_, auto dw_, int chunk) {
      constexpr int PAR = decltype(par_)::value, dh = decltype(dh_)::value, dw = decltype(dw_)::value;
      using CUR = std::integral_constant<int, PAR>;
      using NXT = std::integral_constant<int, PAR ^ 1>;
      const int it = chunk * 3 + dh;
      if (dw == 0 && !LDP_DBG(1)) {
        if (it + 1 < nit) issue_b(it + 1, (it + 1) & 1);
        if (dh == 0 && chunk + 1 < nchunk) issue_a(chunk + 1, (chunk + 1) & 1);
      }
      if (dw == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (!LDP_DBG(8)) {
        if (dw < 2) read_frags(NXT{}, abase(chunk & 1), bbase(it & 1), dh_, std::integral_constant<int, (dw + 1) % 3>{});
        else if (dh < 2) read_frags(NXT{}, abase(chunk & 1), bbase((it + 1) & 1), std::integral_constant<int, (dh + 1) % 3>{}, I0{});
        else if (chunk + 1 < nchunk) read_frags(NXT{}, abase((chunk + 1) & 1), bbase((it + 1) & 1), I0{}, I0{});
      }
      if (!LDP_DBG(2)) mfmas(CUR{});
      // pin the interleave: the 12 fragment reads of the NEXT step behind the first 12 MFMAs of this one, one each (left
      // alone the scheduler sinks the reads to their first use, i.e. prefetch distance zero); they have the other 12
      // MFMAs to land in
      constexpr int NRD = 4 * NPL, NMF = 4 * (NPL == 2 ? 3 : 6);      // 12 reads / 24 matrix instructions (fp16 planes: 8 / 12)
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
    };
    auto chunk_body = [&](auto par_, int chunk) {          // 9 steps: the set parity flips from one chunk to the next
      constexpr int P = decltype(par_)::value;
      using A = std::integral_constant<int, P>;
      using B = std::integral_constant<int, P ^ 1>;
      step(A{}, I0{}, I0{}, chunk); step(B{}, I0{}, I1{}, chunk); step(A{}, I0{}, I2{}, chunk);
      step(B{}, I1{}, I0{}, chunk); step(A{}, I1{}, I1{}, chunk); step(B{}, I1{}, I2{}, chunk);
      step(A{}, I2{}, I0{}, chunk); step(B{}, I2{}, I1{}, chunk); step(A{}, I2{}, I2{}, chunk);
    };
    read_frags(I0{}, abase(0), bbase(0), I0{}, I0{});
    int chunk = 0;
    for (; chunk + 1 < nchunk; chunk += 2) { chunk_body(I0{}, chunk); chunk_body(I1{}, chunk + 1); }
    if (chunk < nchunk) chunk_body(I0{}, chunk);
  }

  // ---- epilogue: + bias (+ residual), fp32 NHWC store, column sums ----
  // element (pixel q, column) of this tile = base[lane part + uniform part]: one 32-bit per-lane offset, the rest scalar
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  const size_t tile_off = (((size_t)n * a.H + h0) * W) * a.cout + ct * 128;
  float* const obase = a.out + tile_off;
  const float* const rbase = a.res_in ? a.res_in + tile_off : nullptr;
  const unsigned int lane_off = (unsigned int)((wm * 64 + 4 * kh_lane) * a.cout + wn * 64 + l31);
  auto finish = [&](auto has_res) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float bias = a.bias[ct * 128 + wn * 64 + nt * 32 + l31];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned int e = lane_off + (unsigned int)((mt * 32 + (r & 3) + 8 * (r >> 2)) * a.cout + nt * 32);
          float v = (NPL == 2 ? accB[mt][nt][r] + accS[mt][nt][r] * (1.0f / 2048.0f) : DUAL ? accS[mt][nt][r] + accB[mt][nt][r] : accB[mt][nt][r]) + bias;
          if (decltype(has_res)::value) v += rbase[e];
          obase[e] = v;
          s1[nt] += v;
          s2[nt] += v * v;
        }
      }
    }
  };
  if (LDP_DBG(4)) return;
  if (a.res_in) finish(std::true_type{}); else finish(std::false_type{});
  if (a.stats_part) {
    float* st = reinterpret_cast<float*>(lds);              // [wm 4][128 columns][2]; every wave is past its last LDS read
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float t1 = s1[nt] + __shfl_xor(s1[nt], 32);
      const float t2 = s2[nt] + __shfl_xor(s2[nt], 32);
      if (kh_lane == 0) {
        st[(wm * 128 + wn * 64 + nt * 32 + l31) * 2] = t1;
        st[(wm * 128 + wn * 64 + nt * 32 + l31) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < 256) {
      const int c = tid >> 1, k = tid & 1;
      const float t = ((st[c * 2 + k] + st[(128 + c) * 2 + k]) + st[(256 + c) * 2 + k]) + st[(384 + c) * 2 + k];
      a.stats_part[((size_t)tile * a.cout + ct * 128 + c) * 2 + k] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// producer: fp32 NHWC -> planes, optionally through GroupNorm (+ swish).  Block = 32 pixels x all channels:
// coalesced float4 reads, the 16-byte units transposed through LDS so that every (plane, 8-channel block) leaves
// as one 512-byte run.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float swish_p(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh);
  m = __builtin_bit_cast(unsigned short, bm);
  l = __builtin_bit_cast(unsigned short, bl);
}

constexpr int PL_PIX = 32, PL_PAD = 33;

__device__ __forceinline__ void split2h(float x, unsigned short& h, unsigned short& l) {
  const _Float16 bh = (_Float16)x;
  const _Float16 bl = (_Float16)((x - (float)bh) * 2048.0f);
  h = __builtin_bit_cast(unsigned short, bh);
  l = __builtin_bit_cast(unsigned short, bl);
}

template <bool GN, int NPL>
__global__ __launch_bounds__(256) void planes_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                                     u32x4* __restrict__ planes, int HW, int C, int G, int act,
                                                     unsigned int plane_units, unsigned int* __restrict__ range) {
  extern __shared__ u32x4 sh[];                            // [3][C/8][PL_PAD] units
  const int tid = threadIdx.x, C8 = C >> 3, cq = C >> 2;
  const size_t pix0 = (size_t)blockIdx.x * PL_PIX;
  const int n = (int)(pix0 / HW), p_in = (int)(pix0 % HW);
  unsigned long long* sh8 = reinterpret_cast<unsigned long long*>(sh);
  for (int i = tid; i < PL_PIX * cq; i += 256) {
    const int p = i / cq, q = i % cq, c = q * 4;
    float4 v = *reinterpret_cast<const float4*>(x + (pix0 + p) * C + c);
    if (GN) {
      const float* st = stats + ((size_t)n * G + c / (C / G)) * 2;
      const float mean = st[0], rstd = st[1];
      const float4 s = *reinterpret_cast<const float4*>(scale + c);
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      v.x = (v.x - mean) * rstd * s.x + b.x; v.y = (v.y - mean) * rstd * s.y + b.y;
      v.z = (v.z - mean) * rstd * s.z + b.z; v.w = (v.w - mean) * rstd * s.w + b.w;
      if (act) { v.x = swish_p(v.x); v.y = swish_p(v.y); v.z = swish_p(v.z); v.w = swish_p(v.w); }
    }
    unsigned short h[4], m[4], l[4];
    if constexpr (NPL == 2) {
      // range guard of the fp16 planes: an element with |x| >= 65504 (or not finite) would be +-inf in its h plane and NaN in the conv.  The
      // producer sees every element once: raise the handle's range word (the host recomputes on three bf16 planes, engine.hpp range_fallback)
      constexpr float LIM = 65504.0f;
      if (LDP_RANGE_GUARD && range && (!(fabsf(v.x) < LIM) || !(fabsf(v.y) < LIM) || !(fabsf(v.z) < LIM) || !(fabsf(v.w) < LIM))) *range = 1u;
      split2h(v.x, h[0], m[0]); split2h(v.y, h[1], m[1]); split2h(v.z, h[2], m[2]); split2h(v.w, h[3], m[3]);
    } else {
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]);
    split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    }
    const int c8 = q >> 1, half = q & 1;
    auto pack = [](const unsigned short* t) {
      return (unsigned long long)t[0] | ((unsigned long long)t[1] << 16) | ((unsigned long long)t[2] << 32) | ((unsigned long long)t[3] << 48);
    };
    sh8[((0 * C8 + c8) * PL_PAD + p) * 2 + half] = pack(h);
    sh8[((1 * C8 + c8) * PL_PAD + p) * 2 + half] = pack(m);
    if constexpr (NPL == 3) sh8[((2 * C8 + c8) * PL_PAD + p) * 2 + half] = pack(l);
  }
  __syncthreads();
  for (int u = tid; u < NPL * C8 * PL_PIX; u += 256) {
    const int p = u % PL_PIX, pc = u / PL_PIX;
    const int plane = pc / C8, c8 = pc % C8;
    planes[(size_t)plane * plane_units + ((size_t)n * C8 + c8) * HW + p_in + p] = sh[pc * PL_PAD + p];
  }
}

int planes_launch(const float* x, const float* stats, const float* scale, const float* bias, void* planes,
                  int N, int HW, int C, int G, int act, hipStream_t s, int npl, unsigned int* range) {
  // (C <= 256: npl * (C / 8) * 33 * 16 bytes of dynamic LDS stay under the 64 KB a kernel gets without hipFuncSetAttribute; the StableVAE reaches 256)
  if (HW % PL_PIX != 0 || C % 8 != 0 || C > 256 || (npl != 2 && npl != 3)) return -100;
  const unsigned int pu = (unsigned int)((size_t)N * (C / 8) * HW);
  const size_t ldsb = (size_t)npl * (C / 8) * PL_PAD * 16;
  const unsigned int grid = (unsigned int)((size_t)N * HW / PL_PIX);
  if (stats && npl == 3)
    hipLaunchKernelGGL((planes_kernel<true, 3>), dim3(grid), dim3(256), ldsb, s, x, stats, scale, bias, (u32x4*)planes, HW, C, G, act, pu, range);
  else if (stats)
    hipLaunchKernelGGL((planes_kernel<true, 2>), dim3(grid), dim3(256), ldsb, s, x, stats, scale, bias, (u32x4*)planes, HW, C, G, act, pu, range);
  else if (npl == 3)
    hipLaunchKernelGGL((planes_kernel<false, 3>), dim3(grid), dim3(256), ldsb, s, x, stats, scale, bias, (u32x4*)planes, HW, C, G, act, pu, range);
  else
    hipLaunchKernelGGL((planes_kernel<false, 2>), dim3(grid), dim3(256), ldsb, s, x, stats, scale, bias, (u32x4*)planes, HW, C, G, act, pu, range);
  return (int)hipGetLastError();
}

std::vector<uint16_t> pack_sconv3(const float* k, int cin, int cout, int npl) {
  const int nct = cout / 128, nchunk = cin / 16;
  std::vector<uint16_t> out((size_t)nct * nchunk * 3 * 3 * npl * 2 * 128 * 8);
  auto hbits = [](_Float16 v) { uint16_t u; std::memcpy(&u, &v, 2); return u; };
  size_t o = 0;
  for (int ct = 0; ct < nct; ++ct)
    for (int chunk = 0; chunk < nchunk; ++chunk)
      for (int dh = 0; dh < 3; ++dh)
        for (int dw = 0; dw < 3; ++dw)
          for (int plane = 0; plane < npl; ++plane)
            for (int kh = 0; kh < 2; ++kh)
              for (int col = 0; col < 128; ++col)
                for (int e = 0; e < 8; ++e) {
                  const int c = chunk * 16 + kh * 8 + e, co = ct * 128 + col;
                  const float v = k[(((size_t)dh * 3 + dw) * cin + c) * cout + co];
                  if (npl == 2) {
                    const _Float16 hh = (_Float16)v;
                    out[o++] = plane == 0 ? hbits(hh) : hbits((_Float16)((v - (float)hh) * 2048.0f));
                    continue;
                  }
                  const uint16_t h = f32_to_bf16_rne(v);
                  const float r1 = v - bf16_to_f32(h);
                  const uint16_t m = f32_to_bf16_rne(r1);
                  const float r2 = r1 - bf16_to_f32(m);
                  const uint16_t l = f32_to_bf16_rne(r2);
                  out[o++] = plane == 0 ? h : plane == 1 ? m : l;
                }
  return out;
}

bool sconv3_supported(int H, int W, int cin, int cout) {
  // cin <= 256: the plane producer's LDS transpose (planes_launch)
  return H == W && (W == 64 || W == 32 || W == 16) && cin % 16 == 0 && cin >= 16 && cin <= 256 && cout % 128 == 0;
}

template <int W, bool DUAL, bool PIPE, int NPL = 3>
static int launch_w(const SConvK& k, int grid, hipStream_t s) {
  static bool init = false;
  auto kern = sconv3_kernel<W, DUAL, PIPE, NPL>;
  if (!init) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SCfg<W, NPL>::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    init = true;
  }
  constexpr int lds_bytes = SCfg<W, NPL>::LDS_BYTES;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, s, k);
  return (int)hipGetLastError();
}

// Two forms are built (round 5 trimmed the A/B arms of round 4 -- two accumulators on bf16 planes: 3-5 % slower; un-pipelined fragment reads:
// the first version): fp16 planes / three products / two accumulators (the default), bf16 planes / six products / one accumulator with the
// fragment reads one step ahead (the fallback of the range guard, and option vae_split_f16 = 0)
template <int W>
static int launch_v(const SConvK& k, int grid, hipStream_t s, int npl) {
  if (npl == 2) return launch_w<W, true, true, 2>(k, grid, s);
  return launch_w<W, false, true>(k, grid, s);
}

int sconv3_launch(const SConvArgs& a, hipStream_t s) {
  if (!sconv3_supported(a.H, a.W, a.cin, a.cout)) return -100;
  PlaneGeom g{a.N, a.H, a.W, a.cin};
  if (g.plane_units() * 3 > 0xffffffffull) return -100;
  SConvK k{(const u32x4*)a.xp, (const u32x4*)a.wp, a.bias, a.res_in, a.out, a.stats_part, (const u32x4*)a.zero,
           a.N, a.H, a.cin, a.cout, (unsigned int)g.plane_units(), a.dbg};
  const int grid = a.N * (a.H * a.W / 256) * (a.cout / 128);
  switch (a.W) {
    case 64: return launch_v<64>(k, grid, s, a.npl);
    case 32: return launch_v<32>(k, grid, s, a.npl);
    case 16: return launch_v<16>(k, grid, s, a.npl);
  }
  return -100;
}

}  // namespace ldp

// times a handle-less primitive left the fp16 planes for the bf16 ones because an operand was out of their range (tests)
static int64_t g_range_fallbacks = 0;
extern "C" int64_t ldp_range_fallbacks(void) { return g_range_fallbacks; }

// ---- unit-testable primitive: one 3x3 convolution on split operands (fp32 in, fp32 out) ----
// x (N, H, W, Cin) device fp32; kernel (3, 3, Cin, Cout) / bias (Cout) host fp32 (Flax layout); res (N, H, W, Cout)
// device fp32 or NULL; stats_out (N * H * W / 256, Cout, 2) device fp32 or NULL.  Synchronises `stream`.
extern "C" int ldp_conv2d_3x3_bf16x3(const float* x, const float* kernel_host, const float* bias_host, const float* res,
                                     float* y, float* stats_out, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                     int32_t Cout, int32_t dual, void* stream) {
  using namespace ldp;
  if (!x || !kernel_host || !bias_host || !y || N <= 0) return fail(LDP_EINVAL, "bad argument");
  if (!sconv3_supported(H, W, Cin, Cout))
    return fail(LDP_EINVAL, "split-operand 3x3 conv: square 64 / 32 / 16 pixel images, Cin %% 16 == 0 and <= 256, Cout %% 128 == 0");
  hipStream_t s = (hipStream_t)stream;
  // dual = 2: two fp16 planes, three products -- guarded like the engine's convs: weights outside the planes' range (|w| >= 65504) select the
  // bf16 planes up front, an activation outside it is caught by the plane producer and the conv reruns on bf16 planes (fp32 range)
  int npl = dual == 2 ? 2 : 3;
  if (npl == 2 && !fits_f16_planes(kernel_host, (size_t)9 * Cin * Cout)) { npl = 3; ++g_range_fallbacks; }
  DevBuf db_, dp_, dz_, dflag;
  LDP_TRY(upload(db_, bias_host, (size_t)Cout * 4, s));
  PlaneGeom g{N, H, W, Cin};
  LDP_TRY(dp_.alloc(g.bytes()));
  LDP_TRY(dz_.alloc(256));
  LDP_TRY(dflag.alloc(16));
  LDP_HIP(hipMemsetAsync(dz_.p, 0, 256, s));
  LDP_HIP(hipMemsetAsync(dflag.p, 0, 16, s));
  for (;;) {
    std::vector<uint16_t> wp = pack_sconv3(kernel_host, Cin, Cout, npl);
    DevBuf dw_;
    LDP_TRY(upload(dw_, wp.data(), wp.size() * 2, s));
    int r = planes_launch(x, nullptr, nullptr, nullptr, dp_.p, N, H * W, Cin, 1, 0, s, npl, dflag.as<unsigned int>());
    if (r != 0) return fail(LDP_EHIP, "planes launch failed (%d)", r);
    SConvArgs a{dp_.p, dw_.p, db_.f(), res, y, stats_out, dz_.p, N, H, W, Cin, Cout};
    a.npl = npl;
    r = sconv3_launch(a, s);
    if (r != 0) return fail(r == -100 ? LDP_EINVAL : LDP_EHIP, "split-operand 3x3 conv launch failed (%d)", r);
    unsigned int flag = 0;
    LDP_HIP(hipMemcpyAsync(&flag, dflag.p, 4, hipMemcpyDeviceToHost, s));
    LDP_HIP(hipStreamSynchronize(s));
    if (npl == 2 && flag != 0u) { npl = 3; ++g_range_fallbacks; continue; }
    break;
  }
  LDP_HIP(hipStreamSynchronize(s));
  return LDP_OK;
}
