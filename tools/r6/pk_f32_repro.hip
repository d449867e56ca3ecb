// pk_f32_repro.hip -- VERDICT r5 item 2 / ADVICE r5: stand-alone reproduction of the packed-fp32 LayerNorm miscompute of
// csrc/idm.hip (idm_block_h16_kernel, round 5): rows normalised wrongly in the LOW element of a register pair on lanes 48..63, only
// when two of those work-groups shared a CU -- i.e. when one work-group's LayerNorm prologue could run next to the other's MFMA phase.
//
// One work-group = 16 waves = 4 per SIMD (the failing occupancy).  Waves 0..7 are VICTIMS: each computes the LayerNorm affine of a 256-wide row
// (4 columns per lane) twice from the same registers -- once with the v_pk_*_f32 instruction forms of the failing build (issued verbatim
// through inline asm), once element by element behind empty asm statements (what the library ships) -- and counts bit differences per
// (lane, element).  Waves 8..15 are AGGRESSORS running one of:
//   0 nothing                       3 ds_read_b128 loop
//   1 v_mfma_f32_16x16x32_f16 loop  4 VALU fma loop
//   2 v_mfma_f32_16x16x4_f32 loop   5 MFMA f16 fed from LDS (the real kernel's phase)
// Victim variants: 0 = statistics through the DPP wave_sum + v_readlane (the real code) and packed form A; 1 = form A, statistics passed in as
// kernel arguments (no readlane, no SGPR-pair packed multiply); 2 = like 0 with `s_nop 4` between the statistics and the packed affine;
// 3 = form B (the subtraction packed too, op_sel picking the mean out of the statistics pair); 4 = form C (no op_sel anywhere).
//   hipcc -O3 --offload-arch=gfx950 tools/r6/pk_f32_repro.hip -o tools/bin/pk_f32_repro && tools/bin/pk_f32_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma clang fp contract(off)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {      // csrc/tconv.hpp wave_sum, verbatim
  v = dpp_add<0xB1, 0xF>(v);
  v = dpp_add<0x4E, 0xF>(v);
  v = dpp_add<0x114, 0xF>(v);
  v = dpp_add<0x118, 0xF>(v);
  v = dpp_add<0x142, 0xA>(v);
  v = dpp_add<0x143, 0xC>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float unit(unsigned h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }      // [-1, 1)

template <int AGG, int VIC>
__global__ __launch_bounds__(1024) void probe(unsigned* counts, float* samples, int iters, float arg_mean, float arg_rstd, float* sink, unsigned victim_mask) {
  extern __shared__ f32x4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every wave allocates 128 VGPRs (the library kernel's 124 round up to that): four waves fill a SIMD's register file, so the third and fourth
  // wave of a SIMD -- waves 8..15 here, the SECOND work-group of a CU in the library -- live in the upper half of it
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  for (int i = tid; i < 8192; i += 1024) lds4[i] = f32x4{0.001f * (i & 255), 0.5f, -0.25f, 1.0f};
  __syncthreads();
  if (!((victim_mask >> wave) & 1u)) {
    // ------------------------------------------------------------------ aggressors
    if (AGG == 0) return;
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 av = lds4[lane], bv = lds4[64 + lane];
    float fa = 0.5f + 0.001f * lane, fb = 1.0f;
    for (int it = 0; it < iters * 4; ++it) {
      if (AGG == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc[u], 0, 0, 0);
      } else if (AGG == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc[u], 0, 0, 0);
      } else if (AGG == 3) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = acc[u & 3] + lds4[((it * 8 + u) * 64 + lane) & 8191];
      } else if (AGG == 4) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { fb = fmaf(fa, fb, 0.25f); fa = fmaf(fb, 0.5f, fa * 0.25f); }
      } else if (AGG == 5) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 a2 = lds4[((it * 4 + u) * 64 + lane) & 8191], b2 = lds4[((it * 4 + u) * 64 + 4096 + lane) & 8191];
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, b2), acc[u], 0, 0, 0);
        }
      }
    }
    float s = fa + fb;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    if (s == 123.456f) sink[tid] = s;
    return;
  }
  // -------------------------------------------------------------------- victims
  const unsigned base = (blockIdx.x * 16u + wave) * 0x9e3779b9u;
  f32x4 ls, lb;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ls[e] = 1.0f + 0.1f * unit(hash32(0xabcd0000u + 4 * lane + e));
    lb[e] = 0.02f * unit(hash32(0x12340000u + 4 * lane + e));
  }
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 3.0f * unit(hash32(base + it * 1024u + 4 * lane + e)) + 0.37f;
    float mean, rstd;
    if (VIC == 1) {
      mean = arg_mean; rstd = arg_rstd;
    } else {
      const float s1 = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
      const float s2 = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
      mean = s1 * (1.0f / 256.0f);
      const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.0f);
      rstd = 1.0f / sqrtf(var + 1e-6f);
      if (VIC == 2) asm volatile("s_nop 4" ::: "memory");
    }
    // (a) the packed forms of the failing build of csrc/idm.hip (round 5; the listing is kept in profiles/r06_pk_f32_isa.txt), issued verbatim:
    //     form A: pk_mul by the (rstd, mean) pair's low element for both halves -> pk_mul ls -> pk_add lb          (VIC 0, 1, 2)
    //     form B: the subtraction packed too: pk_add x, -(statistics pair's high element) for both halves           (VIC 3)
    //     form C: the same chain with NO op_sel anywhere (rstd replicated into a pair first)                        (VIC 4)
    f32x4 y;
    {
      f2 rm = {rstd, mean}, ls01 = {ls[0], ls[1]}, ls23 = {ls[2], ls[3]}, lb01 = {lb[0], lb[1]}, lb23 = {lb[2], lb[3]};
      f2 lo, hi;
      if (VIC == 3) {
        lo = f2{v[0], v[1]}; hi = f2{v[2], v[3]};
        asm volatile("v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_mul_f32 %0, %2, %0 op_sel_hi:[0,1]\n"
                     "v_pk_mul_f32 %0, %3, %0\n"
                     "v_pk_mul_f32 %1, %2, %1 op_sel_hi:[0,1]\n"
                     "v_pk_add_f32 %0, %5, %0\n"
                     "v_pk_mul_f32 %1, %4, %1\n"
                     "v_pk_add_f32 %1, %6, %1\n"
                     : "+v"(lo), "+v"(hi) : "v"(rm), "v"(ls01), "v"(ls23), "v"(lb01), "v"(lb23));
      } else if (VIC >= 5) {
        // ONE packed instruction form at a time: only the subtraction (or, VIC 11, the multiplication by rstd) is packed; the rest of the affine is
        // finished element by element.  Which operand selection is it?
        lo = f2{v[0], v[1]}; hi = f2{v[2], v[3]};
        f2 rmn = {rstd, -mean}, nmr = {-mean, rstd}, nn = {-mean, -mean};
        if (VIC == 5)        // LOW lane reads the HIGH dword of src1 (op_sel:[0,1]), with the neg modifiers of the failing build
          asm volatile("v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                       "v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n" : "+v"(lo), "+v"(hi) : "v"(rm));
        else if (VIC == 6)   // the same 16 wait states after every producer
          asm volatile("s_nop 7\ns_nop 7\n"
                       "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                       "s_nop 7\ns_nop 7\n"
                       "v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n"
                       "s_nop 7\ns_nop 7\n" : "+v"(lo), "+v"(hi) : "v"(rm));
        else if (VIC == 7)   // no neg modifiers (the pair holds -mean)
          asm volatile("v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1]\n"
                       "v_pk_add_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]\n" : "+v"(lo), "+v"(hi) : "v"(rmn));
        else if (VIC == 8)   // the selection on src0 instead of src1
          asm volatile("v_pk_add_f32 %0, %2, %0 op_sel:[1,0] op_sel_hi:[1,1]\n"
                       "v_pk_add_f32 %1, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]\n" : "+v"(lo), "+v"(hi) : "v"(rmn));
        else if (VIC == 9)   // the mirror image: HIGH lane reads the LOW dword (op_sel_hi:[1,0]) -- what form A's multiplications do
          asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0]\n"
                       "v_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0]\n" : "+v"(lo), "+v"(hi) : "v"(nmr));
        else if (VIC == 10)  // control: no selection at all
          asm volatile("v_pk_add_f32 %0, %0, %2\n"
                       "v_pk_add_f32 %1, %1, %2\n" : "+v"(lo), "+v"(hi) : "v"(nn));
        else if (VIC == 11) { // v_pk_mul_f32 with the LOW lane reading the HIGH dword
          lo = f2{v[0] - mean, v[1] - mean}; hi = f2{v[2] - mean, v[3] - mean};
          f2 mr = {mean, rstd};
          asm volatile("v_pk_mul_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1]\n"
                       "v_pk_mul_f32 %1, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]\n" : "+v"(lo), "+v"(hi) : "v"(mr));
        }
        float t4[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = t4[e];
          asm volatile("" : "+v"(t));
          if (VIC != 11) { t = t * rstd; asm volatile("" : "+v"(t)); }
          t = t * ls[e];
          asm volatile("" : "+v"(t));
          t = t + lb[e];
          asm volatile("" : "+v"(t));
          t4[e] = t;
        }
        lo = f2{t4[0], t4[1]}; hi = f2{t4[2], t4[3]};
      } else if (VIC == 4) {
        lo = f2{v[0] - mean, v[1] - mean}; hi = f2{v[2] - mean, v[3] - mean};
        f2 rr = {rstd, rstd};
        asm volatile("v_pk_mul_f32 %0, %2, %0\n"
                     "v_pk_mul_f32 %0, %3, %0\n"
                     "v_pk_mul_f32 %1, %2, %1\n"
                     "v_pk_add_f32 %0, %5, %0\n"
                     "v_pk_mul_f32 %1, %4, %1\n"
                     "v_pk_add_f32 %1, %6, %1\n"
                     : "+v"(lo), "+v"(hi) : "v"(rr), "v"(ls01), "v"(ls23), "v"(lb01), "v"(lb23));
      } else {
        lo = f2{v[0] - mean, v[1] - mean}; hi = f2{v[2] - mean, v[3] - mean};
        asm volatile("v_pk_mul_f32 %0, %2, %0 op_sel_hi:[0,1]\n"
                     "v_pk_mul_f32 %0, %3, %0\n"
                     "v_pk_mul_f32 %1, %2, %1 op_sel_hi:[0,1]\n"
                     "v_pk_add_f32 %0, %5, %0\n"
                     "v_pk_mul_f32 %1, %4, %1\n"
                     "v_pk_add_f32 %1, %6, %1\n"
                     : "+v"(lo), "+v"(hi) : "v"(rm), "v"(ls01), "v"(ls23), "v"(lb01), "v"(lb23));
      }
      y = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
    // (b) element by element, as csrc/idm.hip ships it
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e] - mean;
      asm volatile("" : "+v"(t));
      t = t * rstd;
      asm volatile("" : "+v"(t));
      t = t * ls[e];
      asm volatile("" : "+v"(t));
      r[e] = t + lb[e];
      asm volatile("" : "+v"(r[e]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (__float_as_uint(y[e]) != __float_as_uint(r[e])) {
        atomicAdd(&counts[lane * 4 + e], 1u);
        atomicAdd(&counts[260 + wave], 1u);
        if (nbad < 4) {
          const unsigned slot = atomicAdd(&counts[256], 1u);
          if (slot < 64) {
            float* sp = samples + slot * 8;
            sp[0] = (float)lane; sp[1] = (float)e; sp[2] = v[e]; sp[3] = mean; sp[4] = rstd; sp[5] = y[e]; sp[6] = r[e]; sp[7] = (float)it;
          }
        }
        ++nbad;
      }
  }
}

template <int AGG, int VIC>
static int run(const char* what, int iters, int blocks, unsigned victim_mask = 0x00ffu) {
  unsigned* counts; float *samples, *sink;
  CK(hipMalloc(&counts, 280 * 4)); CK(hipMalloc(&samples, 64 * 8 * 4)); CK(hipMalloc(&sink, 1024 * 4));
  CK(hipMemset(counts, 0, 280 * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<AGG, VIC>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<AGG, VIC>), dim3(blocks), dim3(1024), 8192 * 16, 0, counts, samples, iters, 0.37f, 0.577f, sink, victim_mask);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> c(280); std::vector<float> s(64 * 8);
  CK(hipMemcpy(c.data(), counts, 280 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s.data(), samples, 64 * 8 * 4, hipMemcpyDeviceToHost));
  unsigned long long tot = 0, byrow[4] = {0, 0, 0, 0}, byel[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { tot += c[l * 4 + e]; byrow[l >> 4] += c[l * 4 + e]; byel[e] += c[l * 4 + e]; }
  const double rows = (double)blocks * 8 * iters;
  std::printf("%-58s rows %.3g  bad elements %llu  (lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu; element 0..3: %llu %llu %llu %llu)  %.1f ms\n",
              what, rows, tot, byrow[0], byrow[1], byrow[2], byrow[3], byel[0], byel[1], byel[2], byel[3], ms);
  if (tot) {
    std::printf("    by wave:");
    for (int w = 0; w < 16; ++w) std::printf(" %u", c[260 + w]);
    std::printf("\n");
  }
  const unsigned ns = c[256] < 6 ? c[256] : 6;
  for (unsigned i = 0; i < ns; ++i)
    std::printf("    lane %2d element %d it %5d: v %.9g mean %.9g rstd %.9g  packed %.9g  scalar %.9g\n", (int)s[i * 8], (int)s[i * 8 + 1], (int)s[i * 8 + 7],
                s[i * 8 + 2], s[i * 8 + 3], s[i * 8 + 4], s[i * 8 + 5], s[i * 8 + 6]);
  CK(hipFree(counts)); CK(hipFree(samples)); CK(hipFree(sink));
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 2000, blocks = argc > 2 ? std::atoi(argv[2]) : 512;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  std::printf("# %s, %d CUs; %d work-groups of 16 waves (8 victims + 8 aggressors: 4 waves per SIMD), %d rows per victim wave\n", p.gcnArchName, p.multiProcessorCount, blocks, iters);
  const bool full = argc > 3;
  if (full) {
  if (run<0, 0>("victims alone, form A", iters, blocks)) return 1;
  if (run<1, 0>("form A next to v_mfma_f32_16x16x32_f16 (registers)", iters, blocks)) return 1;
  if (run<5, 0>("form A next to v_mfma_f32_16x16x32_f16 fed from LDS", iters, blocks)) return 1;
  if (run<2, 0>("form A next to v_mfma_f32_16x16x4_f32", iters, blocks)) return 1;
  if (run<3, 0>("form A next to ds_read_b128 loops", iters, blocks)) return 1;
  if (run<4, 0>("form A next to VALU fma loops", iters, blocks)) return 1;
  if (run<1, 1>("form A, MFMA f16 aggressors, statistics from kernel args", iters, blocks)) return 1;
  if (run<1, 2>("form A, MFMA f16 aggressors, s_nop 4 before the affine", iters, blocks)) return 1;
  if (run<1, 4>("form C (no op_sel) next to v_mfma_f32_16x16x32_f16", iters, blocks)) return 1;
  if (run<1, 0>("upper victims, form A, MFMA f16 below", iters, blocks, 0xff00u)) return 1;
  if (run<0, 0>("sixteen victims, form A", iters, blocks, 0xffffu)) return 1;
  }
  std::printf("# form B = the failing build's packed subtract: v_pk_add_f32 d, x, (rstd, mean) op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1] + the form-A chain\n");
  if (run<0, 3>("victims alone, form B", iters, blocks)) return 1;
  if (run<1, 3>("form B next to v_mfma_f32_16x16x32_f16 (registers)", iters, blocks)) return 1;
  if (run<5, 3>("form B next to v_mfma_f32_16x16x32_f16 fed from LDS", iters, blocks)) return 1;
  if (run<2, 3>("form B next to v_mfma_f32_16x16x4_f32", iters, blocks)) return 1;
  if (run<3, 3>("form B next to ds_read_b128 loops", iters, blocks)) return 1;
  if (run<4, 3>("form B next to VALU fma loops", iters, blocks)) return 1;
  if (run<1, 3>("form B, victims in the upper half, MFMA f16 below", iters, blocks, 0xff00u)) return 1;
  if (run<1, 3>("form B, victims = waves 0-3 and 8-11", iters, blocks, 0x0f0fu)) return 1;
  if (run<0, 3>("form B, sixteen victims, no aggressor", iters, blocks, 0xffffu)) return 1;
  std::printf("# one packed instruction at a time, next to v_mfma_f32_16x16x32_f16 (registers)\n");
  if (run<1, 5>("pk_add src1 op_sel:[0,1] + neg (LOW lane <- HIGH dword)", iters, blocks)) return 1;
  if (run<1, 6>("  the same, 16 wait states around every instruction", iters, blocks)) return 1;
  if (run<1, 7>("pk_add src1 op_sel:[0,1], no neg modifiers", iters, blocks)) return 1;
  if (run<1, 8>("pk_add src0 op_sel:[1,0]", iters, blocks)) return 1;
  if (run<1, 9>("pk_add src1 op_sel_hi:[1,0] (HIGH lane <- LOW dword)", iters, blocks)) return 1;
  if (run<1, 10>("pk_add, no operand selection (control)", iters, blocks)) return 1;
  if (run<1, 11>("pk_mul src1 op_sel:[0,1]", iters, blocks)) return 1;
  std::printf("# the same next to v_mfma_f32_16x16x4_f32 / VALU loops / nothing\n");
  if (run<2, 5>("pk_add src1 op_sel:[0,1] + neg, f32 MFMA aggressors", iters, blocks)) return 1;
  if (run<4, 5>("pk_add src1 op_sel:[0,1] + neg, VALU aggressors", iters, blocks)) return 1;
  if (run<0, 5>("pk_add src1 op_sel:[0,1] + neg, no aggressors", iters, blocks)) return 1;
  if (run<0, 5>("pk_add src1 op_sel:[0,1] + neg, sixteen victims", iters, blocks, 0xffffu)) return 1;
  return 0;
}
