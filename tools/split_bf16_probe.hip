// split_bf16_probe.hip -- stage 0(b) of the round-4 question "can the fp32 convolutions run on the bf16 matrix pipe?"
// An fp32 value splits EXACTLY into three bf16 planes x = h + m + l (8 + 8 + 8 significand bits, round-to-nearest
// residues); every bf16 x bf16 product is exact in fp32; 9 products reproduce a*b exactly, 6 (dropping m*l, l*m, l*l)
// leave <= 2^-23 |a*b|.  What the matrix pipe does with the SUM is the open question, so this probe measures
//   A. the sustained issue rate of the 6- and 9-product inner loop (operands in registers; and with the split done
//      by the VALU next to the MFMAs), for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16;
//   B. the error of C = A(32 x K) * B(K x 32) against float64 for K = 256 .. 5120, for the present exact-fp32 path
//      (v_mfma_f32_16x16x4_f32), 3-, 6- and 9-product split forms in several accumulation orders, on N(0,1)-like and
//      on all-positive operands (a truncating accumulator shows as a bias there);
//   C. a handful of one-MFMA cases that tell round-to-nearest from truncation and a fused wide sum from a chain.
//   hipcc -O3 --offload-arch=gfx950 tools/split_bf16_probe.hip -o tools/bin/split_bf16_probe && tools/bin/split_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Planes { bf8 h, m, l; };
__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; const float r1 = x - (float)h; m = (__bf16)r1; const float r2 = r1 - (float)m; l = (__bf16)r2;
}
__device__ inline Planes split8(const float* v) {
  Planes p;
#pragma unroll
  for (int i = 0; i < 8; ++i) { __bf16 h, m, l; split3(v[i], h, m, l); p.h[i] = h; p.m[i] = m; p.l[i] = l; }
  return p;
}

// ---------------------------------------------------------------- A. rate
// NP products per (A tile, B tile) pair; wave tile MT x NT MFMA tiles; SPLIT = the A operand is split from fp32
// registers inside the loop (worst case: once per use).
template <int NP, int MT, int NT, bool SPLIT>
__global__ __launch_bounds__(256) void rate32(float* out, int iters) {
  f16v acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float src[8];
  for (int i = 0; i < 8; ++i) src[i] = 1.0f + 1e-3f * (threadIdx.x + 7 * i);
  Planes a[MT], b[NT];
  for (int i = 0; i < MT; ++i) a[i] = split8(src);
  for (int j = 0; j < NT; ++j) b[j] = split8(src);
  for (int it = 0; it < iters; ++it) {
    if (SPLIT) {
#pragma unroll
      for (int i = 0; i < MT; ++i) { for (int q = 0; q < 8; ++q) src[q] += acc[0][0][q] * 1e-30f; a[i] = split8(src); }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f16v c = acc[i][j];
        if (NP >= 9) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].m, c, 0, 0, 0); }
        if (NP >= 6) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].m, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].h, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].l, c, 0, 0, 0); }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].h, c, 0, 0, 0);
        acc[i][j] = c;
      }
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}
// the same loop on RANDOM operand bits (every lane / element / plane a different bf16 in +-[0.5, 2)): the sustained clock
// depends on how many multiplier inputs toggle (MI355X_MICROARCH.md "DVFS give-back"), so this is the ceiling a real
// kernel can be priced against
__device__ inline unsigned int mix(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int NP, int MT, int NT>
__global__ __launch_bounds__(256) void rate32_random(float* out, int iters) {
  f16v acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  Planes a[MT], b[NT];
  unsigned int seed = threadIdx.x * 977u + blockIdx.x * 131071u;
  auto fill = [&](bf8& v) {
    for (int e = 0; e < 8; ++e) {
      seed = mix(seed + 0x9e3779b9u);
      const unsigned short bits = (unsigned short)((seed & 0x80ffu) | 0x3f00u | ((seed >> 20) & 0x80u));   // sign, exponent 126..127, random mantissa
      v[e] = __builtin_bit_cast(__bf16, bits);
    }
  };
  for (int i = 0; i < MT; ++i) { fill(a[i].h); fill(a[i].m); fill(a[i].l); }
  for (int j = 0; j < NT; ++j) { fill(b[j].h); fill(b[j].m); fill(b[j].l); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f16v c = acc[i][j];
        if (NP >= 9) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].m, c, 0, 0, 0); }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].l, b[j].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].m, b[j].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].h, b[j].h, c, 0, 0, 0);
        // keep the accumulators bounded without leaving the matrix pipe: nothing (values random-walk, |c| < 1e6 over 20000 iterations)
        acc[i][j] = c;
      }
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}
template <int NP, int MT, int NT>
__global__ __launch_bounds__(256) void rate16(float* out, int iters) {
  f4 acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  float src[8];
  for (int i = 0; i < 8; ++i) src[i] = 1.0f + 1e-3f * (threadIdx.x + 7 * i);
  Planes a[MT], b[NT];
  for (int i = 0; i < MT; ++i) a[i] = split8(src);
  for (int j = 0; j < NT; ++j) b[j] = split8(src);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f4 c = acc[i][j];
        if (NP >= 9) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].l, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b[j].l, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].l, b[j].m, c, 0, 0, 0); }
        if (NP >= 6) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b[j].m, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].l, b[j].h, c, 0, 0, 0);
                       c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].l, c, 0, 0, 0); }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b[j].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b[j].h, c, 0, 0, 0);
        acc[i][j] = c;
      }
  }
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[0] = s;
}

// ---------------------------------------------------------------- B. error of a K-long contraction
// One wave computes C(32 x 32) = A(32 x K) B(K x 32), A row-major (32, K), B row-major (K, 32).
// form 0: v_mfma_f32_16x16x4_f32 (the product library's arithmetic)     form 1: 3 products hh, hm, mh
// form 2: 6 products, one accumulator, small terms first                 form 3: 6 products, hh in its own accumulator
// form 4: 9 products, one accumulator, small terms first                 form 5: 6 products, big terms first
// form 6: 6 products, three accumulators by magnitude class (hh | hm mh | mm hl lh), summed small to large at the end
__global__ __launch_bounds__(64) void dot_forms(const float* A, const float* B, float* C, int K, int form) {
  const int l = threadIdx.x;
  if (form == 0) {
    // four 16x16 tiles; lane l: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15]; C row = (l >> 4) * 4 + r
    for (int ti = 0; ti < 2; ++ti) for (int tj = 0; tj < 2; ++tj) {
      f4 c = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[(ti * 16 + (l & 15)) * K + k0 + (l >> 4)];
        const float b = B[(k0 + (l >> 4)) * 32 + tj * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
      }
      for (int r = 0; r < 4; ++r) C[(ti * 16 + (l >> 4) * 4 + r) * 32 + tj * 16 + (l & 15)] = c[r];
    }
    return;
  }
  f16v c0, c1, c2;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = A[(l & 31) * K + k0 + (l >> 5) * 8 + i]; bv[i] = B[(k0 + (l >> 5) * 8 + i) * 32 + (l & 31)]; }
    const Planes a = split8(av), b = split8(bv);
#define MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0)
    if (form == 1) { MF(a.m, b.h, c0); MF(a.h, b.m, c0); MF(a.h, b.h, c0); }
    if (form == 2) { MF(a.m, b.m, c0); MF(a.l, b.h, c0); MF(a.h, b.l, c0); MF(a.m, b.h, c0); MF(a.h, b.m, c0); MF(a.h, b.h, c0); }
    if (form == 3) { MF(a.m, b.m, c1); MF(a.l, b.h, c1); MF(a.h, b.l, c1); MF(a.m, b.h, c1); MF(a.h, b.m, c1); MF(a.h, b.h, c0); }
    if (form == 4) { MF(a.l, b.l, c0); MF(a.m, b.l, c0); MF(a.l, b.m, c0);
                     MF(a.m, b.m, c0); MF(a.l, b.h, c0); MF(a.h, b.l, c0); MF(a.m, b.h, c0); MF(a.h, b.m, c0); MF(a.h, b.h, c0); }
    if (form == 5) { MF(a.h, b.h, c0); MF(a.h, b.m, c0); MF(a.m, b.h, c0); MF(a.h, b.l, c0); MF(a.l, b.h, c0); MF(a.m, b.m, c0); }
    if (form == 6) { MF(a.m, b.m, c2); MF(a.l, b.h, c2); MF(a.h, b.l, c2); MF(a.m, b.h, c1); MF(a.h, b.m, c1); MF(a.h, b.h, c0); }
#undef MF
  }
  for (int r = 0; r < 16; ++r) {
    const float v = (form == 6) ? (c2[r] + c1[r]) + c0[r] : c1[r] + c0[r];
    C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = v;
  }
}

// ---------------------------------------------------------------- C. rounding behaviour of one MFMA
// Row 0 / column 0 of a 32x32x16 bf16 MFMA: a[k] * b[k] for k < 16 and an accumulator input; everything else zero.
__global__ __launch_bounds__(64) void one_mfma(const float* a16, const float* b16, float cin, float* out) {
  const int l = threadIdx.x;
  bf8 a, b;
  for (int i = 0; i < 8; ++i) {
    const int k = (l >> 5) * 8 + i;
    a[i] = (l & 31) == 0 ? (__bf16)a16[k] : (__bf16)0.f;
    b[i] = (l & 31) == 0 ? (__bf16)b16[k] : (__bf16)0.f;
  }
  f16v c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (l == 0) c[0] = cin;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (l == 0) out[0] = c[0];
}

static double now_ms(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }

template <typename F> static int time_rate(const char* name, F launch, double flop_per_iter_per_wave, int waves_per_wg, int np) {
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wgs : {256, 512}) {
    const int iters = 20000;
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); launch(wgs, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      best = std::fmin(best, now_ms(e0, e1));
    }
    const double raw = flop_per_iter_per_wave * iters * waves_per_wg * wgs / best / 1e9;   // bf16 TFLOP/s issued
    std::printf("rate %-34s %4d WGs x %d waves: %8.3f ms  raw %7.1f TF/s bf16 (%.1f %% of 2500)  fp32-equivalent %6.1f TF/s (x%.2f of 157.3)\n",
                name, wgs, waves_per_wg, best, raw, raw / 25.0, raw / np, raw / np / 157.3);
  }
  CK(hipFree(out));
  return 0;
}

int main() {
  // ---------------- A
  std::printf("== A. issue rate (operands in registers)\n");
#define R32(NP, MT, NT, SP, W) time_rate("32x32x16 NP=" #NP " tile " #MT "x" #NT " split=" #SP, \
    [](int wgs, float* o, int it) { hipLaunchKernelGGL((rate32<NP, MT, NT, SP>), dim3(wgs), dim3(64 * W), 0, 0, o, it); }, \
    2.0 * 32 * 32 * 16 * NP * MT * NT, W, NP)
#define R16(NP, MT, NT, W) time_rate("16x16x32 NP=" #NP " tile " #MT "x" #NT, \
    [](int wgs, float* o, int it) { hipLaunchKernelGGL((rate16<NP, MT, NT>), dim3(wgs), dim3(64 * W), 0, 0, o, it); }, \
    2.0 * 16 * 16 * 32 * NP * MT * NT, W, NP)
  if (R32(6, 2, 2, false, 4)) return 1;
  if (R32(9, 2, 2, false, 4)) return 1;
  if (R32(6, 1, 2, false, 4)) return 1;
  if (R32(6, 1, 1, false, 4)) return 1;
  if (R32(6, 2, 2, true, 4)) return 1;
  if (R32(3, 2, 2, false, 4)) return 1;
#define R32R(NP, MT, NT, W) time_rate("32x32x16 NP=" #NP " tile " #MT "x" #NT " RANDOM bits", \
    [](int wgs, float* o, int it) { hipLaunchKernelGGL((rate32_random<NP, MT, NT>), dim3(wgs), dim3(64 * W), 0, 0, o, it); }, \
    2.0 * 32 * 32 * 16 * NP * MT * NT, W, NP)
  if (R32R(6, 2, 2, 4)) return 1;
  if (R32R(9, 2, 2, 4)) return 1;
  if (R16(6, 2, 2, 4)) return 1;
  if (R16(9, 2, 2, 4)) return 1;
  if (R16(6, 4, 2, 4)) return 1;
  if (R16(6, 1, 1, 4)) return 1;

  // ---------------- B
  std::printf("== B. error of C(32x32) = A(32xK) B(Kx32) against float64; unit = 2^-24 * sum_k |a_k b_k| (an fp32 ulp of the absolute sum)\n");
  const char* names[7] = {"fp32 mfma 16x16x4 (product path)", "3 products (hh hm mh)", "6 products, small first, 1 acc",
                          "6 products, hh own acc", "9 products, small first, 1 acc", "6 products, big first, 1 acc", "6 products, 3 accs by class"};
  std::mt19937_64 rng(1234);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  for (int dist = 0; dist < 3; ++dist) for (int K : {256, 1280, 5120, 10240}) {
    std::vector<float> A(32 * (size_t)K), B((size_t)K * 32);
    for (auto& v : A) v = dist == 1 ? (float)ud(rng) : (float)nd(rng);
    for (auto& v : B) v = dist == 1 ? (float)(ud(rng) / K) : (float)(nd(rng) / std::sqrt((double)K));
    if (dist == 2) for (auto& v : A) v = (float)std::fabs(v) * 0.3f + (float)(nd(rng) > 1.5 ? 3.0 : 0.0);   // Mish-like: positive, a few large
    float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> ref(32 * 32), absum(32 * 32);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0, t = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)A[i * (size_t)K + k] * (double)B[(size_t)k * 32 + j]; s += p; t += std::fabs(p); }
      ref[i * 32 + j] = s; absum[i * 32 + j] = t;
    }
    std::printf("-- operands %s, K = %d (|C| ~ %.2f, sum|ab| ~ %.2f)\n", dist == 0 ? "N(0,1) x N(0,1/K)" : dist == 1 ? "U(0,1) x U(0,1)/K (all positive)" : "Mish-like positive x N(0,1/K)",
                K, std::fabs(ref[5]), absum[5]);
    for (int form = 0; form < 7; ++form) {
      CK(hipMemset(dC, 0, 32 * 32 * 4));
      hipLaunchKernelGGL(dot_forms, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, form);
      std::vector<float> C(32 * 32); CK(hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost));
      double mx = 0, rms = 0, mxabs = 0, bias = 0;
      for (int i = 0; i < 1024; ++i) {
        const double e = (double)C[i] - ref[i], u = e / (absum[i] * std::ldexp(1.0, -24));
        mx = std::fmax(mx, std::fabs(u)); rms += u * u; bias += u; mxabs = std::fmax(mxabs, std::fabs(e));
      }
      std::printf("   %-36s max %8.2f  rms %8.3f  mean %+8.3f units   max|err| %.3e\n", names[form], mx, std::sqrt(rms / 1024), bias / 1024, mxabs);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }

  // ---------------- C
  std::printf("== C. one v_mfma_f32_32x32x16_bf16: c_in + sum_k a_k b_k, products exact in fp32\n");
  struct Case { const char* what; float cin; int n; float a[16], b[16]; };
  const float e7 = 1.0f + 0.0078125f;   // 1 + 2^-7, exact in bf16
  const float p12 = std::ldexp(1.0f, -12), p13 = std::ldexp(1.0f, -13);
  std::vector<Case> cases = {
    {"1 + 2^-24 (tie)                      RN-even 1, RZ 1, RU 1+ulp", 1.0f, 1, {p12}, {p12}},
    {"1 + 2^-24(1+2^-7)                    RN 1+ulp, RZ 1", 1.0f, 1, {p12 * e7}, {p12}},
    {"1 - 2^-26                            RN 1, RZ 1-ulp/2", 1.0f, 1, {-p13}, {p13}},
    {"1 + 3 x 2^-25 in one MFMA            wide sum 1+ulp, chain of RN adds 1", 1.0f, 3, {p12, p12, p12}, {p13, p13, p13}},
    {"0 + (1 + 3 x 2^-25 as 4 products)    wide sum 1+ulp, chain 1", 0.0f, 4, {1.0f, p12, p12, p12}, {1.0f, p13, p13, p13}},
    {"0 + (3 x 2^-25 first, then 1)        order independence", 0.0f, 4, {p12, p12, p12, 1.0f}, {p13, p13, p13, 1.0f}},
    {"1 + 16 x 2^-27                       wide sum 1+ulp (=1+2^-23), chain 1", 1.0f, 16, {p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13, p13},
       {p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f, p13 * 0.5f}},
    {"2^24 + 1 - 2^24 (k = 0, 1, 2)        exact wide sum 1, fp32 chain 0", 0.0f, 3, {4096.f, 1.0f, -4096.f}, {4096.f, 1.0f, 4096.f}},
    {"c_in 2^24, products 1 and -2^24... c + p: (2^24 + 1) - 2^24", 16777216.0f, 2, {1.0f, -4096.f}, {1.0f, 4096.f}},
    {"2^40 + 1 - 2^40                      how wide is the internal sum?", 0.0f, 3, {1048576.f, 1.0f, -1048576.f}, {1048576.f, 1.0f, 1048576.f}},
    {"2^60 + 1 - 2^60", 0.0f, 3, {1073741824.f, 1.0f, -1073741824.f}, {1073741824.f, 1.0f, 1073741824.f}},
  };
  float *da, *db, *dout; CK(hipMalloc(&da, 64)); CK(hipMalloc(&db, 64)); CK(hipMalloc(&dout, 4));
  for (auto& c : cases) {
    float a[16] = {0}, b[16] = {0};
    for (int i = 0; i < c.n; ++i) { a[i] = c.a[i]; b[i] = c.b[i]; }
    CK(hipMemcpy(da, a, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, c.cin, dout);
    float r; CK(hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost));
    uint32_t bits; std::memcpy(&bits, &r, 4);
    std::printf("   %-80s -> %.9g (0x%08x)  (r - 1) / 2^-23 = %+.3f\n", c.what, r, bits, ((double)r - 1.0) * 8388608.0);
  }
  return 0;
}
