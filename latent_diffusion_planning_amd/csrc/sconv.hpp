// sconv.hpp -- fp32 convolutions on the bf16 matrix pipe through three-plane split operands (round 4).
//
// An fp32 value splits EXACTLY into three bf16 planes x = h + m + l (8 + 8 + 8 significand bits, round-to-nearest
// residues).  Every bf16 x bf16 product is exact in fp32, so a * b = sum of 9 plane products exactly; the six kept here
// (hh, hm, mh, hl, lh, mm) leave out m*l + l*m + l*l <= 2^-23 |a b| -- the size of ONE fp32 rounding of the product.
// v_mfma_f32_32x32x16_bf16 accumulates in fp32 (round-to-nearest at the accumulator, measured: tools/split_bf16_probe.hip),
// 16x the rate of v_mfma_f32_16x16x4_f32: six products = 2.65x the fp32 matrix peak.  Measured error of a K = 5120
// contraction against float64: 0.16 (two accumulators: hh | the five small products) .. 0.41 (one accumulator) fp32
// ulps of sum|a b| rms, against 0.48 for the fp32 MFMA chain the rest of this library computes with
// (profiles/r04_split_probe.txt); the 100-step loops emulated on the CPU with these products stay below the plain fp32
// run's error against the float64 goldens (profiles/r04_split_emulation.json).
//
// Late round 4, the default (`npl = 2`): TWO fp16 planes, x ~ h + l' / 2^11 with h = fp16(x), l' = fp16((x - h) * 2^11) -- 22 significand bits, the scale keeps
// l' a normal fp16 whatever |x| --, THREE exact products per multiply-add (v_mfma_f32_32x32x16_f16): h h in one accumulator, h l' + l' h in a second one that the
// epilogue adds with weight 2^-11; the l' l' term (2^-22 |a b|) is dropped.  Half the matrix instructions, two thirds of the plane bytes, the same measured
// margins (profiles/r04_split_emulation_f16.json, r04_split_vae_margins.json).
//
// Reference arithmetic this stands in for: fp32 `nn.Conv` of diffusers' FlaxAutoencoderKL (model/stable_vae_model.yaml:4-16,
// call sites agent/ldp_agent.py:59,83), SURVEY.md A.3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace ldp {

// A tensor (N, H, W, C) as planes: [plane 3][n][C/8][h][w] units of 16 bytes = 8 consecutive channels in bf16.
// Channel-blocked so that a run of pixels of one 8-channel block is contiguous: the conv's LDS-DMA staging reads
// 1 KB lines, the producer (GroupNorm apply) writes 512-byte runs.
struct PlaneGeom {
  int N, H, W, C;
  size_t plane_units() const { return (size_t)N * (C / 8) * H * W; }
  size_t bytes() const { return plane_units() * 16 * 3; }
};

struct SConvArgs {
  const void* xp;        // input planes (PlaneGeom{N, H, W, cin})
  const void* wp;        // packed weight planes, pack_sconv3()
  const float* bias;     // (cout)
  const float* res_in;   // (N, H, W, cout) fp32 added to the output, or nullptr
  float* out;            // (N, H, W, cout) fp32
  float* stats_part;     // per 256-pixel tile (sum, sum of squares) of every output column, or nullptr:
                         //   [(n * tiles_per_image + tile) * cout + column] * 2 + {0, 1}
  const void* zero;      // >= 16 zero bytes in device memory (source of the zero padding)
  int N, H, W, cin, cout;
  int dbg = 0;           // timing ablations, honoured by -DLDP_ABLATE builds only (tools/)
  int npl = 3;           // operand planes: 3 = bf16 (h, m, l), six products; 2 = fp16 (h, l' = (x - h) * 2^11), three products (planes_launch / pack_sconv3 alike)
};

// 3x3, stride 1, pad 1.  W in {64, 32, 16} (square images), cin % 16 == 0, cin <= 256, cout % 128 == 0.
bool sconv3_supported(int H, int W, int cin, int cout);
int sconv3_launch(const SConvArgs& a, hipStream_t s);        // 0 or a hipError_t / -100 (unsupported shape)

// (3, 3, cin, cout) Flax kernel -> [cout/128][cin/16][dh][dw][plane][k half][128 columns][8 channels] bf16:
// the LDS image of one (chunk, dh) iteration is one contiguous 36 KB block
std::vector<uint16_t> pack_sconv3(const float* k, int cin, int cout, int npl = 3);

// y planes = split(act(GroupNorm(x))) or split(x):  x (N, HW, C) fp32 NHWC, stats (N, G, 2) = (mean, rstd) or nullptr
// range (npl = 2 only): device word set to 1 when an element is outside the fp16 planes' range (|x| >= 65504 or not finite), or nullptr
int planes_launch(const float* x, const float* stats, const float* scale, const float* bias, void* planes,
                  int N, int HW, int C, int G, int act, hipStream_t s, int npl = 3, unsigned int* range = nullptr);
bool fits_f16_planes(const float* w, size_t n);      // every |w| < 65504 and finite (engine.hip)

uint16_t f32_to_bf16_rne(float f);
float bf16_to_f32(uint16_t b);

}  // namespace ldp
