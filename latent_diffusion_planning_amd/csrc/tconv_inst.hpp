// tconv_inst.hpp -- launcher body shared by the per-mode instantiation units.
#pragma once
#include "tconv.hpp"

namespace ldp {

template <int MODE, int TO, int NWN, int KS, int CPI, bool RES_OUT, int MB = 1, bool KWS = false, int SPLIT = 0>
static int init_one() {
  using C = TConvCfg<MODE, TO, NWN, KS, CPI, MB, SPLIT>;
  auto kern = tconv_kernel<MODE, TO, NWN, KS, CPI, RES_OUT, MB, KWS, SPLIT>;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
}

template <int MODE, int TO, int NWN, int KS, int CPI, bool RES_OUT, int MB = 1, bool KWS = false, int SPLIT = 0>
static int launch_one(const ConvArgs& a_in, hipStream_t stream) {
  using C = TConvCfg<MODE, TO, NWN, KS, CPI, MB, SPLIT>;
  ConvArgs a = a_in;
  auto kern = tconv_kernel<MODE, TO, NWN, KS, CPI, RES_OUT, MB, KWS, SPLIT>;
  const int ncb = a.cout / C::BN;
  const int nsb = (a.B + 16 * MB - 1) / (16 * MB);
  const int cs = a.cs > 1 ? a.cs : 1;
  if (cs != 1 && cs != 2 && cs != 4) return (int)hipErrorInvalidValue;
  // fp32 tiles: two row blocks + column split measured slower three times (HISTORY 6; round 5: profiles/r05_t2_mb2_ab.txt), never the default:
  // only the quarter-group T = 2 tile of that A/B (option t2_mb2) may be launched this way
  if (MB > 1 && cs > 1 && !SPLIT && !(MODE == MODE_K5 && TO == 2 && NWN == 2 && cs == 4)) return (int)hipErrorInvalidValue;
  if ((a.flags & ~mode_flag_mask(MODE)) != 0 || (a.flags & mode_flag_forced(MODE)) != mode_flag_forced(MODE))
    return (int)hipErrorInvalidValue;            // feature compiled out of / always on in this mode
  // x = GroupNorm group (fastest: a group's work-groups share an XCD), y = column part (x zf when
  // there are more than 32768 sample blocks), z = sample block
  // a GroupNorm launch normalises over the columns of its cs work-groups: they must be exactly one group
  if ((a.flags & EP_GN) && MODE == MODE_K5 && a.cout != 8 * C::BN * cs) return (int)hipErrorInvalidValue;
  const int kw = a.kw > 1 ? a.kw : 1;
  if (kw > 1 && (!KWS || !tconv_kw_ok(MODE, TO, NWN, MB) || nsb * ncb * kw > 256 || (kw & (kw - 1)) || kw > KW_MAX || !a.kw_slab || a.kw_slot < 0 || a.kw_slot > 63 ||
                 ((a.ca + a.cb) / C::CH_IT) % kw != 0))
    return (int)hipErrorInvalidValue;
  const int zf = (nsb + 32767) / 32768;
  const int gz = (nsb + zf - 1) / zf;
  // by_sample = 2: group-major grid with the two-dimensional XCD placement (tconv.hpp) -- split tiles only, 8 groups, whole quads of sample blocks
  if ((a.by_sample & 2) && !(SPLIT != 0 && LDP_KERNARG_PRELOAD && ncb / cs == 8 && kw == 1 && zf == 1 && gz % 4 == 0)) a.by_sample = 0;
  if (a.by_sample & 1) {
    if (kw > 1 || zf > 1) return (int)hipErrorInvalidValue;
    // x rounded up to a multiple of 8: block id % 8 (the XCD) then depends on the sample block alone whatever
    // the batch size; the padding work-groups leave at once (sb * 16 >= B)
    int gx = (gz + 7) & ~7;
    if (a.sb_qs > 0) {
      if (MODE != MODE_P1) return (int)hipErrorInvalidValue;
      gx = ((((gz + (1 << a.sb_qs) - 1) >> a.sb_qs) + 7) & ~7) << a.sb_qs;       // whole sample blocks, 8 at a time
    }
    hipLaunchKernelGGL(kern, dim3(gx, cs, ncb / cs), dim3(C::NT), C::LDS_BYTES, stream, LDP_KERNEL_ARGS(a, false));
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3(ncb / cs, cs * kw * zf, gz), dim3(C::NT), C::LDS_BYTES, stream, LDP_KERNEL_ARGS(a, zf > 1));
  return (int)hipGetLastError();
}

// key: mode | TO<<4 | NWN<<12 | KS<<16 | CPI<<20 | res<<24 | (MB-1)<<25 | kws<<26 | split<<27 (2 bits)
constexpr uint32_t plan_key(int mode, int to, int nwn, int ks, int cpi, int res, int mb = 1, int kws = 0, int split = 0) {
  return (uint32_t)mode | ((uint32_t)to << 4) | ((uint32_t)nwn << 12) | ((uint32_t)ks << 16) |
         ((uint32_t)cpi << 20) | ((uint32_t)res << 24) | ((uint32_t)(mb - 1) << 25) | ((uint32_t)kws << 26) | ((uint32_t)split << 27);      // split: 3 bits
}
// split-operand instantiations (MB = 2, plain k = 5)
#define LDP_CASE_S(MODE, TO, NWN, KS, CPI, RES)                 \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 2, 0, 1):          \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 1>(a, stream);
#define LDP_INIT_S(MODE, TO, NWN, KS, CPI, RES)                                 \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 1>(); if (r_) return r_; }

// 16-row split tiles (MB = 1, v_mfma_f32_16x16x32_bf16)
#define LDP_CASE_S1(MODE, TO, NWN, KS, CPI, RES)                \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 1, 0, 1):          \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 1, false, 1>(a, stream);
#define LDP_INIT_S1(MODE, TO, NWN, KS, CPI, RES)                                \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 1, false, 1>(); if (r_) return r_; }

// 16-row matrix instruction over two row blocks per wave (MB = 2, SPLIT = 2)
#define LDP_CASE_S2(MODE, TO, NWN, KS, CPI, RES)                \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 2, 0, 2):          \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 2>(a, stream);
#define LDP_INIT_S2(MODE, TO, NWN, KS, CPI, RES)                                \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 2>(); if (r_) return r_; }

// 16-row split tiles on two fp16 planes, three products (SPLIT = 3), one or two row blocks per wave
#define LDP_CASE_SH(MODE, TO, NWN, KS, CPI, RES, MB)            \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, MB, 0, 3):         \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, MB, false, 3>(a, stream);
#define LDP_INIT_SH(MODE, TO, NWN, KS, CPI, RES, MB)                            \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, MB, false, 3>(); if (r_) return r_; }

// 32-row split tiles on two fp16 planes (SPLIT = 4, MB = 2)
#define LDP_CASE_SH32(MODE, TO, NWN, KS, CPI, RES)              \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 2, 0, 4):          \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 4>(a, stream);
#define LDP_INIT_SH32(MODE, TO, NWN, KS, CPI, RES)                              \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2, false, 4>(); if (r_) return r_; }

#define LDP_CASE(MODE, TO, NWN, KS, CPI, RES)                   \
  case plan_key(MODE, TO, NWN, KS, CPI, RES):                   \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0>(a, stream);
#define LDP_CASE2(MODE, TO, NWN, KS, CPI, RES)                  \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 2):                \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2>(a, stream);
#define LDP_INIT2(MODE, TO, NWN, KS, CPI, RES)                                  \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 2>(); if (r_) return r_; }
#define LDP_CASE3(MODE, TO, NWN, KS, CPI, RES)                  \
  case plan_key(MODE, TO, NWN, KS, CPI, RES, 1, 1):             \
    return launch_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 1, true>(a, stream);
#define LDP_INIT3(MODE, TO, NWN, KS, CPI, RES)                                  \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0, 1, true>(); if (r_) return r_; }
#define LDP_INIT(MODE, TO, NWN, KS, CPI, RES)                                   \
  { const int r_ = init_one<MODE, TO, NWN, KS, CPI, (RES) != 0>(); if (r_) return r_; }

}  // namespace ldp
