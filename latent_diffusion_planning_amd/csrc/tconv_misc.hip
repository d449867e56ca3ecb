// Downsample1d (k3 s2), Upsample1d (transposed k4 s2), 1x1 convs (final conv + scheduler step, IDM dense layers) and the top-level dispatcher
#include "tconv_inst.hpp"
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#define LIST(X) \
  X(MODE_DOWN, 4, 2, 4, 1, 0) \
  X(MODE_DOWN, 2, 4, 2, 2, 0) \
  X(MODE_DOWN, 8, 2, 2, 1, 0) \
  X(MODE_DOWN, 4, 4, 2, 1, 0) \
  X(MODE_UP, 4, 4, 2, 4, 0) \
  X(MODE_UP, 8, 2, 4, 2, 0) \
  X(MODE_UP, 8, 4, 2, 2, 0) \
  X(MODE_UP, 16, 2, 2, 1, 0) \
  X(MODE_UP, 16, 1, 4, 1, 0) \
  X(MODE_DOWN, 8, 1, 4, 1, 0) \
  X(MODE_P1, 8, 2, 4, 1, 0) \
  X(MODE_P1, 16, 2, 2, 1, 0) \
  X(MODE_P1, 4, 2, 4, 2, 0) \
  X(MODE_P1, 4, 2, 2, 1, 0) \
  X(MODE_P1, 4, 8, 1, 2, 0) \
  X(MODE_DOWN, 4, 1, 8, 1, 0) \
  X(MODE_DOWN, 2, 2, 4, 2, 0) \
  X(MODE_UP, 4, 2, 4, 4, 0) \
  X(MODE_UP, 8, 1, 8, 2, 0) \
  X(MODE_P1, 8, 1, 8, 1, 0) \
  X(MODE_P1, 4, 4, 2, 2, 0) \
  X(MODE_P1, 4, 1, 8, 2, 0) \
  X(MODE_P1, 2, 2, 4, 1, 0) \
  X(MODE_UP, 4, 2, 4, 2, 0) \
  X(MODE_UP, 8, 1, 8, 1, 0)
// small batches: the stride-2 / transposed convs with the K split over work-groups (half-depth chunks for the latter)
#define LIST3(X) \
  X(MODE_DOWN, 4, 1, 8, 1, 0) \
  X(MODE_DOWN, 2, 2, 4, 2, 0) \
  X(MODE_UP, 4, 2, 4, 2, 0) \
  X(MODE_UP, 8, 1, 8, 1, 0) \
  X(MODE_DOWN, 4, 4, 2, 1, 0) \
  X(MODE_UP, 8, 4, 2, 2, 0)
namespace ldp {
int tconv_launch_misc(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  switch (plan_key(p.mode, p.to, p.nwn, p.ks, p.cpi, p.res_out, 1, p.kws)) {
    LIST(LDP_CASE)
    LIST3(LDP_CASE3)
    default: return -100;
  }
}
int tconv_init_misc() {
  LIST(LDP_INIT)
  LIST3(LDP_INIT3)
  return 0;
}
// every distinct instantiation launched by this process, whoever asked (engine loops, StableVAE, primitives): option "dump_plans" = 2
// prints it; tests/conftest.py collects it over the -m gpu suite (which instantiations does anything still use? profiles/r05_plans_used.txt).
// Keyed on the packed plan fields: a launch costs one uncontended lock and a lookup in a map of a few dozen integers (no formatting, no allocation
// after the first launch of a plan -- ADVICE r5); the text is made when somebody asks.
uint64_t plan_key(const ConvPlan& p) {
  return (uint64_t)(p.mode & 15) | (uint64_t)(p.to & 255) << 4 | (uint64_t)(p.nwn & 255) << 12 | (uint64_t)(p.ks & 255) << 20 | (uint64_t)(p.cpi & 255) << 28 |
         (uint64_t)(p.res_out ? 1 : 0) << 36 | (uint64_t)(p.mb & 15) << 37 | (uint64_t)(p.kws ? 1 : 0) << 41 | (uint64_t)(p.split & 15) << 42;
}
std::string plan_text(uint64_t k) {
  char key[96];
  snprintf(key, sizeof key, "mode=%d to=%d nwn=%d ks=%d cpi=%d res=%d mb=%d kws=%d split=%d", (int)(k & 15), (int)(k >> 4 & 255), (int)(k >> 12 & 255), (int)(k >> 20 & 255),
           (int)(k >> 28 & 255), (int)(k >> 36 & 1), (int)(k >> 37 & 15), (int)(k >> 41 & 1), (int)(k >> 42 & 15));
  return key;
}
namespace {
std::mutex& plan_mu() { static std::mutex mu; return mu; }
std::map<uint64_t, int64_t>& plan_counts() { static std::map<uint64_t, int64_t> m; return m; }
}  // namespace
std::map<std::string, int64_t> tconv_plan_log() {      // a snapshot, as text
  std::lock_guard<std::mutex> lock(plan_mu());
  std::map<std::string, int64_t> out;
  for (const auto& kv : plan_counts()) out[plan_text(kv.first)] = kv.second;
  return out;
}

int tconv_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
  {
    // (handles on different threads may launch at the same time: the log is the one piece of process-wide state on this path)
    std::lock_guard<std::mutex> lock(plan_mu());
    plan_counts()[plan_key(p)]++;
  }
  if (p.split) return tconv_launch_split(p, a, stream);
  if (p.mode == MODE_K5) return p.res_out ? tconv_launch_k5r(p, a, stream) : tconv_launch_k5(p, a, stream);
  if (mode_2d(p.mode)) return tconv_launch_2d(p, a, stream);
  return tconv_launch_misc(p, a, stream);
}
int tconv_init_all() {
  int r = tconv_init_k5();
  if (!r) r = tconv_init_k5r();
  if (!r) r = tconv_init_misc();
  if (!r) r = tconv_init_2d();
  if (!r) r = tconv_init_split();
  return r;
}
}  // namespace ldp
