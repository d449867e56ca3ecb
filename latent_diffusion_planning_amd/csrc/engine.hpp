// engine.hpp -- host-side state of one ldp_handle: weight store, packed device weights,
// constant tables, per-batch workspaces and the hipGraphExec cache.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "tconv.hpp"

namespace ldp {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// one convolution's device-resident parameters
struct ConvW {
  DevBuf w, bias, gn_scale, gn_bias, bres;      // w holds nj (+1 with has_res: the 1x1 projection) taps per chunk
  DevBuf wsplit;                                // the same kernel as three bf16 planes, or empty: 3x3 convs of the StableVAE (sconv.hpp); planner k = 5 convs on 32-row tiles (tconv SPLIT)
  DevBuf wsplith;                               // 32-row tiles' weights as two fp16 planes (tconv SPLIT = 4), or empty
  DevBuf wsplit16h;                             // the same convs as two fp16 planes (tconv SPLIT = 3, option planner_split_f16), or empty
  DevBuf wsplit16;                              // planner k = 5 convs on 16-row split tiles (tconv SPLIT, MB = 1), or empty
  bool f16_refused = false;                     // a weight of this conv is outside the fp16 planes' range (|w| >= 65504 or not finite): bf16 planes / exact fp32 only
  int nj = 0, cin = 0, cout = 0, cin_p = 0, cout_p = 0;
  bool has_gn = false, has_res = false;
};

struct ResBlock {
  ConvW c1, c2;          // c1 carries the fused 1x1 residual projection when proj
  bool proj = false;
  int cin = 0, cout = 0;
  int film_off = 0;      // column offset of this block's (scale|bias) slice in the FiLM tables
  std::string prefix;    // weight-store path of the block (the plane-packed copies are built lazily from it)
};

struct PlannerState {
  bool ready = false;
  int D = 0, DP = 0, G = 0, T = 0, L = 0, E = 0, n_train = 0;
  int C0P = 128;                  // input-channel chunk the first conv is packed for (DP real, rest zero)
  std::vector<int> dims;
  std::vector<ResBlock> blocks;
  std::vector<ConvW> down, up;
  ConvW fin_block, fin_conv;
  int F = 0;                      // total FiLM width = sum 2*Cout
  DevBuf film_t;                  // (n_train, F): Mish(temb_k) @ W[:E] + b
  DevBuf wfilm_g;                 // (G, F): rows E.. of every block's FiLM Dense kernel
  // workspaces (sized for ws_B samples)
  int ws_B = 0;
  DevBuf state, cond, film_g, bufA, bufB, bufC, bufR, noise, xchg, kw_slab;
  size_t xchg_stride = 0;         // granules per conv launch
  uint64_t calls = 0;             // planner calls on this handle (slab hygiene, see planner_prepare)
  std::vector<DevBuf> skip;
};

struct IdmState {
  bool ready = false;
  int D = 0, A = 0, AP = 0, H = 0, NB = 0, n_train = 0, TD = 0;
  ConvW in_a;                      // (A, H) rows of MLPResNet_0/Dense_0 (kept in Flax layout, VALU)
  DevBuf w_in_s, b_in;             // (2D, H) rows + bias
  DevBuf ctab;                     // (n_train, H): cond_k @ W0[A+2D:]
  struct Blk { DevBuf ln_s, ln_b; ConvW d0, d1; };
  std::vector<Blk> blks;
  ConvW out;                       // (H -> AP)
  DevBuf wout_t, bout;             // fused path: Dense_1 kernel transposed to (A, H) + bias (A)
  int ws_R = 0;
  DevBuf state, state2, spart, h0, h1, y, z, noise, trans, part0, part1;
  int state_cur = 0;               // fused path: which of state / state2 holds the current a_t
};

struct GraphKey {
  int kind, B, n_steps, sampler, noise_mode;      // kind 0 planner loop, 1 IDM loop, 2 joint sample()
  int n_steps2 = 0, noise_mode2 = 0;              // joint graph: the IDM loop's step count / noise mode
  bool operator<(const GraphKey& o) const {
    if (kind != o.kind) return kind < o.kind;
    if (B != o.B) return B < o.B;
    if (n_steps != o.n_steps) return n_steps < o.n_steps;
    if (sampler != o.sampler) return sampler < o.sampler;
    if (noise_mode != o.noise_mode) return noise_mode < o.noise_mode;
    if (n_steps2 != o.n_steps2) return n_steps2 < o.n_steps2;
    return noise_mode2 < o.noise_mode2;
  }
};

// Runtime options (ldp_set_option).  The first group changes how work is split over work-groups
// (results stay correct, to fp32 round-off); `dbg` / `repeat` are timing ablations for tools/
// (results wrong by construction; bench.py marks such a run INVALID).  Nothing here is read from
// the environment.
struct Options {
  int no_csplit = 0;      // one work-group per GroupNorm group at any B
  int no_mb2 = 0;         // one 16-sample row block per work-group at any B
  int planner_split_8w = 1;  // 257 .. 512 plans (one work-group per CU): the four-wave fp16-plane tiles of the T = 4 / T = 8 layers as eight waves (twice the K slices)
  int planner_split_t2res16 = 1;  // 353 .. 512 plans: the T = 2 convs with the projection on fp16 planes as 16-row tiles over whole groups (else: bf16 planes, 32-row tile over half groups)
  int planner_split_t2all16 = 1;  // ... and the T = 2 convs WITHOUT the projection likewise (0: the 32-row fp16 tile over half groups with its in-launch statistics exchange)
  int t2_mb2 = 0;         // A/B (round 5): the 1024 -> 1024 T = 2 convs at 129..256 plans as quarter groups x two row blocks (profiles/r05_t2_mb2_ab.txt)
  int no_kw = 0;          // no K split over work-groups
  int kw_min_it = 1, kw_bmax = 128;
  int vae_split = 1;      // StableVAE stride-1 3x3 convs at 64 / 32 / 16 pixels on split bf16 operands (sconv.hpp: 6 plane products, fp32 accumulate); 0 = exact-fp32 MFMA
  int vae_split_f16 = 1;  // those convs on TWO fp16 planes / THREE products (sconv3 NPL = 2: x = h + l' / 2^11) instead of three bf16 planes / six (0, A/B)
  int vae_split_s2 = 1;   // the StableVAE's stride-2 convs (Downsample2D) on two fp16 planes too (tconv MODE_K3S, SPLIT = 3; round 5)
  int vae_split_gn_only = 0; // split operands only behind a GroupNorm (the resnet convs), not for the decoder's upsampler convs on raw inputs (A/B)
  int vae_no_conv_stats = 0;    // GroupNorm statistics always by their own pass (cross-check of the sums the 3x3 convs leave in their epilogue)
  int vae_no_conv_in_stats = 0; // the same for conv_in
  int planner_split = 1;  // planner: k = 5 convs of the 256- / 512- / 1024-channel levels on split bf16 operands (tconv SPLIT) above 256 plans, i.e. where one
                          // work-group owns a whole GroupNorm group (0: exact-fp32 kernels everywhere; 2: at any batch whose plan has no column / K split -- tests); the
                          // plane-packed weights are built at the first call that needs them
  int planner_split_cs2 = 1;    // the T = 2 layers around 512 plans on 32-row split tiles with every GroupNorm group over two work-groups (0: exact fp32 there)
  int planner_split_updown = 1; // the stride-2 / transposed convs between the levels on 16-row split tiles too (0: exact fp32; A/B)
  int planner_split_c256 = 1;   // 0: the 256-channel level stays on the exact-fp32 kernel (A/B)
  int planner_split_mb2 = 1;    // 16-row split tiles of the 1024-channel T = 4 layers over two row blocks per wave (tconv SPLIT = 2) once 32-sample
                                // work-groups cover the chip (from 993 plans); 0 = one row block (A/B; the plans are bit-identical either way)
  int planner_split_f16 = 1;    // the k = 5 split tiles on TWO fp16 planes and THREE products (tconv SPLIT = 3 / 4: x = h + l' / 2^11) instead of three
                                // bf16 planes and six: half the matrix instructions, two thirds of the operand bytes, the same margins; 0 = the bf16 form (A/B)
  int planner_split_t16 = 1;    // pred_horizon 16's (16, 256) level on fp16 planes too (needs planner_split_f16; 0: exact fp32 there, A/B)
  int planner_split_t2res = 1;  // the two T = 2 convs with the projection on 16-row fp16 tiles over two row blocks per wave from 993 plans (0: 32-row bf16 tiles; A/B)
  int up_full_depth = 0;  // transposed convs on 256-channel chunks (the round-2 choice) instead of 128
  int no_fin_rows = 0;    // final 1x1 conv over whole samples (round-2 launch shape) instead of position pairs
  int no_batch_split = 0; // never run the leading power-of-two part of an in-between batch as its own loop (batch_split())
  int idm_unfused = 0;    // IDM as one launch per Dense / LayerNorm (the round-1 path)
  int by_sample = 2;      // XCD affinity by sample block while weights < by_sample x input activations (0: always by group)
  int place2d = 0;          // split tiles (> 256 plans): two-dimensional XCD placement of the group-major launches (tconv.hpp; A/B)
  int idm_noring = 0;     // fused IDM: never use the ringed (one work-group per CU) variant
  int idm_rt_major = 1;   // fused IDM: XCD affinity by row tile (1) or by hidden slice (0)
  int idm_stream = -1;    // fused IDM: K-partials non-temporal (1), plain (0), by row count (-1)
  int idm_hs = 0;         // hidden slices per row tile of the fused IDM block (0 = by row count)
  int idm_f16 = 1;        // fused IDM blocks on two fp16 planes / three products over 32-row tiles (idm.hip idm_block_h16_kernel) for every batch above 256 plans (idm_f16_min_rows rows)
  int idm_f16_min_rows = 1040;      // (the first 16-row bucket above 256 plans x 4 rows)
  int idm_f16_hs = 0;     // A/B: hidden slices of that kernel (0 = four, 2)
  int train_fuse_reduce = 1; // training GEMMs: the last work-group of a split-K tile adds the partials up inside the launch (0: a reduce launch per GEMM; A/B)
  int train_sides = 1;       // training: side streams per module the weight-gradient work is dealt to (1 .. 3)
  int train_gn4 = 1;         // training: the one-pass GroupNorm kernels where a (sample, group) block has exactly 256 values (0: the generic two-pass kernels; A/B)
  int train_group_proj = 1;  // training: a projection block's two convolutions over its input (and their two data gradients) as one launch each (0: two; A/B)
  int train_intra_split = 0; // training GEMMs: the first factor of two of a K split inside the work-group (two wave quartets, hand-over through LDS) instead of over
                            // work-groups (1; measured 3.45 - 3.55 ms per step against 3.40 - 3.42: no partial block and no ticket, but 512-thread work-groups place worse)
  int train_streams = 1;     // training: weight-gradient GEMMs and parameter column sums on a side stream next to the data-gradient chain (0: one stream; A/B)
  int train_split = 1;      // training GEMMs: split the K steps of a launch over work-groups until the grid fills the chip (0: never; A/B)
  int train_wg_target = 384; // ... until the launch has this many work-groups (192 / 384 / 768 / 1536: 5.47 / 4.99 / 5.19 / 5.67 ms per step)
  int train_big = 0;        // training GEMMs: 128 x 128 tiles where both M and N reach 128 (1; measured slower than the 32 / 64-row tiles with split K: profiles/r06_update_gemm_ab.txt)
  int train_small_wg = 1 << 20; // training GEMMs (train.hip seg_gemm): 32-row tiles when the 64-row tiling has fewer work-groups than this.  Final tree at 256
                            // samples: 128 / 256 / 512 / 768 / always = 3.52 / 3.44 / 3.38 / 3.35 / 3.33 ms per step (the 64-row tiles were worth it before the
                            // in-launch split-K finish and the side streams: 5.44 -> 4.99 ms then)
  int dbg = 0, repeat = 1;
  int64_t timeline_ptr = 0;   // device buffer of tools/timeline.py (64 slots x 1 MiB); only -DLDP_TIMELINE builds write to it
  bool any_debug() const { return dbg != 0 || repeat != 1 || timeline_ptr != 0; }
};

struct GraphEntry {
  hipGraphExec_t exec;
  int64_t conv_launches, total_launches;
  uint64_t last_use = 0;          // handle's graph clock at the last replay (LRU eviction)
  hipEvent_t done = nullptr;      // recorded behind every replay on the caller's stream: what drop_graphs waits for (not the whole device)
};

}  // namespace ldp

struct ldp_handle {
  ldp_config cfg;
  std::map<std::string, ldp::HostTensor> weights;
  ldp::PlannerState pl;
  ldp::IdmState idm;
  // device control words, one block of 4 uint64 per loop: {seed, first global row, call epoch, unused};
  // block 0 = planner, block 1 = IDM.  Written by set_seed_launch on the caller's stream before a loop
  // (or its captured graph) runs, so captured graphs never bake a seed.
  ldp::DevBuf seed;
  uint64_t epoch_planner = 0, epoch_idm = 0;   // host-side call epochs, handed to set_seed_launch (never read back from the device)
  uint64_t* ctl_planner() const { return seed.as<uint64_t>(); }
  uint64_t* ctl_idm() const { return seed.as<uint64_t>() + 4; }
  // Fault word: pinned host memory mapped into the device.  A split work-group whose peer never
  // answered (bounded spin) stores 1 here; the host reads it without synchronising anything.
  volatile unsigned int* fault_host = nullptr;
  unsigned int* fault_dev = nullptr;
  int fault_pending = 0;                 // faults seen and not yet acknowledged through ldp_poll_fault (bit 0: exchange, bit 1: fp16-plane range)
  int64_t faults_seen = 0;
  bool safe_mode = false;                // after a fault: no in-launch cross-work-group exchange any more
  // Range guard of the two-fp16-plane operand form: fp16 planes hold |x| < 65504 where fp32 (and the three
  // bf16 planes) hold 3.4e38.  Word 1 of the pinned fault block is raised by the producers / consumers of fp16 planes
  // when an operand left that range (planes_kernel: |x| >= 65504 or not finite; tconv F16 tiles: a non-finite conv
  // output, which an overflowed plane always produces).  The host then treats the calls since the last poll as
  // faulted, and runs every split conv on three bf16 planes (fp32 range) from then on.
  bool range_fallback = false;           // sticky until ldp_set_option("range_fallback", 0)
  int64_t range_faults_seen = 0;
  unsigned int* range_dev() const { return fault_dev ? fault_dev + 1 : nullptr; }
  bool f16_planner() const { return opt.planner_split_f16 != 0 && !range_fallback; }
  bool f16_vae() const { return opt.vae_split_f16 != 0 && !range_fallback; }
  ldp::Options opt;
  ldp::DevBuf plan_out, act_out, obs_last;   // joint sample(): handle-owned outputs the graph writes
  hipStream_t cap_stream = nullptr;      // internal stream used only for graph capture
  int n_cu = 256;                        // compute units of cfg.device (co-residency bound of the column split)
  // Captured loops, least-recently-used eviction at `graph_cap` entries.  Batches are bucketed to whole 16-row tiles
  // (bucket_rows): the env harness changes B call to call (utils/rm_env_utils.py:150-199) and a best-of-N service
  // asks for arbitrary N; every B of a bucket replays the same graph.
  std::map<ldp::GraphKey, ldp::GraphEntry> graphs;
  std::vector<ldp::GraphEntry> retired_execs;   // evicted from the cache, possibly still running: destroyed at the next idle point
  uint64_t graph_clock = 0;
  int graph_cap = 32;
  int64_t graphs_captured = 0, graphs_evicted = 0;
  int64_t last_conv_launches = 0, last_total_launches = 0;
  int64_t stat_f16_launches = 0;         // likewise: conv launches on fp16 planes
  std::map<uint64_t, int64_t> plan_log;       // every distinct tconv instantiation this handle launched, keyed by plan_key (option "dump_plans" prints it: tools/r5/plans_used.py)
  int64_t stat_mb2_launches = 0;         // conv launches enqueued (eagerly or into a capture) on two-row-block split tiles since ldp_create: read-only option
  void* vae = nullptr;                   // VaeState (vae.hip)
  void* train = nullptr;                 // Trainer (train.hip): master parameters, gradients, Adam moments, launch tables; created by ldp_train_init
};

namespace ldp {
std::map<std::string, int64_t> tconv_plan_log();       // tconv_misc.hip: every instantiation this process launched (a snapshot, as text)
uint64_t plan_key(const ConvPlan& p);
std::string plan_text(uint64_t key);
// schedule tables (host, float64 -> float32), mirror of schedule.py
void make_step_coefs(int n_train, int n_steps, int sampler, std::vector<StepCoef>& out);
void sinusoid_table(int n, int dim, bool cos_first, std::vector<float>& out);

// weight packing into the lane-linear MFMA fragment layout
std::vector<float> pack_conv(const float* w, int nj, int cin, int cout, int cin_p, int cout_p);

bool fits_f16_planes(const float* w, size_t n);
std::vector<uint16_t> pack_conv_split16h(const float* w, int nj, int cin, int cout);
int get_weight(ldp_handle* h, const std::string& path, const HostTensor** out,
               std::initializer_list<int64_t> shape);
int make_conv(ldp_handle* h, const std::string& prefix, int nj, int cin, int cout, int cin_p,
              int cout_p, const char* gn_prefix, hipStream_t s, ConvW& out);

void drop_graphs(ldp_handle* h);
void reap_retired_graphs(ldp_handle* h);
// Rows a sampling loop is launched over: B rounded up to whole 16-row MFMA tiles.  The tiles compute all 16 rows
// anyway, rows never mix (GroupNorm / LayerNorm statistics are per row, MFMA rows are independent), the workspaces
// are sized in whole tiles, and only the B real rows are copied in and out (eagerly, outside the captured loop): the
// dead rows chew on whatever the buffers hold.  A row's bits therefore depend on its 16-bucket only, and every B of a
// bucket shares one launch plan and one captured graph.  Explicit step noise is laid out (steps, B, ...), i.e. its
// stride is B itself: those (parity-test) calls keep the exact B.
inline int bucket_rows(int B, bool explicit_noise) { return explicit_noise ? B : (B + 15) / 16 * 16; }
// the pieces of ldp_plan_sample / ldp_idm_sample, shared with the joint ldp_agent_sample
struct LoopSpec { int n_steps = 0, sampler = 0; bool explicit_noise = false; };
int check_sampler(int sampler, int n_steps, int n_train, const char* what);
int entry_fault_check(ldp_handle* h);
int planner_pre(ldp_handle* h, const float* cond, const float* x_init, const float* step_noise, uint64_t seed,
                int64_t row_offset, const LoopSpec& L, int B, hipStream_t s);
int planner_loop(ldp_handle* h, int B, const LoopSpec& L, hipStream_t q);
int idm_pre(ldp_handle* h, const float* transition, const float* a_init, const float* step_noise, uint64_t seed,
            int64_t row_offset, const LoopSpec& L, int R, hipStream_t s);
int idm_loop(ldp_handle* h, int R, const LoopSpec& L, hipStream_t q);
const float* idm_result(ldp_handle* h);          // (R, AP) padded a_0 after idm_loop
int idm_workspace(ldp_handle* h, int R);
int planner_workspace(ldp_handle* h, int B);
int run_or_replay(ldp_handle* h, const GraphKey& key, bool use_graph, hipStream_t s,
                  const std::function<int(hipStream_t)>& enqueue);
int planner_finalize(ldp_handle* h, hipStream_t s);
int planner_forward_launch(ldp_handle* h, int B, const int* k_dev, int k, bool step,
                           const StepCoef* coef, const float* noise, int step_idx, float* eps_out,
                           hipStream_t s);
int idm_finalize(ldp_handle* h, hipStream_t s);
int vae_finalize(ldp_handle* h, hipStream_t s);
void vae_destroy(ldp_handle* h);
void train_destroy(ldp_handle* h);
void betas_squaredcos(int n, std::vector<float>& betas, std::vector<float>& alphas, std::vector<float>& acp);
}  // namespace ldp
