/* libldp_hip.so -- C ABI of the MI355X-native LDP denoising hot path.
 *
 * The reference (amberxie88/latent_diffusion_planning) is pure Python on JAX/Flax; it has no
 * native layer and therefore no FFI of its own.  Each entry point below replaces one traced
 * JAX function of the reference's sampling path; a binding (ctypes, see
 * latent_diffusion_planning_amd/_lib.py and INTEGRATION.md) calls them with raw device
 * pointers.  No torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LDP_E* code; ldp_last_error() returns a
 *     thread-local message for the last failing call.  Nothing throws or exits across the ABI.
 *   - all tensors are contiguous float32 in device (HBM) memory unless marked "host";
 *     the caller owns them.  The handle owns weights, constant tables, workspaces and the
 *     hipGraphExec cache.
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises the device except ldp_finalize / destroy.
 *   - one handle per device; calls on one handle are not re-entrant.
 *   - layouts are the reference's channels-last ones: plans (B, T, D), images (N, H, W, 3).
 */
#ifndef LDP_HIP_H
#define LDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#define LDP_API __attribute__((visibility("default")))

#define LDP_OK 0
#define LDP_EINVAL (-1)   /* bad argument / unsupported shape            */
#define LDP_ESTATE (-2)   /* call order (weights missing, not finalized) */
#define LDP_EHIP (-3)     /* a HIP runtime call failed                   */
#define LDP_ENOMEM (-4)
#define LDP_EKEY (-5)     /* unknown weight path / option name           */
#define LDP_EFAULT (-6)   /* results since the last ldp_poll_fault are invalid and must be recomputed: a split
                             work-group timed out on its peer (the handle switched to safe mode), or an operand
                             left the range of the two-fp16-plane convolutions (the handle switched them to
                             three bf16 planes, which have fp32's range) -- see the fault protocol below      */

#define LDP_SAMPLER_DDPM 0 /* FlaxDDPMScheduler.step semantics (reference) */
#define LDP_SAMPLER_DDIM 1 /* eta = 0, defined by this repo */

#define LDP_MAX_LEVELS 4

typedef struct ldp_handle ldp_handle;

/* Hyper-parameters fixed at construction (agent/ldp_agent.yaml:7-34,47-51 + the dims
 * LDPAgent.create derives, agent/ldp_agent.py:534-540). */
typedef struct ldp_config {
  int32_t obs_dim;            /* D: planner input_dim                                  */
  int32_t action_dim;         /* A                                                     */
  int32_t global_cond_dim;    /* obs_horizon * D (width of obs_cond)                   */
  int32_t pred_horizon;       /* T: planner sequence length, multiple of 2^(levels-1)  */
  int32_t action_horizon;     /* IDM rows per plan                                     */
  int32_t n_levels;           /* len(down_dims) (<= LDP_MAX_LEVELS)                    */
  int32_t down_dims[LDP_MAX_LEVELS];
  int32_t kernel_size;        /* 5                                                     */
  int32_t n_groups;           /* 8                                                     */
  int32_t step_embed_dim;     /* diffusion_step_embed_dim = 256                        */
  int32_t planner_train_steps;/* planner_n_diffusion_steps = 100                       */
  int32_t idm_train_steps;    /* idm_n_diffusion_steps = 100                           */
  int32_t idm_hidden;         /* 256                                                   */
  int32_t idm_blocks;         /* 3                                                     */
  int32_t idm_time_dim;       /* FourierFeatures output_size = 256                     */
  int32_t image_size;         /* 64 (StableVAE input H = W; 96 and 128 also built); 0 disables the VAE module */
  int32_t vae_latent_channels;/* 4                                                     */
  int32_t device;             /* HIP device ordinal                                    */
} ldp_config;

LDP_API const char* ldp_last_error(void);
LDP_API const char* ldp_version(void);

/* -- lifecycle ----------------------------------------------------------------------------
 * replaces: LDPAgent.create (agent/ldp_agent.py:516-672) for the inference-side state. */
LDP_API int ldp_create(const ldp_config* cfg, ldp_handle** out);
LDP_API int ldp_destroy(ldp_handle* h);

/* Upload one parameter leaf.  `path` = "<module>/<flax path>/<leaf>" with module in
 * {"planner","idm","vae"}, e.g. "planner/ConditionalResidualBlock1D_3/Conv1dBlock_0/Conv_0/kernel".
 * `host` is a host float32 array in the Flax layout (Conv (k,Cin,Cout); Dense (in,out)).
 * replaces: planner_state.params / idm_state.params / vae_params pytrees
 * (agent/ldp_agent.py:575,614,551; train_bc.py:210-240 load_snapshot). */
LDP_API int ldp_set_weight(ldp_handle* h, const char* path, const float* host, const int64_t* shape,
                   int32_t ndim);

/* Pack weights into the MFMA streaming layout and build the timestep-only tables
 * (time-MLP output, per-block FiLM time parts, IDM cond parts, scheduler coefficients).
 * `modules` is a bitmask: 1 planner, 2 idm, 4 vae.  Synchronises `stream`. */
LDP_API int ldp_finalize(ldp_handle* h, int32_t modules, void* stream);

/* -- planner ------------------------------------------------------------------------------
 * eps = ConditionalUnet1D.apply(params, x, k, cond)   (networks/diffusion_nets_v2.py:113-169)
 * x (B,T,D), cond (B,global_cond_dim) -> eps (B,T,D).  Timestep: `k_dev` (B,) int32 device
 * array, or NULL to use the scalar `k` for every sample. */
LDP_API int ldp_unet_forward(ldp_handle* h, const float* x, const int32_t* k_dev, int32_t k,
                     const float* cond, float* eps, int32_t B, void* stream);

/* The planner fori_loop of sample_viz_step (agent/ldp_agent.py:459-476):
 *   x <- x_init (or N(0,I) from the Philox stream when x_init == NULL);
 *   for i in 0..n_steps-1: eps = unet(x, t_i, cond); x = scheduler.step(eps, t_i, x, z_i)
 * step_noise: (n_steps, B, T, D) explicit N(0,1) draws, row i used at executed step i
 * (parity mode), or NULL to draw z_i in-kernel from Philox4x32-10 keyed by
 * (seed, global element = (row_offset*T + local row)*32 + channel, step); row_offset = global
 * index of this call's first plan, so a plan's noise does not depend on how a batch is sharded.  sampler/n_steps: DDPM requires n_steps ==
 * planner_train_steps; DDIM requires n_steps | planner_train_steps.
 * use_graph != 0 replays a cached hipGraph of the whole loop (keyed by B, n_steps, sampler,
 * noise mode).  out: (B, T, D). */
LDP_API int ldp_plan_sample(ldp_handle* h, const float* cond, const float* x_init,
                    const float* step_noise, uint64_t seed, int64_t row_offset,
                    int32_t sampler, int32_t n_steps, float* out, int32_t B,
                    int32_t use_graph, void* stream);

/* -- inverse dynamics ---------------------------------------------------------------------
 * eps = MLPDiffusion.apply(params, s, a, k)   (networks/mlp_diffusion_nets.py:56-68)
 * s (R, 2D), a (R, A) -> eps (R, A). */
LDP_API int ldp_idm_forward(ldp_handle* h, const float* s, const float* a, const int32_t* k_dev,
                    int32_t k, float* eps, int32_t R, void* stream);

/* The IDM fori_loop (agent/ldp_agent.py:489-503; also :409-427, :368-386).
 * transition (R, 2D); a_init (R, A) or NULL; step_noise (n_steps, R, A) or NULL; out (R, A)
 * (still normalised; the caller applies unnormalize/clip, utils/data_utils.py:12-15,61-65).
 * row_offset = global index of this call's first row (any value; rows are keyed one by one). */
LDP_API int ldp_idm_sample(ldp_handle* h, const float* transition, const float* a_init,
                   const float* step_noise, uint64_t seed, int64_t row_offset,
                   int32_t sampler, int32_t n_steps, float* out, int32_t R,
                   int32_t use_graph, void* stream);

/* sample_viz_step without the image decode (agent/ldp_agent.py:452-505) as ONE call and, with
 * use_graph != 0, ONE captured hipGraph per (B, steps, sampler, noise modes):
 *   obs_cond = obs_emb[:, :obs_horizon].reshape(B, -1)                       (:459)
 *   x        = planner loop as in ldp_plan_sample                             (:461-476)
 *   plan     = concat(obs_emb[:, obs_horizon-1 : obs_horizon], x[:, :action_horizon])   (:478-479)
 *   s        = concat(plan[:, :-1], plan[:, 1:], -1) -> (B*action_horizon, 2D)           (:485-487)
 *   a        = IDM loop as in ldp_idm_sample on s                              (:488-503)
 *   action   = unnormalize(a) (utils/data_utils.py:12-15; act_mode 0) or clip (:61-65; act_mode 2)
 * obs_emb (B, obs_frames, D) normalised observation embeddings (get_obs_cond output), only the first
 * obs_horizon frames are read.  x_init (B,T,D) / x_noise (planner_steps,B,T,D) / a_init (B*ah,A) /
 * a_noise (idm_steps,B*ah,A): explicit-noise parity inputs or NULL for the Philox streams keyed by
 * (seed, row_offset + plan index).  Outputs: x_out (B,T,D) or NULL; plan_out (B, ah+1, D);
 * action_out (B*ah, A).  act_lo/act_hi: device bounds of length act_dim (1 or A), act_dim 0 = leave
 * the actions normalised. */
LDP_API int ldp_agent_sample(ldp_handle* h, const float* obs_emb, int32_t obs_frames, int32_t obs_horizon,
                     const float* x_init, const float* x_noise, const float* a_init,
                     const float* a_noise, uint64_t seed, int64_t row_offset, int32_t sampler,
                     int32_t planner_steps, int32_t idm_steps, float* x_out, float* plan_out,
                     float* action_out, const float* act_lo, const float* act_hi, int32_t act_dim,
                     int32_t act_mode, int32_t B, int32_t use_graph, void* stream);

/* -- StableVAE ----------------------------------------------------------------------------
 * FlaxAutoencoderKL.encode(x).latent_dist.mean   (call site agent/ldp_agent.py:55-60)
 * img (N, S, S, 3) NHWC already normalised to [-1,1] -> mean (N, S/32, S/32, latent_channels)
 * NHWC, i.e. the (h, w, c) flattening order the agent reshapes to (B, H, 16). */
LDP_API int ldp_vae_encode(ldp_handle* h, const float* img_nhwc, float* mean_out, int32_t N,
                   void* stream);

/* FlaxAutoencoderKL.decode(z).sample   (call site agent/ldp_agent.py:81-84; "next" row 8f-1)
 * z (N, S/32, S/32, latent_channels) NHWC, already un-normalised -> image (N, 3, S, S) NCHW.
 * Needs the decoder weights (vae/post_quant_conv, vae/decoder/...) to have been finalized. */
LDP_API int ldp_vae_decode(ldp_handle* h, const float* z_nhwc, float* img_nchw_out, int32_t N, void* stream);

/* -- elementwise pre/post-processing (utils/data_utils.py:9-16,61-65) ----------------------
 * y = (x - lo) / (hi - lo) * 2 - 1            (normalize != 0)
 * y = clip((x + 1) / 2 * (hi - lo) + lo, lo, hi)   (normalize == 0)
 * y = clip(x, lo, hi)                               (normalize == 2; the clip_min/clip_max entries)
 * lo/hi: device arrays of length `dim` broadcast over the trailing axis (dim == 1: scalar). */
LDP_API int ldp_normalize_bounds(const float* x, float* y, int64_t n, const float* lo, const float* hi,
                         int32_t dim, int32_t normalize, void* stream);

/* mean((a - b)^2) over n elements -> out[0] (device scalar): the `plan_mse` metric of
 * agent/ldp_agent.py:447-448,497-499 (a = sampled latents x_0, b = the batch's future latents).
 * One work-group, fixed summation order (bit-reproducible). */
LDP_API int ldp_mean_sq_diff(const float* a, const float* b, int64_t n, float* out, void* stream);

/* -- forward-only loss metrics (agent/ldp_agent.py:113-180, get_metrics_step :328-349) --------
 * noisy = FlaxDDPMScheduler.add_noise(x0, noise, t): out[r] = sqrt(abar[t[r]]) * x0[r] + sqrt(1 - abar[t[r]]) * noise[r]
 * (call sites agent/ldp_agent.py:119,136); x0 / noise / out (rows, width), t_dev (rows) int32 in [0, n_train);
 * abar = the float32 cumprod table of the squaredcos_cap_v2 schedule with n_train (<= 256) steps. */
LDP_API int ldp_add_noise(const float* x0, const float* noise, const int32_t* t_dev, int32_t n_train,
                  float* out, int64_t rows, int32_t width, void* stream);

/* out4[0..3] = min, max, mean, population std of n floats (the emb_* / action_* / <key>_min/_max
 * scalars of agent/ldp_agent.py:163-178: jnp.min / max / mean / std).  One work-group, fixed order. */
LDP_API int ldp_reduce_stats(const float* x, int64_t n, float* out4, void* stream);

/* -- training step (agent/ldp_agent.py:113-180 losses, :223-272 update / update_step, :274-323 update_mixed) ------------------------
 * The reference's step is `grads = jax.grad(loss)(params)`, `g_norm = optax.global_norm(grads)`, one `TrainState.apply_gradients` per
 * network with tx = optax.adam(warmup_cosine_decay_schedule) (:583-596, :621-634).  Here the handle keeps, per module (1 planner, 2 idm),
 * fp32 master parameters, gradients and the two Adam moments as flat arenas in the reference's Flax leaf layouts; every GEMM-shaped
 * piece of the forward and backward pass (Dense, k = 5 / stride-2 / transposed / 1x1 convolutions, dgrad and wgrad) runs on the exact-fp32
 * MFMA (csrc/train.hip).  Timesteps and noise are inputs (the reference draws them from its JAX key inside the traced step); the caller
 * evaluates the learning-rate schedule (optax evaluates it on the host-visible step count too).
 *
 * ldp_train_init: TrainStateEMA.create(params = the leaves last given with ldp_set_weight, tx = adam) -- moments zero, step 0.
 *                 Synchronises the device.  Call again after ldp_set_weight to restart from other parameters.
 * Streams: ldp_train_planner_grad and ldp_train_idm_grad write disjoint state (one workspace lane per module) and may be enqueued on two
 * different streams so that the two tapes overlap; each also forks its weight-gradient GEMMs to an internal side stream and joins it before
 * it returns to `stream`'s order.  Whatever reads a gradient (grad_norm, apply, arena, read) must be ordered after both (the caller's join). */
LDP_API int ldp_train_init(ldp_handle* h, int32_t modules, void* stream);

/* alpha * plan_loss (agent/ldp_agent.py:113-127, 146) and its gradient w.r.t. every planner leaf:
 *   noisy = add_noise(x0, noise, t); pred = ConditionalUnet1D(noisy, t, cond); loss = alpha * mean((pred - noise)^2)
 * x0 / noise (B, T, D) = obs_emb[:, obs_horizon:] and the N(0,1) draw; t_dev (B) int32 in [0, planner_train_steps); cond (B, global_cond_dim);
 * loss_out: device scalar.  The gradients replace the module's gradient arena. */
LDP_API int ldp_train_planner_grad(ldp_handle* h, const float* x0, const float* noise, const int32_t* t_dev, const float* cond,
                                   float alpha, float* loss_out, int32_t B, void* stream);

/* alpha * idm_loss (agent/ldp_agent.py:129-140, 154) and its gradient: s (R, 2D) = s_sprime rows, a0 / noise (R, A), t_dev (R). */
LDP_API int ldp_train_idm_grad(ldp_handle* h, const float* s, const float* a0, const float* noise, const int32_t* t_dev, float alpha,
                               float* loss_out, int32_t R, void* stream);

/* optax.global_norm over the gradient arenas of the listed modules (agent/ldp_agent.py:253) -> out[0] (device scalar).  Two-stage
 * reduction in a fixed order (bit-reproducible).  Called AFTER ldp_train_apply of the same gradients it costs one small launch: the optimiser
 * kernel leaves the per-stripe sums of squares behind (same stripes, same order: the same bits as the stand-alone first stage). */
LDP_API int ldp_train_grad_norm(ldp_handle* h, int32_t modules, float* out, void* stream);

/* One TrainState.apply_gradients with tx = optax.adam(lr): mu = (1 - b1) g + b1 mu; nu = (1 - b2) g^2 + b2 nu; count += 1;
 * p += -lr * (mu / (1 - b1^count)) / (sqrt(nu / (1 - b2^count)) + eps)   (optax 0.2.2 scale_by_adam, eps_root = 0; one launch over the arena).
 * lr = schedule(count before the increment), evaluated by the caller. */
LDP_API int ldp_train_apply(ldp_handle* h, int32_t module, float lr, float b1, float b2, float eps, void* stream);

/* TrainState.step of a module: *out = the count (number of ldp_train_apply calls since init); set_to >= 0 overwrites it first (checkpoint
 * restore). */
LDP_API int ldp_train_step_count(ldp_handle* h, int32_t module, int64_t set_to, int64_t* out);

/* One leaf of a module's state, Flax layout, host float32: which = 0 parameters, 1 gradients, 2 Adam mu, 3 Adam nu.  `path` is the Flax path
 * inside the module ("ConditionalResidualBlock1D_3/Conv1dBlock_0/Conv_0/kernel").  Both synchronise `stream`.
 * replaces: reading / restoring planner_state / idm_state (params, opt_state) in train_bc.py:203-240. */
LDP_API int ldp_train_read(ldp_handle* h, int32_t module, int32_t which, const char* path, float* host_out, int64_t numel, void* stream);
LDP_API int ldp_train_write(ldp_handle* h, int32_t module, int32_t which, const char* path, const float* host_in, int64_t numel,
                            void* stream);

/* The flat fp32 arena of a module's state on the device (which as above): *dev_out = its base, *numel_out = its length in floats (leaves
 * in Flax layouts at fixed offsets, zero padding between them; the padding of the gradient arena is written as zeros by every
 * ldp_train_*_grad).  The pointer stays valid until the next ldp_train_init / ldp_destroy.  This is the data-parallel seam: with the batch
 * rows split over processes (the reference shards its batch with PositionalSharding, utils/py_utils.py:27-39, and XLA inserts the gradient
 * all-reduce), the caller sums the gradient arenas over the ranks with ONE collective per module between ldp_train_*_grad and
 * ldp_train_grad_norm / ldp_train_apply, in order on `stream`. */
LDP_API int ldp_train_arena(ldp_handle* h, int32_t module, int32_t which, float** dev_out, int64_t* numel_out);

/* Make the sampling path use the trained parameters: master parameters -> the handle's weight store -> ldp_finalize of the listed
 * modules (what train_bc.py:143-155 relies on when it evaluates the agent it is training).  Synchronises `stream`. */
LDP_API int ldp_train_publish(ldp_handle* h, int32_t modules, void* stream);

/* -- unit-testable primitives (one Conv1dBlock / sampling conv of the U-Net) ----------------
 * y = [FiLM](Mish(GroupNorm8(Conv1d_k5_pad2(x) + b)))  with kernel in Flax layout on the host.
 * x (B,T,Cin) device, kernel (5,Cin,Cout)/bias/gn_scale/gn_bias host; film (B, 2*Cout)
 * device or NULL; y (B,T,Cout) device.  Cin is zero-padded to a multiple of 32 internally;
 * Cout must be a multiple of 8*16.  Synchronises `stream` (it packs weights on the fly). */
LDP_API int ldp_conv1d_gn_mish_film_f32(const float* x, const float* kernel_host, const float* bias_host,
                                const float* gn_scale_host, const float* gn_bias_host,
                                const float* film, float* y, int32_t B, int32_t T, int32_t Cin,
                                int32_t Cout, void* stream);

/* y = Conv1d(k=3, stride 2, XLA 'SAME' => pads (0,1))(x)  (Downsample1d, :51-56);
 * x (B,T,C) -> y (B,T/2,C). */
LDP_API int ldp_downsample1d_f32(const float* x, const float* kernel_host, const float* bias_host,
                         float* y, int32_t B, int32_t T, int32_t C, void* stream);

/* y = ConvTranspose(k=4, stride 2, 'SAME', transpose_kernel=False)(x)  (Upsample1d, :58-63);
 * x (B,T,C) -> y (B,2T,C). */
LDP_API int ldp_upsample1d_f32(const float* x, const float* kernel_host, const float* bias_host,
                       float* y, int32_t B, int32_t T, int32_t C, void* stream);

/* y = Conv3x3(x) of the StableVAE on NHWC images: stride 1 => pad 1 (ResnetBlock2D / conv_out),
 * stride 2 => pad (0,1),(0,1) + VALID (Downsample2D).  kernel (3,3,Cin,Cout) Flax layout, host. */
LDP_API int ldp_conv2d_3x3_f32(const float* x, const float* kernel_host, const float* bias_host, float* y,
                       int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t stride,
                       void* stream);

/* The same stride-1 convolution on the bf16 matrix pipe through three-plane split operands
 * (csrc/sconv.hpp: x = h + m + l exactly, six of the nine plane products, fp32 accumulate;
 * |error| at the fp32 round-off level, see profiles/r04_split_probe.txt).  This is what
 * ldp_vae_encode / ldp_vae_decode run for their 64 / 32 / 16 pixel ResnetBlock2D convolutions
 * unless option "vae_split" is 0.  x (N,H,W,Cin) device fp32, H == W in {64, 32, 16},
 * Cin % 16 == 0, Cin <= 256, Cout % 128 == 0; res (N,H,W,Cout) device fp32 added to the result, or NULL;
 * stats_out (N*H*W/256, Cout, 2) device fp32: per 256-pixel tile (sum, sum of squares) of every
 * output column (what the following GroupNorm reads), or NULL; dual: 2 = the default form of the
 * library since late round 4: TWO fp16 planes per operand (x = h + l' / 2^11, h = fp16(x),
 * l' = fp16((x - h) * 2^11)) and THREE exact products, hh in one accumulator, hl' + l'h in a second
 * one (option "vae_split_f16"); 0 / 1 = three bf16 planes, six products, one
 * accumulator (round 4's two-accumulator A/B arm was removed in round 5).  With dual = 2, operands
 * outside the fp16 planes' range (|x| >= 65504) make this primitive rerun on the bf16 planes by itself
 * (ldp_range_fallbacks() counts).
 * Reference: fp32 nn.Conv of diffusers' ResnetBlock2D. */
LDP_API int ldp_conv2d_3x3_bf16x3(const float* x, const float* kernel_host, const float* bias_host,
                          const float* res, float* y, float* stats_out, int32_t N, int32_t H,
                          int32_t W, int32_t Cin, int32_t Cout, int32_t dual, void* stream);

/* -- introspection for bench.py -------------------------------------------------------------
 * Number of kernels enqueued by the last planner / IDM call (for a graph replay: the launches the
 * captured graph contains).  which: 0 = MFMA conv kernels (the dominant kernel), 1 = all kernels. */
LDP_API int ldp_launch_count(ldp_handle* h, int32_t which, int64_t* launches);
/* -- fault protocol of the in-launch exchanges ------------------------------------------------
 * At small batches a GroupNorm group (or a K range) is split over work-groups that exchange
 * partial results inside one launch; that needs the launch's whole grid co-resident, which holds
 * while the handle has the GPU to itself.  The wait is a bounded spin: a work-group whose peer
 * never answers stores 1 into a pinned host word instead of hanging.
 *   ldp_poll_fault : host read of the pinned words, no stream is synchronised.  *faulted != 0 if a
 *                    fault was recorded since the last poll (bit 0: exchange fault -- the handle has
 *                    switched to safe mode, no in-launch exchange at all; bit 1: range fault, below);
 *                    captured graphs are dropped (the first fault of a kind waits for the handle's own
 *                    graph replays, nothing else) and the results of calls enqueued since the previous
 *                    poll must be recomputed.  Poll after synchronising on the results.
 *   ldp_check_fault: hipStreamSynchronize(stream) + poll; LDP_EFAULT when a fault was recorded.
 * Every sampling / VAE entry point also looks at the words first and fails with LDP_EFAULT while an
 * unacknowledged fault is pending, so a caller that never polls cannot keep consuming bad results.
 *
 * Range guard of the split-operand convolutions.  Above 256 plans and in the StableVAE the large
 * convolutions run on the 16-bit matrix pipe with every fp32 operand split into TWO fp16 planes
 * (x = h + l' / 2^11: 22 significand bits).  fp16 planes hold |x| < 65504 where the
 * reference's fp32 nn.Conv holds 3.4e38.  Weights are checked when their planes are packed (a conv
 * with |w| >= 65504 keeps three bf16 planes, which have fp32's range); activations are checked by the
 * kernels that touch them anyway (the plane producer of the StableVAE convs; the planner tiles see
 * the non-finite sum of squares an overflowed plane always produces).  A violation raises the second
 * pinned word: the calls since the last poll are reported faulted exactly like an exchange fault, and
 * the handle runs every split convolution on three bf16 planes from then on ("range_fallback" = 1,
 * readable / resettable through the options; "range_faults_seen" counts).  Results are then those of
 * fp32-range arithmetic: never inf / NaN where the reference's are finite. */
LDP_API int ldp_poll_fault(ldp_handle* h, int32_t* faulted);
LDP_API int ldp_check_fault(ldp_handle* h, void* stream);

/* Times a handle-less primitive (ldp_conv2d_3x3_bf16x3 with dual = 2) left the fp16 planes for the
 * bf16 ones because a weight or an activation was outside their range (process-wide; tests). */
LDP_API int64_t ldp_range_fallbacks(void);

/* -- runtime options --------------------------------------------------------------------------
 * Work-splitting switches (results stay correct to fp32 round-off): "no_csplit", "no_mb2",
 * "no_kw", "kw_min_it", "kw_bmax", "by_sample" (XCD placement threshold), "idm_hs",
 * "idm_rt_major", "idm_noring", "idm_unfused", "safe_mode", "range_fallback"; "vae_split" (1: the StableVAE's
 * large 3x3 convs on split operands of the 16-bit matrix pipe, 0: exact-fp32 MFMA), "vae_split_f16"
 * (1: two fp16 planes / three products, 0: three bf16 planes / six), "vae_split_s2" (its stride-2 convs on fp16 planes too); the planner
 * above 256 plans: "planner_split" (0: exact fp32 everywhere), "planner_split_f16" (as for the
 * StableVAE), "planner_split_mb2", "planner_split_t16"; the IDM above 256 plans: "idm_f16" (1: its MLPResNet blocks on two fp16 planes over 32-row
 * tiles, 0: exact fp32), "idm_f16_min_rows", "idm_f16_hs"; and the A/B switches listed in
 * csrc/engine.hpp; read-only counters "stat_mb2_launches", "stat_f16_launches".  Timing ablations for
 * tools/ (results WRONG by construction): "dbg" (bit mask), "repeat".  Test hook: "inject_fault" (1: an
 * exchange fault, 2: a range fault).
 * Read-only through ldp_get_option: "any_debug", "faults_seen", "range_faults_seen", "n_cu", "graphs".
 * Nothing is ever read from the environment. */
LDP_API int ldp_set_option(ldp_handle* h, const char* name, int64_t value);
LDP_API int ldp_get_option(ldp_handle* h, const char* name, int64_t* value);

/* -- noise source primitives (tests) ----------------------------------------------------------
 * The in-kernel generator is Philox4x32-10 with counter (elem lo, elem hi, step, stream_id) and key
 * (seed lo, seed hi); element i of a call uses elem0 + i.  raw: 4 uint32 words per element
 * (out_dev has 4n words); normal: one N(0,1) per element (Box-Muller on words 0, 1). */
LDP_API int ldp_philox_raw(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id,
                   uint32_t* out_dev, int64_t n, void* stream);
LDP_API int ldp_philox_normal(uint64_t seed, uint64_t elem0, uint32_t step, uint32_t stream_id,
                      float* out_dev, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDP_HIP_H */
