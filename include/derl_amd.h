/* derl_amd -- C-ABI of the MI355X (gfx950) on-policy hot path.
 *
 * mknbv/derl has no FFI: its extension boundary is the Python API.  This
 * header is the boundary UNDER that API: each entry point replaces the inside of one
 * reference function (cited as file:line relative to the reference root) with
 * hand-written CDNA4 HIP kernels.  Host Python (the derl_amd Python package, same class / argument
 * names as derl) binds these through ctypes; INTEGRATION.md shows the stub a derl
 * maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller allocates and owns every buffer (including workspaces); the library keeps
 *    no pointer past a call;
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); calls
 *    enqueue work asynchronously and return without synchronising;
 *  - every function returns 0 on success, a negative DX_E* code on failure and never
 *    throws; dx_last_error() returns a thread-local message for the last failure;
 *  - shapes are validated on the host before any launch (a bad shape never reaches a
 *    kernel);
 *  - the dx_cnn_* calls of ONE device must come from one host thread at a time: the native
 *    rollout, the backward of minibatches >= 2,048 samples and dx_cnn_ppo_epoch put part of
 *    their launches on side streams the library owns per device (created on first use, ordered
 *    against `stream` with events and always joined before the call's later launches on
 *    `stream`; one process per GPU -- the launcher contract -- satisfies this).
 */
#ifndef DERL_AMD_H
#define DERL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_OK 0
#define DX_EINVAL (-1)   /* bad argument (shape, null pointer, alignment)            */
#define DX_EHIP (-2)     /* a HIP runtime call failed                               */
#define DX_ENOSUP (-3)   /* configuration not supported by the compiled kernels     */
#define DX_EWS (-4)      /* workspace too small                                     */
#define DX_ETIMEOUT (-5) /* a persistent kernel's grid barrier gave up (dx_mlp_ppo_epoch) */

#define DX_ABI_VERSION 6

int dx_abi_version(void);
/* The DX_* environment switches ("Diagnostic switches") are parsed when a call first needs them and cached.
 * dx_reload_env drops the cache: every integer switch is read again at its next use (a host that changes a switch at
 * run time; the test suite, which walks the switch settings in one process).  NOT reloadable, read once per process:
 * DX_ROCTX (whether roctx ranges are emitted), DERL_AMD_RCCL_LIBRARY (which RCCL is dlopen'ed) and the diag flavour's
 * DX_COMM_TEST_HOOK ("delay_us:factor", a string); DX_CS_DIAG / DX_CS_STEP / DX_CS_VARIANT / DX_C0_DIAG /
 * DX_MLP_PERSIST_SPIN_LIMIT bypass the cache altogether (plain getenv at every call: always current). */
int dx_reload_env(void);
const char *dx_last_error(void);
/* Kernels this library has launched in this process so far (every entry point counts its own
 * launches; RCCL's and the caller's are not included): measurement aid, e.g. launches per update. */
long long dx_launch_count(void);
/* name_host: buffer of >= 256 bytes; cu_count/lds_bytes may be NULL. */
int dx_device_info(int device, char *name_host, int *cu_count, int *lds_bytes);

/* ---------------------------------------------------------------------------------
 * GAE backward scan -- replaces the loop of derl/runners/trajectory_transforms.py:45-65
 * (GAE.__call__).  Layout is the reference's np.asarray stacking: time-major (T, N)
 * row-major; `values` is the squeezed (T, N) view of the (T, N, 1) array, `resets` one
 * byte per element (numpy/torch bool), `last_values` (N) = value of
 * state["latest_observations"] (:47-50).
 *   delta_t = r_t + (1-reset_t)*gamma*v_{t+1} - v_t
 *   adv_t   = delta_t + (1-reset_t)*gamma*lambda*adv_{t+1},  adv_T = 0
 *   value_targets = adv + values                                   (:63)
 * Wave-level segmented affine scan: lanes cover envs (coalesced rows), waves cover
 * chunks of T, chunk aggregates are combined through LDS.  fp32; differs from the
 * reference's float64-intermediate evaluation by <= ~1e-6 (tests state the tolerance).
 * Algorithmic HBM traffic: 17 B per (t, n) element + 4 B per env.
 * --------------------------------------------------------------------------------- */
int dx_gae_f32(const float *rewards, const uint8_t *resets, const float *values,
               const float *last_values, int T, int N, float gamma, float lambda,
               float *advantages, float *value_targets, void *stream);


/* ---------------------------------------------------------------------------------
 * Advantage normalisation -- replaces derl/runners/trajectory_transforms.py:89-92
 * (NormalizeAdvantages): out = (a - mean) / (std + eps), population std.
 * `stats` = 3 doubles {sum, sum of squares, count}.  With stats_ready = 0 they are
 * computed here; with stats_ready = 1 the caller supplies them (e.g. after summing the
 * per-GPU shards' stats with one small all-reduce).
 * Algorithmic traffic: 4 B read + 4 B written per element (+ one read for the stats).
 * --------------------------------------------------------------------------------- */
int dx_adv_stats_f32(const float *advantages, long long n, double *stats, void *stream);
/* The statistics of ALL minibatches of one epoch in one launch: segment s is the minibatch
 * advantages[index[s*seglen .. min((s+1)*seglen, n))] (index = the epoch's composed
 * permutation, derl/runners/onpolicy.py:44-62; NULL = identity), stats = ceil(n/seglen) x 3
 * doubles, bit-identical to dx_adv_stats_f32 on the gathered minibatch.  Lets a sharded run
 * sum every minibatch's statistics over the ranks with one all-reduce per rollout. */
int dx_adv_stats_segments_f32(const float *advantages, const int32_t *index, long long n,
                              long long seglen, double *stats, void *stream);
int dx_adv_normalize_f32(const float *advantages, float *out, long long n, float eps,
                         double *stats, int stats_ready, void *stream);

/* ---------------------------------------------------------------------------------
 * Global gradient norm, clipping and the optimizer step -- replaces
 * derl/alg/common.py:56-64,76 (Trainer.preprocess_gradients = torch clip_grad_norm_,
 * optimizer.step) for the optimizers configured in derl/factory/ppo.py:78-81 (Adam,
 * eps=1e-5) and derl/factory/a2c.py:68-73 (RMSprop alpha=.99, eps=1e-5), over the FLAT
 * parameter / gradient buffers (reference state_dict order).
 *   dx_grad_sumsq_f32: partials[i] = float64 partial sums of g^2 (npartials blocks);
 *   step kernels     : coef = min(1, max_norm / (sqrt(sum partials) + 1e-6)) (max_norm <= 0:
 *                      no clipping); g <- g*coef is written back (clip is in place);
 *                      Adam   m,v update, p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);
 *                      RMSprop s update,  p -= lr * g / (sqrt(s) + eps);
 *                      norm_out (optional) receives the pre-clip norm.
 * Algorithmic traffic: Adam 28 B / parameter (read p,g,m,v; write p,m,v) + 4 B (g write).
 * --------------------------------------------------------------------------------- */
int dx_grad_sumsq_f32(const float *grads, long long n, double *partials, int npartials,
                      void *stream);
int dx_clip_adam_step_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                          long long n, const double *sumsq_partials, int npartials,
                          double max_norm, double lr, double beta1, double beta2, double eps,
                          long long step, float *norm_out, void *stream);
int dx_clip_rmsprop_step_f32(float *params, float *grads, float *square_avg, long long n,
                             const double *sumsq_partials, int npartials, double max_norm,
                             double lr, double alpha, double eps, float *norm_out, void *stream);

/* The composed minibatch permutations of ALL epochs of a rollout -- the host side of
 * derl/runners/onpolicy.py:44-49 (one np.random.permutation per epoch; the shuffles compose) --
 * drawn from NumPy's legacy MT19937 stream without the Python GIL: mt_key_host (624 words) and
 * *mt_pos_host are np.random.get_state()'s and are advanced in place (np.random.set_state puts
 * them back), orders_out_host is (epochs, n) int32 with row e = the order of epoch e.  Bit-for-bit
 * what `order = order[np.random.permutation(n)]` per epoch produces (tests/test_host_logic.py).
 * Host memory only; no GPU work. */
int dx_host_compose_permutations(uint32_t *mt_key_host, int *mt_pos_host, long long n, int epochs,
                                 int shuffle, int32_t *orders_out_host);

/* Row gather dst[i] = src[idx[i]] (rows of row_bytes bytes) -- the minibatch selection of
 * derl/runners/onpolicy.py:44-49,59-62 for the small per-sample arrays (frames are
 * gathered by index inside the conv loader instead of being copied). */
int dx_gather_rows(const void *src, const int32_t *idx, void *dst, long long nrows,
                   long long row_bytes, void *stream);
/* The same selection for up to 16 arrays sharing one index vector, in one launch (host
 * arrays of narrays device pointers / row sizes): onpolicy.py:59-62 loops over every key of
 * the interactions dict. */
int dx_gather_rows_multi(const void *const *src, void *const *dst, const long long *row_bytes,
                         int narrays, const int32_t *idx, long long nrows, void *stream);

/* Observation / return normalisation of a batched env -- replaces the per-step body of
 * derl/env/mujoco_wrappers.py:64-124 (Normalize.step / .reset / .observation) and
 * RunningMeanVar (:8-61).  obs (N, D) f32; rewards (N) f32 and resets (N) bytes may be NULL
 * (reset(): observations only).  State, float64 on the device, updated in place:
 * obs_stats = {mean[D], var[D], count} (NULL: observations pass through), ret_stats =
 * {mean, var, count} (NULL: rewards pass through unclipped), ret (N) discounted returns.
 * workspace: >= 256 * D doubles of scratch (partial moments per row block).
 * update_stats = 0 applies the current statistics without updating them. */
int dx_normalize_step_f32(const float *obs, int N, int D, const float *rewards,
                          const uint8_t *resets, double *obs_stats, double *ret_stats, double *ret,
                          double *workspace, long long workspace_count, float clipobs, float cliprew,
                          double gamma, double eps, int update_stats, float *obs_out, float *rew_out,
                          void *stream);

/* Atari frame pipeline of a batched env, uint8, batch-first.  `dones` (N bytes)
 * with `reset` frames reproduces the auto-reset of derl/env/env_batch.py:66-70; both NULL = no
 * env finished.
 *  dx_frame_max_u8   -- derl/env/atari_wrappers.py:121-137 (MaxBetweenFrames): out = max(raw,
 *                       last); last := raw, or the reset frame of an env that finished.
 *                       per_env = bytes of one frame (a multiple of 4).
 *  dx_frame_queue_u8 -- :140-163 (QueueFrames): the K most recent frames on the last axis,
 *                       stacked (H, W[, C], K) or concatenated (H, W, C*K); an env that finished
 *                       gets K copies of its reset frame.  elems = H*W*C of one frame; prev and
 *                       out are different buffers (rollout slots t and t + 1).
 *  dx_gray_resize_u8 -- :95-118 (ImagePreprocessing): BT.601 luma + bilinear resize as the
 *                       reference's cv2 calls resolve; cv2 is absent here: PARITY UNPINNED.
 * max and queue are pinned bit for bit to the reference's classes (tests/golden/atari_frames.npz). */
int dx_frame_max_u8(const uint8_t *raw, uint8_t *last, const uint8_t *dones, const uint8_t *reset,
                    uint8_t *out, int N, long long per_env, void *stream);
int dx_frame_queue_u8(const uint8_t *prev, const uint8_t *frame, const uint8_t *dones,
                      const uint8_t *reset, uint8_t *out, int N, long long elems, int C, int K,
                      int concat, void *stream);
int dx_gray_resize_u8(const uint8_t *in, uint8_t *out, int N, int H, int W, int C, int OH, int OW,
                      int gray, void *stream);

/* Episode-reward statistics of a batched env over the T steps of a rollout -- replaces
 * RewardSummarizer.step of derl/env/summarize.py:40-52 (and add_summaries :25-38) applied step
 * by step.  rewards (T, N) f32, resets (T, N) bytes.  State (device, updated in place): acc, ep_len
 * (N) f64; ended (N) bytes; queue (N, Q) f64 ring with qlen / qpos (N) int32; step_count (1) int64.
 * With record != 0, every step after which all envs have finished an episode since the last
 * summary appends a row {total_reward, episode_length, min_reward, max_reward, reward_mean_Q,
 * step_count} to rows (max_rows, 6) and increments *nrows (clamped to max_rows). */
int dx_reward_summary_f32(const float *rewards, const uint8_t *resets, int T, int N, int Q, int record,
                          double *acc, double *ep_len, uint8_t *ended, double *queue, int *qlen,
                          int *qpos, long long *step_count, double *rows, int max_rows, int *nrows,
                          void *stream);

/* ---------------------------------------------------------------------------------
 * Categorical head -- replaces the distribution part of derl/policies.py:61-80
 * (ActorCriticPolicy.act with torch Categorical) and derl/alg/ppo.py:24-108 (PPOLoss) /
 * derl/alg/a2c.py:19-79 (A2CLoss) including their backward w.r.t. the head outputs.
 * `head_out` is (B, 32) row-major: columns 0..A-1 logits, column A the value (the padded
 * output of dx_cnn_forward).
 *   act : action = inverse-CDF sample from uniforms[b] (or, if uniforms == NULL, from a
 *         counter-based generator keyed by (seed, counter, b)), log_prob, value.
 *   loss: mode 0 = PPO (clipped ratio + clipped value, cliprange < 0 means None),
 *         mode 1 = A2C.  Writes dL/dhead_out (B, 32) and loss_out[8] =
 *         {loss, policy_loss, entropy, value_loss, mean adv, mean value, mean target, r^2}.
 *         Means are over B; gradients are scaled by 1/global_batch (global_batch <= 0: B)
 *         so that sharded batches sum to the single-process gradient after all-reduce.
 *         `partials` is scratch of >= 8 * ceil(B/8) doubles.
 * --------------------------------------------------------------------------------- */
int dx_categorical_act_f32(const float *head_out, int B, int A, const float *uniforms,
                           uint64_t seed, uint64_t counter, int64_t *actions, float *log_prob,
                           float *values, void *stream);
int dx_categorical_loss_f32(const float *head_out, const int64_t *actions,
                            const float *old_log_prob, const float *advantages,
                            const float *old_values, const float *value_targets, int B, int A,
                            int mode, float cliprange, float value_loss_coef,
                            float entropy_coef, long long global_batch, float *dhead_out,
                            double *partials, int partials_capacity, float *loss_out,
                            void *stream);

/* ---------------------------------------------------------------------------------
 * Nature-DQN actor-critic network -- replaces derl/models.py:94-124 (NatureCNNBase.forward)
 * + :198-214 (NatureCNNModel.forward, output_units=[A, 1]) and the autograd backward of
 * derl/alg/common.py:70, as fp32-MFMA implicit GEMMs over NHWC activations.
 *
 * The caller fills the geometry fields and struct_bytes, calls dx_cnn_init (host only) to
 * get the derived sizes / offsets, allocates the device buffers and stores their pointers.
 * params/grads are flat in the reference's state_dict order and layout:
 *   base.conv-0.weight (32,C,8,8) .bias, conv-1 (64,32,4,4), conv-2 (64,64,3,3),
 *   base.linear (512, flat), output_layers.0 (A,512), output_layers.1 (1,512)
 * at element offsets off_w[i], off_b[i].  dx_cnn_pack must run after every parameter change.
 * --------------------------------------------------------------------------------- */
typedef struct dx_cnn_ctx {
  int struct_bytes;                 /* = sizeof(dx_cnn_ctx), checked */
  int in_h, in_w, in_c;             /* observation (84, 84, 4), NHWC */
  int num_actions;                  /* A <= 31 */
  int max_batch;                    /* capacity of the activation buffers */
  /* ---- derived by dx_cnn_init ---- */
  int h0, w0, h1, w1, h2, w2, flat; /* 20,20, 9,9, 7,7, 3136 for 84x84 */
  int reserved0;
  long long off_w[6], off_b[6];     /* conv0, conv1, conv2, linear, policy head, value head */
  long long param_count;
  long long pk_c0f, pk_c1f, pk_c2f, pk_fcf, pk_hdf, pk_hdb, pk_c1d[4], pk_c2d, pk_fcd, pk_hdd;
  long long packed_count, slab_count;
  long long y0_count, y1_count, y2_count, hid_count, head_count;  /* floats per buffer */
  long long hid_slab_count;         /* split-K partials of the linear layer's forward (rollout, small and mid-size minibatches) */
  /* offsets (in floats, inside `packed`) of the bf16 planes [3][N][K] of the NT weight mirrors:
   * conv1 fwd, conv2 fwd, linear fwd, conv1 dgrad (4 parity packs as one matrix), conv2 dgrad,
   * linear dgrad -- operands of the bf16-split GEMMs (igemm_b3.hip) */
  long long pb_c1f, pb_c2f, pb_fcf, pb_c1d, pb_c2d, pb_fcd;
  long long pb_c0f;                 /* conv0 fwd planes: operand of the rollout first-layer kernel */
  /* the factored tail (csrc/tail.hip): Wc = Wh Wfc as [8][flat] in y2's column order, beff [8],
   * and the scratch of their product -- offsets in floats inside `packed` */
  long long pk_wc, pk_beff, pk_wcs;
  /* the rollout's conv-stack kernel (csrc/convstack.hip) reads conv1 / conv2's bf16 planes and Wc from
   * copies in ITS waves' fragment order (one contiguous KB per wave load) -- offsets in floats */
  long long ps_c1f, ps_c2f, ps_wc;
  /* conv1 / conv2 data-gradient weights, three bf16 planes in the fragment order of csrc/dgrad_b6.hip */
  long long ps_c1d, ps_c2d;
  /* ---- device buffers (caller-allocated, fp32) ---- */
  float *params, *grads;            /* param_count */
  float *packed;                    /* packed_count; ZERO-FILLED once by the owner: dx_cnn_pack
                                       never writes the padding of the 32-wide head matrices */
  float *y0, *y1, *y2, *hid, *head; /* activations kept for backward */
  float *dy0, *dy1, *dy2, *dhid, *dhead; /* same sizes as the activations */
  float *slabs;                     /* slab_count: split-reduction partials of wgrad */
  float *hid_slabs;                 /* hid_slab_count */
} dx_cnn_ctx;

int dx_cnn_init(dx_cnn_ctx *ctx);
int dx_cnn_pack(const dx_cnn_ctx *ctx, void *stream);
/* obs: (B, in_h, in_w, in_c) uint8 (dequantised x/255 exactly) or float32; sample_idx
 * (optional) selects obs[sample_idx[b]] -- the minibatch permutation. */
int dx_cnn_forward(const dx_cnn_ctx *ctx, const void *obs, int obs_is_u8,
                   const int32_t *sample_idx, int B, void *stream);
int dx_cnn_backward(const dx_cnn_ctx *ctx, const void *obs, int obs_is_u8,
                    const int32_t *sample_idx, int B, void *stream);
/* The forward of a training minibatch without the heads (conv stack + linear layer -> ctx->hid), and
 * the heads with the loss in ONE launch: ctx->head = ctx->hid Wh^T + bh (derl/models.py:198-214), the
 * categorical PPO / A2C loss of derl/alg/ppo.py:24-108 / derl/alg/a2c.py:19-79 (loss_out[8] as
 * dx_categorical_loss_f32), its gradient w.r.t. the head outputs (ctx->dhead), ctx->dhid = dhead Wh
 * and the heads' weight / bias gradient slabs -- what dx_cnn_forward's last stage,
 * dx_categorical_loss_f32 and dx_cnn_backward's first two stages do in five launches.  Continue with
 * dx_cnn_backward_part(part = 2 [then 1], or 3).  norm_stats != NULL: `advantages` are raw and
 * normalised here with {sum, sumsq, n} (derl/runners/trajectory_transforms.py:89-92), written to
 * adv_normalized_out if given.  `counter`: one word, zero before the first call (the launch leaves
 * it zero).  `partials`: >= 8 * ceil(B / 8) doubles covers every route (the factored tail asks for
 * 8 * min(ceil(B / 8), 256) where its loss also runs its backward pass -- dx_cnn_tail_fused -- and 8 * min(ceil(B / 16), 256) otherwise, the layer-by-layer heads for 8 * ceil(B / 64); less is DX_EINVAL).  DX_ENOSUP where
 * dx_cnn_fused_heads(ctx) is 0 (more than 18 actions; more than 7 on the layer-by-layer route).
 *
 * Routes (same outputs and gradients, other association / arithmetic; csrc/cnn.hip):
 *  - 84 x 84 uint8 frames: the three conv layers of the whole minibatch are ONE launch of the image-resident
 *    kernel (csrc/convstack.hip: bf16 matrix cores at fp32 accuracy, ctx->y0 / y1 / y2 written for the backward);
 *    float32 frames or other sizes: three layer-by-layer launches on fp32 MFMA.
 *  - 84 x 84 frames and <= 7 actions: derl's linear layer has no activation behind it, so dx_cnn_forward_trunk ends
 *    at ctx->y2 and dx_cnn_heads_loss_f32 applies linear layer + heads as ONE affine map of y2 (csrc/tail.hip);
 *    ctx->hid is then not written.  dx_cnn_tail_factored(ctx) says which.
 *  - the conv layers' backward: data and weight gradients of conv1 / conv2 on the bf16 matrix cores, image-resident
 *    (csrc/dgrad_b6.hip, csrc/wgrad_b6.hip), for the 84 x 84 geometry; else the fp32 kernels. */
int dx_cnn_forward_trunk(const dx_cnn_ctx *ctx, const void *obs, int obs_is_u8,
                         const int32_t *sample_idx, int B, void *stream);
int dx_cnn_heads_loss_f32(const dx_cnn_ctx *ctx, const int64_t *actions, const float *old_log_prob,
                          const float *advantages, const float *old_values,
                          const float *value_targets, const double *norm_stats, float norm_eps,
                          float *adv_normalized_out, int B, int mode, float cliprange,
                          float value_loss_coef, float entropy_coef, long long global_batch,
                          double *partials, int partials_capacity, unsigned *counter,
                          float *loss_out, void *stream);
/* The same backward in two calls, for overlapping the gradient all-reduce with it (the one
 * exchange step of the path): part 0 = heads + linear layer, after which
 * grads[off_w[3] .. param_count) -- 95 % of the bytes -- are final; part 1 = the conv layers,
 * grads[0 .. off_w[3]).  Part 0 first.  After dx_cnn_heads_loss_f32: part 2 = the linear layer only
 * (in part 0's place), part 3 = linear layer + conv layers with one finalisation. */
int dx_cnn_backward_part(const dx_cnn_ctx *ctx, const void *obs, int obs_is_u8,
                         const int32_t *sample_idx, int B, int part, void *stream);
/* Rollout step of the policy -- replaces derl/policies.py:61-80 for one batch of observations:
 * for 84 x 84 uint8 frames and <= 7 actions ONE launch, one workgroup per observation (csrc/convstack.hip:
 * conv stack, y2 Wc^T + beff, Categorical sampling; ctx->y2 is written too); otherwise conv stack, the
 * 3136->512 linear layer as split-K partial slabs and one fused launch for both heads + sampling
 * (uniforms == NULL: counter-based generator keyed by (seed, counter, row)).
 * Outputs: actions int64 (B), log_prob f32 (B), values f32 (B).  ctx->head is not written. */
int dx_cnn_act(const dx_cnn_ctx *ctx, const void *obs, int obs_is_u8, int B,
               const float *uniforms, uint64_t seed, uint64_t counter, int64_t *actions,
               float *log_prob, float *values, void *stream);
/* T rollout steps against the synthetic device env, enqueued from one call -- the inner loop
 * of derl/runners/env_runner.py:43-65 for the measurement env.  Where dx_cnn_act is one launch the WHOLE
 * horizon is one launch (every env's chain frame -> policy -> sample -> next frame is local to its workgroup);
 * otherwise the same per-step launches as dx_cnn_act + dx_synth_atari_step.  Bit-identical buffers either way.
 * obs (T+1, N, H, W, 4) uint8 with obs[0] given; actions (T, N) int64; log_prob, values,
 * rewards (T, N) float32; resets (T, N) bytes. */
int dx_cnn_rollout_synth(const dx_cnn_ctx *ctx, uint8_t *obs, int T, int N, int64_t *actions,
                         float *log_prob, float *values, float *rewards, uint8_t *resets,
                         uint64_t policy_seed, uint64_t policy_counter, uint64_t env_seed,
                         uint64_t env_counter, float p_reward, float p_reset, void *stream);
/* Every minibatch update of one PPO epoch (or the one update of an A2C rollout) from ONE call --
 * the loop of derl/alg/common.py:66-78 (Trainer.step) over the minibatches of
 * derl/runners/onpolicy.py:44-62, with derl/runners/trajectory_transforms.py:84-92 in front of
 * each.  Minibatch k is samples [k * mbsize, min((k + 1) * mbsize, samples)) of the EPOCH-ORDERED
 * per-sample arrays; its frames are obs[index[...]] (index = the epoch's composed permutation,
 * gathered inside the conv loader) or, with index == NULL, the rows of obs themselves.  Same
 * launches in the same order as dx_adv_normalize_f32 -> dx_cnn_pack -> dx_cnn_forward ->
 * dx_categorical_loss_f32 -> dx_cnn_backward[_part] -> dx_grad_sumsq_f32 -> dx_clip_*_step_f32 per
 * minibatch: bit-identical results.  With `allreduce` the two halves of the flat gradient buffer
 * are summed over the ranks through the library's communicator (dx_comm_init) between / after the
 * two halves of the backward, overlapping the conv layers' backward. */
typedef struct dx_cnn_epoch {
  int struct_bytes;
  int mbsize;
  long long samples;
  const void *obs;             /* (rows, in_h, in_w, in_c) uint8 or float32                 */
  int obs_is_u8;
  int mode;                    /* 0 = PPO, 1 = A2C                                          */
  const int32_t *index;        /* (samples) rows of obs in epoch order, or NULL (identity)  */
  const int64_t *actions;      /* (samples), epoch order like every array below             */
  const float *old_log_prob;   /* (samples) or NULL (A2C)                                   */
  const float *advantages;     /* (samples), raw                                            */
  const float *old_values;     /* (samples) or NULL (A2C)                                   */
  const float *value_targets;  /* (samples)                                                 */
  int normalize;               /* 1: (a - mean) / (std + norm_eps) per minibatch            */
  float norm_eps;
  const double *stats_ready;   /* (minibatches, 3) GLOBAL {sum, sumsq, n} per minibatch (sharded
                                  runs: summed over the ranks beforehand) or NULL: local      */
  double *stats;               /* (minibatches, 3) scratch when stats_ready == NULL         */
  float *adv_normalized;       /* (samples): every minibatch's normalised advantages        */
  float cliprange, value_loss_coef, entropy_coef;
  int world;                   /* ranks: gradients are scaled by 1 / (B * world)            */
  int allreduce;               /* 1: all-reduce the gradient halves (needs dx_comm_init)    */
  int optimizer;               /* 0 = Adam (state0 = exp_avg, state1 = exp_avg_sq, beta1,
                                  beta2), 1 = RMSprop (state0 = square_avg, beta1 = alpha)   */
  int npartials;
  float *state0, *state1;
  double *sumsq_partials;
  double *loss_partials;
  int loss_partials_capacity;
  int grad_norm_stride;        /* 0: grad_norm_out[0] = the last minibatch's; 1: one each   */
  int mirrors_current;         /* 1: the packed mirrors match ctx->params (no pack before the
                                  first minibatch; every update is followed by one); 2: the
                                  previous call had more_epochs = 1 and nothing touched the
                                  parameters since -- the mirrors the training kernels read are
                                  current, the rollout's are not                               */
  int more_epochs;             /* 1: the caller's next use of ctx is another dx_cnn_ppo_epoch (a further
                                  epoch over the same rollout): the LAST update re-packs only what the
                                  training kernels read, like the updates before it; the caller
                                  then passes mirrors_current = 2 to that call, or calls dx_cnn_pack
                                  before anything else (dx_cnn_act ...) reads the mirrors.  0: the
                                  last update leaves every mirror current                       */
  unsigned *loss_counter;      /* one zeroed word: heads + loss + heads' backward run as ONE launch
                                  (dx_cnn_heads_loss_f32) where num_actions <= 7; NULL: separate
                                  heads / loss launches                                        */
  double max_grad_norm;        /* <= 0: no clipping                                         */
  double lr, beta1, beta2, opt_eps;
  long long first_step;        /* Adam's step number of minibatch 0 (1-based)               */
  float *grad_norm_out;        /* pre-clip norms (see grad_norm_stride) or NULL             */
  float *loss_out;             /* (ceil(samples / mbsize), 8)                               */
} dx_cnn_epoch;
int dx_cnn_ppo_epoch(const dx_cnn_ctx *ctx, const dx_cnn_epoch *epoch, void *stream);
/* One launch of the network (one profiler row), for per-kernel timing and layer tests.
 * Stages in execution order: 0 conv0_fwd, 1 conv1_fwd, 2 conv2_fwd, 3 fc_fwd, 4 heads_fwd,
 * 5 heads_wgrad, 6 heads_dgrad, 7 fc_wgrad, 8 fc_dgrad, 9 conv2_wgrad, 10 conv2_dgrad,
 * 11 conv1_wgrad, 12 conv1_dgrad, 13 conv0_wgrad, 14 finalize (slabs -> gradients).
 * forward = 0..4, backward = 5..14. */
/* Which kernel family the most recent launch of `stage` took in this process ("ntp" = the persistent
 * LDS-DMA ring, "wgrad_direct" / "wgrad_fc" = the image-resident / linear-layer weight gradients,
 * "conv0_b16" / "conv0_ks" = the first layer's tile kernels / its K-split weight gradient, "wgrad_b6" / "dgrad_b6" /
 * "convstack_train" = the bf16 x6 conv stages, "tail_factored", "igemm_lat", "igemm_pix", "igemm_nt", "igemm_tn",
 * "nt_dma", ...; "" before any launch).
 * The routes depend on tile counts and divisibility by 128 images: tests pin BASELINE's minibatches
 * to the fast families, bench.py prints the route of every stage. */
const char *dx_cnn_last_route(int stage);
/* 1 when this ctx takes the FACTORED tail (csrc/tail.hip): derl's linear layer has no activation behind
 * it (derl/models.py:112-115, 198-203), so linear layer + heads are one affine map out = Wc y2 + beff with
 * Wc = Wh Wfc; dx_cnn_forward_trunk then ends at y2, dx_cnn_heads_loss_f32 reads y2, dx_cnn_backward_part
 * (2 / 3) forms dL/dWfc = Wh^T (dout^T y2), dL/dWh, dL/dbfc, dL/dbh and dL/dy2 = dout Wc, and
 * dx_cnn_act / dx_cnn_rollout_synth sample from y2 Wc^T -- the same outputs and gradients as the
 * layer-by-layer route to fp32 rounding.  84 x 84 frames, <= 18 actions (the A + 1 outputs padded to 8, 16 or 24 rows);
 * DX_FC_FACTORED=0 turns it off. */
int dx_cnn_tail_factored(const dx_cnn_ctx *ctx);
/* 1 when dx_cnn_forward_trunk + dx_cnn_heads_loss_f32 apply to this ctx: up to 18 actions (the full Atari action set,
 * derl/env/make_env.py:94-106 builds heads of any width, derl/models.py:186-203) where the tail is factored, up to 7 on
 * the layer-by-layer route; 0: dx_cnn_forward + dx_categorical_loss_f32 + dx_cnn_backward. */
int dx_cnn_fused_heads(const dx_cnn_ctx *ctx);
/* 1 when dx_cnn_heads_loss_f32 on this ctx ALSO forms dy2 and the partial gradients of the linear layer + heads (the
 * factored tail with up to 7 actions: its loss and its backward pass over y2 are one launch, and dx_cnn_backward_part
 * 2 / 3 then start behind them); 0: that pass is the first launch of dx_cnn_backward_part 2 / 3.  (DX_TAIL_FUSED=0.) */
int dx_cnn_tail_fused(const dx_cnn_ctx *ctx);
int dx_cnn_stage(const dx_cnn_ctx *ctx, int stage, const void *obs, int obs_is_u8,
                 const int32_t *sample_idx, int B, void *stream);

/* ---------------------------------------------------------------------------------
 * Two-net tanh MLP actor-critic -- replaces derl/models.py:224-237 (MLP) + :240-271
 * (MuJoCoModel.forward; also the vector-observation categorical policy of BASELINE config 1)
 * and its autograd backward.  Flat params/grads in state_dict order:
 *   [logstd (P) if has_logstd] module_list.0.{0,2,4}.{weight,bias} module_list.1.{0,2,4}.{...}
 * with module_list.0 = policy net (obs_dim -> 64 -> 64 -> P), module_list.1 = value net
 * (-> 1).  Head output (B, 32): columns 0..P-1 policy outputs (Gaussian mean or logits),
 * column P the value.  dx_mlp_pack after every parameter change (a no-op when the observation fits
 * 64 columns: forward and backward then run as ONE launch each for both nets, csrc/mlp_fused.hip,
 * straight from the canonical parameters; wider observations go layer by layer through the
 * implicit-GEMM kernels and need the packed mirrors).
 * --------------------------------------------------------------------------------- */
typedef struct dx_mlp_ctx {
  int struct_bytes;
  int obs_dim, policy_out, has_logstd, max_batch;
  /* ---- derived by dx_mlp_init ---- */
  int obs_pad, reserved0, reserved1;
  long long off_logstd;             /* -1 without logstd */
  long long off_w[6], off_b[6];     /* net0 L0,L1,L2, net1 L0,L1,L2 */
  long long param_count;
  long long pk_f0[2], pk_d2[2], pk_d1[2];
  long long packed_count, slab_per_net, slab_count;
  long long x_count, h_count, head_count;
  /* ---- device buffers ---- */
  float *params, *grads, *packed;
  float *xpad;                      /* x_count */
  float *h1[2], *h2[2];             /* h_count each */
  float *head, *dhead;              /* head_count each; head must start zeroed */
  float *da, *db;                   /* h_count each (backward scratch) */
  float *slabs;                     /* slab_count */
} dx_mlp_ctx;

int dx_mlp_init(dx_mlp_ctx *ctx);
int dx_mlp_pack(const dx_mlp_ctx *ctx, void *stream);
int dx_mlp_forward(const dx_mlp_ctx *ctx, const float *obs, int B, void *stream);
int dx_mlp_backward(const dx_mlp_ctx *ctx, int B, void *stream);
/* T rollout steps of the Gaussian policy (derl/policies.py:61-80 with derl/models.py:240-271) against the MuJoCo-shaped
 * synthetic device env from ONE launch -- the inner loop of derl/runners/env_runner.py:43-65 for the measurement env:
 * every env's chain observation -> both nets -> sample -> next observation is local to its workgroup.  Buffers are
 * bit-identical to dx_mlp_forward + dx_normal_act_f32 + dx_synth_mujoco_step per step.  obs (T+1, N, obs_dim) float32
 * with obs[0] given; actions (T, N, P); log_prob, values, rewards (T, N) float32; resets (T, N) bytes.  DX_ENOSUP for a
 * categorical MLP or observations wider than 64. */
int dx_mlp_rollout_synth(const dx_mlp_ctx *ctx, float *obs, int T, int N, float *actions, float *log_prob,
                         float *values, float *rewards, uint8_t *resets, uint64_t policy_seed,
                         uint64_t policy_counter, uint64_t env_seed, uint64_t env_counter, float p_reset,
                         void *stream);

/* Every minibatch update of one epoch of the MLP actor-critic from ONE call -- the loop of
 * derl/alg/common.py:66-78 (Trainer.step) over the minibatches of
 * derl/runners/onpolicy.py:44-62, with derl/runners/trajectory_transforms.py:84-92's
 * per-minibatch advantage normalisation.  The arrays are the EPOCH's permuted copies (what
 * dx_gather_rows_multi produces); minibatch k is rows [k*mbsize, (k+1)*mbsize).  The launches are
 * those of the per-step entry points in the same order (bit-identical results); loss_out gets
 * 8 floats per minibatch (the loss kernels' terms).  Adam only (derl/factory/ppo.py:78-81);
 * `lr` is constant over the call (the schedule follows the runner's step count, which does not
 * move inside an epoch, derl/alg/common.py:72-75). */
typedef struct dx_mlp_epoch {
  int struct_bytes;
  int mbsize;
  long long samples;
  const float *obs;            /* (samples, obs_dim)                                     */
  const void *actions;         /* (samples, P) float32 (Gaussian) or (samples) int64     */
  int action_is_f32;
  int mode;                    /* 0 = PPO, 1 = A2C                                        */
  const float *old_log_prob;   /* (samples) or NULL (A2C)                                 */
  const float *advantages;     /* (samples), raw                                          */
  const float *old_values;     /* (samples) or NULL (A2C)                                 */
  const float *value_targets;  /* (samples)                                               */
  int normalize;               /* 1: (a - mean) / (std + norm_eps) per minibatch          */
  float norm_eps;
  float cliprange, value_loss_coef, entropy_coef;
  long long global_batch;      /* 0 = the minibatch size                                  */
  float *adv_normalized;       /* (samples): every minibatch's normalised advantages      */
  double *stats;               /* (3) scratch                                             */
  float *exp_avg, *exp_avg_sq; /* Adam state (param_count)                                */
  double *sumsq_partials;
  int npartials;
  int loss_partials_capacity;
  double *loss_partials;
  double max_grad_norm;        /* <= 0: no clipping                                       */
  double lr, beta1, beta2, adam_eps;
  long long first_step;        /* Adam's step number of minibatch 0 (1-based)             */
  float *grad_norm_out;        /* pre-clip norms (see grad_norm_stride), or NULL          */
  float *loss_out;             /* (ceil(samples / mbsize), 8)                             */
  int grad_norm_stride;        /* 0: grad_norm_out[0] = the last minibatch's; 1: one each */
  int persistent;              /* 1: ONE persistent launch for the whole epoch where
                                  dx_mlp_persist_plan covers the shape (see below)        */
  void *workspace;             /* persistent epoch: workspace_bytes from dx_mlp_persist_plan, ZERO before the first
                                  call (every launch leaves its barrier words zero again)   */
  long long workspace_bytes;
  double *stats_all;           /* persistent epoch with normalize: (minibatches, 3) scratch */
  unsigned *status_host;       /* persistent epoch: ONE pinned host word, zero before the first call,
                                  or NULL.  A persistent launch whose grid barrier gives up leaves
                                  the caller's parameters / moments / gradient untouched, NaN in
                                  loss_out, poisons the workspace (later launches on it exit at
                                  once) and, through an asynchronous copy behind the launch, a
                                  non-zero code here; every dx_mlp_ppo_epoch given a non-zero
                                  word returns DX_ETIMEOUT naming the barrier                  */
} dx_mlp_epoch;
int dx_mlp_ppo_epoch(const dx_mlp_ctx *ctx, const dx_mlp_epoch *epoch, void *stream);
/* The route the last dx_mlp_ppo_epoch of this process took: 1 = one persistent launch, 0 = the
 * per-stage launches, -1 = none yet (tests pin BASELINE config 3 to the persistent route). */
int dx_mlp_last_route(void);
/* The persistent form of the same epoch (csrc/mlp_persist.hip): every workgroup keeps the whole
 * model in LDS and its share of the Adam moments in registers; per minibatch forward / loss /
 * backward of its row tiles -> slab, grid barrier, fixed-order slab reduction + |g|^2 partials, grid
 * barrier, clip + Adam on every workgroup's own copy.  Two grid barriers per update instead of ~9
 * dependent launches.  Results equal the launch-per-stage epoch to float32 rounding (other partition
 * of the sums), not bit for bit.  Covered: Gaussian policy, obs_dim <= 32, >= 512-row minibatches,
 * <= 64 minibatches, single process (never while a communicator exists), and a grid of one
 * workgroup per CU that the occupancy query grants.  *workgroups = 0 when the shape is not covered (then
 * `persistent` is ignored and the launch-per-stage epoch runs). */
int dx_mlp_persist_plan(const dx_mlp_ctx *ctx, int mbsize, long long samples, int *workgroups_host,
                        long long *workspace_bytes_host);

/* Diagonal-Gaussian head -- replaces Independent(Normal(mean, exp(logstd)), 1) of
 * derl/policies.py:40-42,66,76-77 and the PPO/A2C loss on it (derl/alg/ppo.py:24-108) with
 * its gradient w.r.t. mean, value and logstd.
 *   act : a = mean + std*eps with eps ~ N(0,1) from `normals` (B, P) or, if NULL, a
 *         counter-based Box-Muller generator; log_prob (B), values (B).
 *   loss: as dx_categorical_loss_f32; additionally writes dL/dlogstd (P) to dlogstd_out
 *         (scaled by 1/global_batch like the other gradients). */
int dx_normal_act_f32(const float *head_out, const float *logstd, int B, int P,
                      const float *normals, uint64_t seed, uint64_t counter, float *actions,
                      float *log_prob, float *values, void *stream);
int dx_normal_loss_f32(const float *head_out, const float *logstd, const float *actions,
                       const float *old_log_prob, const float *advantages,
                       const float *old_values, const float *value_targets, int B, int P,
                       int mode, float cliprange, float value_loss_coef, float entropy_coef,
                       long long global_batch, float *dhead_out, float *dlogstd_out,
                       double *partials, int partials_capacity, float *loss_out, void *stream);

/* Synthetic Atari-shaped environment step (measurement input only; the
 * reference's env stack derl/env/ is out of scope): fills `frames` (nenvs x 84x84x4 uint8,
 * any byte count that is a multiple of 16) with uniform bytes, rewards in {-1,0,1} with
 * P(!=0) = p_reward, resets Bernoulli(p_reset), all hashed from (seed, counter, position). */
int dx_synth_atari_step(void *frames, long long frame_bytes_total, float *rewards,
                        uint8_t *resets, int nenvs, uint64_t seed, uint64_t counter,
                        float p_reward, float p_reset, void *stream);
/* The MuJoCo-shaped measurement env (no MuJoCo in this image): one step of nenvs envs -- observations
 * (nenvs, obs_dim <= 64) float32 N(0, 1) clipped to +-10 (the range derl/env/mujoco_wrappers.py:64-124's Normalize
 * produces), rewards N(0, 1), resets Bernoulli(p_reset), all hashed from (seed, counter, env, component). */
int dx_synth_mujoco_step(float *obs, float *rewards, uint8_t *resets, int nenvs, int obs_dim, uint64_t seed,
                         uint64_t counter, float p_reset, void *stream);

/* ---------------------------------------------------------------------------------
 * Gradient exchange.  derl has no distributed code; the step these sit
 * inside is derl/alg/common.py:66-78 (Trainer.step: backward -> [all-reduce] -> clip ->
 * optimizer step).  One process per GPU; the library owns ONE RCCL communicator (resolved
 * with dlopen at dx_comm_init -- no link-time dependency) plus a high-priority stream the
 * gradient reductions run on.
 *   dx_comm_unique_id   rank 0 makes the 128-byte id (HOST buffer); the host side hands it to
 *                       the other ranks by any means (derl_amd/distributed.py: one
 *                       torch.distributed broadcast -- the only thing torch.distributed does).
 *   dx_comm_available   side-effect-free readiness check of THIS rank: RCCL loads (dlopen; the
 *                       library named by DERL_AMD_RCCL_LIBRARY if set), no communicator exists yet,
 *                       the current device answers.  ncclCommInitRank is itself a collective, so
 *                       the ranks must agree that all of them can enter it BEFORE any does
 *                       (derl_amd/distributed.py: a MIN all-reduce of this call's outcome).
 *   dx_comm_init        collective: every rank calls it with the same id, on its own device.
 *   dx_allreduce_grads  in-place SUM of count floats, ASYNCHRONOUS to `stream`: it starts once
 *                       the work already enqueued on `stream` is done, and runs on the
 *                       library's stream; `stream` continues without waiting.  Each rank's loss
 *                       kernel scales by 1 / global_batch, so the SUM is the single-process
 *                       mean gradient.
 *   dx_allreduce_wait   makes `stream` wait for every reduction issued so far (no-op without a
 *                       communicator: single-process callers need not branch).
 *   dx_allreduce_sum_f64 / dx_comm_broadcast_f32   ordered like a kernel on `stream` (they run on
 *                       the communicator's own stream -- one communicator is only ever driven
 *                       from one stream -- between a wait for `stream` and a wait BY `stream`):
 *                       the per-minibatch advantage statistics {sum, sumsq, n} of a rollout
 *                       (derl/runners/trajectory_transforms.py:89-92 made global) and the
 *                       initial parameter broadcast.
 * World size 1 is valid (RCCL accepts one rank): the same code path, used by the GPU tests.
 * Not thread-safe against concurrent use of the same communicator.
 * --------------------------------------------------------------------------------- */
#define DX_COMM_ID_BYTES 128
int dx_comm_available(void);
int dx_comm_unique_id(void *id_out_host);
int dx_comm_init(const void *unique_id_host, int rank, int world);
/* any output may be NULL; world = 0 when there is no communicator */
int dx_comm_info(int *rank_host, int *world_host, long long *allreduces_issued_host,
                 long long *allreduce_bytes_host);
int dx_comm_destroy(void);
int dx_allreduce_grads(float *flat_grads, long long count, void *stream);
int dx_allreduce_wait(void *stream);
int dx_allreduce_sum_f64(double *buf, long long count, void *stream);
int dx_comm_broadcast_f32(float *buf, long long count, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DERL_AMD_H */
