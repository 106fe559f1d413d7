// All minibatch updates of one PPO epoch (or the single update of an A2C rollout) of the
// Nature-CNN actor-critic enqueued from ONE native call -- the loop of derl/alg/common.py:66-78
// (Trainer.step: loss -> backward -> clip -> optimizer step) over the minibatches of
// derl/runners/onpolicy.py:44-62 with the per-minibatch advantage normalisation of
// derl/runners/trajectory_transforms.py:84-92.  Per minibatch: normalise -> pack -> forward
// (frames gathered by the epoch's index inside the conv loader) -> fused loss -> backward of the
// linear layer + heads -> [all-reduce of that tail of the gradient buffer, started on the
// library's RCCL stream] -> backward of the conv layers -> [all-reduce of the head of the buffer]
// -> norm -> clip + Adam / RMSprop.  The launches are the ones the per-step entry points issue, in
// the same order on the same buffers (bit-identical results); what is gone is the host
// interpreter between them -- at the 8-GPU shard size of BASELINE config 2 (32 envs per GPU) the
// interpreter took as long as the kernels -- and the exchange step now sits INSIDE the call
// (dx_allreduce_grads between the two halves of the backward).
#include <cstdlib>

#include "common.hpp"
#include "igemm.hpp"

namespace dx {
bool comm_active();
int comm_allreduce_async(float *buf, long long count, hipStream_t stream);
int comm_wait(hipStream_t stream);
// cnn.hip: pieces of dx_cnn_forward / dx_cnn_pack and the library's side stream
int cnn_forward_range(const dx_cnn_ctx *c, int first, int last, const void *obs, int obs_is_u8,
                      const int32_t *sample_idx, int B, hipStream_t s);
int cnn_pack_part(const dx_cnn_ctx *c, int part, hipStream_t s);
int cnn_pack_between_updates(const dx_cnn_ctx *c, bool last_update, int obs_is_u8, hipStream_t s);
bool cnn_fc_factored(const dx_cnn_ctx *c);
bool cnn_forward_fused(const dx_cnn_ctx *c, int obs_is_u8);
hipStream_t cnn_side_begin(hipStream_t s);
int cnn_side_end(hipStream_t s);
}  // namespace dx

using namespace dx;

static int epoch_body(const dx_cnn_ctx *c, const dx_cnn_epoch *e, void *stream, bool &tail_on_side);

extern "C" int dx_cnn_ppo_epoch(const dx_cnn_ctx *c, const dx_cnn_epoch *e, void *stream) {
  DX_TRACE("dx_cnn_ppo_epoch");
  bool tail_on_side = false;  // part of an update's tail is still on the library's side stream
  const int rc = epoch_body(c, e, stream, tail_on_side);
  // only after a failed launch: the caller's stream never runs ahead of the side stream
  if (tail_on_side) (void)cnn_side_end(as_stream(stream));
  return rc;
}

static int epoch_body(const dx_cnn_ctx *c, const dx_cnn_epoch *e, void *stream, bool &tail_on_side) {
  DX_REQUIRE(c != nullptr && e != nullptr, "dx_cnn_ppo_epoch: null argument");
  DX_REQUIRE(e->struct_bytes == static_cast<int>(sizeof(dx_cnn_epoch)),
             "dx_cnn_ppo_epoch: struct size mismatch (caller %d, library %d)", e->struct_bytes,
             static_cast<int>(sizeof(dx_cnn_epoch)));
  DX_REQUIRE(c->struct_bytes == static_cast<int>(sizeof(dx_cnn_ctx)) && c->param_count > 0,
             "dx_cnn_ppo_epoch: ctx not initialised by dx_cnn_init");
  DX_REQUIRE(e->samples >= 1 && e->mbsize >= 1 && e->mbsize <= c->max_batch,
             "dx_cnn_ppo_epoch: %lld samples in minibatches of %d (max_batch %d)", e->samples, e->mbsize,
             c->max_batch);
  DX_REQUIRE(e->obs && e->actions && e->advantages && e->value_targets && e->state0 && e->sumsq_partials &&
                 e->loss_partials && e->loss_out && c->grads && c->dhead,
             "dx_cnn_ppo_epoch: null buffer");
  DX_REQUIRE(!e->normalize || (e->adv_normalized && (e->stats_ready || e->stats)),
             "dx_cnn_ppo_epoch: normalisation needs adv_normalized and statistics (given or scratch)");
  DX_REQUIRE(e->mode == 1 || (e->mode == 0 && e->old_log_prob && e->old_values),
             "dx_cnn_ppo_epoch: mode 0 (PPO) needs the rollout's log_prob / values; mode must be 0 or 1");
  DX_REQUIRE(e->optimizer == 1 || (e->optimizer == 0 && e->state1), "dx_cnn_ppo_epoch: optimizer 0 = Adam (two "
             "state buffers), 1 = RMSprop");
  DX_REQUIRE(e->world >= 1, "dx_cnn_ppo_epoch: world < 1");
  DX_REQUIRE((e->more_epochs == 0 || e->more_epochs == 1) && e->mirrors_current >= 0 && e->mirrors_current <= 2,
             "dx_cnn_ppo_epoch: more_epochs must be 0 / 1 and mirrors_current 0 / 1 / 2");
  DX_REQUIRE(!e->allreduce || comm_active(), "dx_cnn_ppo_epoch: allreduce requested without a communicator "
             "(dx_comm_init)");
  DX_REQUIRE(e->loss_partials_capacity >= 8 * ((e->mbsize + 7) / 8), "dx_cnn_ppo_epoch: loss_partials too small");
  hipStream_t s = as_stream(stream);
  const long long tail = c->off_w[3];  // grads[tail ..) = linear layer + heads
  // heads + loss + heads' backward in one launch where the action count allows (dx_cnn_fused_heads), like the
  // per-update path (models._cnn_loss_forward_backward): the same kernels on both paths
  const bool fused_heads = e->loss_counter != nullptr && dx_cnn_fused_heads(c) != 0;
  // with the fused launch the normalisation happens inside it: the statistics of EVERY minibatch of
  // the epoch come from one launch (bit-identical to the per-minibatch kernel's) instead of one
  // normalisation launch per update
  const double *stats_all = e->stats_ready;
  if (e->normalize && fused_heads && stats_all == nullptr) {
    if (int rc = dx_adv_stats_segments_f32(e->advantages, nullptr, e->samples, e->mbsize, e->stats, stream)) return rc;
    stats_all = e->stats;
  }
  // DX_EPOCH_TAIL_OVERLAP=1 (an experiment, off by default): step and re-pack the first conv layer
  // first and leave the rest of an update's tail to the side stream, beside the next minibatch's
  // first layer.  Measured SLOWER (same box, alternating): 256 envs 43.5 -> 44.1 ms per iteration,
  // 32 envs 11.33 -> 12.17 ms -- a dependency between two HIP streams costs tens of microseconds on
  // this stack, more than the ~35 us of small launches it hides (the backward's side stream pays the
  // same price and wins only because it hides whole GEMM stages).
  const bool tail_overlap = DX_ENV("DX_EPOCH_TAIL_OVERLAP", 0) != 0;
  int k = 0;
  for (long long start = 0; start < e->samples; start += e->mbsize, ++k) {
    const int B = static_cast<int>(e->samples - start < e->mbsize ? e->samples - start : e->mbsize);
    const float *adv = e->advantages + start;
    if (e->normalize && !fused_heads) {  // what NormalizeAdvantages launches per minibatch; kept for the caller
      float *norm = e->adv_normalized + start;
      // sharded: the GLOBAL {sum, sumsq, n} of this minibatch, summed over the ranks beforehand
      double *stats = e->stats_ready ? const_cast<double *>(e->stats_ready) + 3LL * k : e->stats;
      if (int rc = dx_adv_normalize_f32(adv, norm, B, e->norm_eps, stats, e->stats_ready != nullptr, stream)) return rc;
      adv = norm;
    }
    // (after every update below the mirrors are repacked; the first minibatch needs a pack only when
    // the caller's parameters changed since its last pack)
    if (k == 0 && !e->mirrors_current)
      if (int rc = dx_cnn_pack(c, stream)) return rc;
    const int32_t *idx = e->index ? e->index + start : nullptr;
    const void *obs = e->obs;
    if (!idx)  // no gather: minibatch k is rows [start, start + B) of obs
      obs = static_cast<const char *>(e->obs) +
            start * c->in_h * c->in_w * c->in_c * (e->obs_is_u8 ? 1 : 4);
    const float *olp = e->old_log_prob ? e->old_log_prob + start : nullptr;
    const float *ov = e->old_values ? e->old_values + start : nullptr;
    if (tail_on_side) {
      // the previous update left everything but the first conv layer's parameters and mirrors to the
      // side stream: this minibatch's first layer runs beside that, the rest of the forward after it
      if (int rc = cnn_forward_range(c, ST_CONV0_FWD, ST_CONV0_FWD, obs, e->obs_is_u8, idx, B, s)) return rc;
      tail_on_side = false;
      if (int rc = cnn_side_end(s)) return rc;
      const int trunk_last = cnn_fc_factored(c) ? ST_CONV2_FWD : ST_FC_FWD;
      if (int rc = cnn_forward_range(c, ST_CONV1_FWD, fused_heads ? trunk_last : ST_HEADS_FWD, obs, e->obs_is_u8, idx, B, s)) return rc;
    } else if (fused_heads) {
      if (int rc = dx_cnn_forward_trunk(c, obs, e->obs_is_u8, idx, B, stream)) return rc;
    } else {
      if (int rc = dx_cnn_forward(c, obs, e->obs_is_u8, idx, B, stream)) return rc;
    }
    if (fused_heads) {
      const double *st = e->normalize ? stats_all + 3LL * k : nullptr;
      if (int rc = dx_cnn_heads_loss_f32(c, e->actions + start, olp, e->advantages + start, ov, e->value_targets + start,
                                         st, e->norm_eps, e->normalize ? e->adv_normalized + start : nullptr, B, e->mode,
                                         e->cliprange, e->value_loss_coef, e->entropy_coef,
                                         static_cast<long long>(B) * e->world, e->loss_partials,
                                         e->loss_partials_capacity, e->loss_counter, e->loss_out + 8LL * k, stream))
        return rc;
    } else {
      if (int rc = dx_categorical_loss_f32(c->head, e->actions + start, olp, adv, ov, e->value_targets + start, B,
                                           c->num_actions, e->mode, e->cliprange, e->value_loss_coef, e->entropy_coef,
                                           static_cast<long long>(B) * e->world, c->dhead, e->loss_partials,
                                           e->loss_partials_capacity, e->loss_out + 8LL * k, stream))
        return rc;
    }
    if (e->allreduce) {
      if (int rc = dx_cnn_backward_part(c, obs, e->obs_is_u8, idx, B, fused_heads ? 2 : 0, stream)) return rc;
      if (int rc = comm_allreduce_async(c->grads + tail, c->param_count - tail, s)) return rc;
      if (int rc = dx_cnn_backward_part(c, obs, e->obs_is_u8, idx, B, 1, stream)) return rc;
      if (int rc = comm_allreduce_async(c->grads, tail, s)) return rc;
      if (int rc = comm_wait(s)) return rc;
    } else if (fused_heads) {
      if (int rc = dx_cnn_backward_part(c, obs, e->obs_is_u8, idx, B, 3, stream)) return rc;
    } else {
      if (int rc = dx_cnn_backward(c, obs, e->obs_is_u8, idx, B, stream)) return rc;
    }
    if (int rc = dx_grad_sumsq_f32(c->grads, c->param_count, e->sumsq_partials, e->npartials, stream)) return rc;
    float *norm_out = e->grad_norm_out ? e->grad_norm_out + static_cast<long long>(e->grad_norm_stride) * k : nullptr;
    const bool more = start + e->mbsize < e->samples;
    if (e->optimizer == 0 && more && tail_overlap && !cnn_forward_fused(c, e->obs_is_u8)) {
      // Another minibatch follows: the first conv layer's parameters (weight + bias: the head of the
      // parameter vector) are stepped and re-packed FIRST, on this stream; the other 99.5 % of the
      // parameters and their mirrors follow on the side stream, beside the next minibatch's first
      // layer.  The same element-wise kernels over two ranges: the parameters are bit-identical to the
      // one-launch step's.
      const long long head_n = c->off_b[0] + 32;  // conv0 weight + bias (multiple of 4 floats)
      if (int rc = dx_clip_adam_step_f32(c->params, c->grads, e->state0, e->state1, head_n, e->sumsq_partials,
                                         e->npartials, e->max_grad_norm, e->lr, e->beta1, e->beta2, e->opt_eps,
                                         e->first_step + k, norm_out, stream))
        return rc;
      if (int rc = cnn_pack_part(c, 1, s)) return rc;
      hipStream_t side = cnn_side_begin(s);
      if (side == nullptr) return DX_EHIP;
      tail_on_side = true;
      int rc = dx_clip_adam_step_f32(c->params + head_n, c->grads + head_n, e->state0 + head_n, e->state1 + head_n,
                                     c->param_count - head_n, e->sumsq_partials, e->npartials, e->max_grad_norm, e->lr,
                                     e->beta1, e->beta2, e->opt_eps, e->first_step + k, nullptr, side);
      if (rc == DX_OK) rc = cnn_pack_part(c, 2, side);
      if (rc != DX_OK) return rc;
      continue;
    }
    if (e->optimizer == 0) {
      if (int rc = dx_clip_adam_step_f32(c->params, c->grads, e->state0, e->state1, c->param_count, e->sumsq_partials,
                                         e->npartials, e->max_grad_norm, e->lr, e->beta1, e->beta2, e->opt_eps,
                                         e->first_step + k, norm_out, stream))
        return rc;
    } else {
      if (int rc = dx_clip_rmsprop_step_f32(c->params, c->grads, e->state0, c->param_count, e->sumsq_partials,
                                            e->npartials, e->max_grad_norm, e->lr, e->beta1, e->opt_eps, norm_out, stream))
        return rc;
    }
    // the mirrors follow every update (after the last one: everything, the next rollout can act at once)
    // (more_epochs: the next call is another epoch of this rollout -- its kernels read what a minibatch of this one reads)
    if (int rc = cnn_pack_between_updates(c, (!more && e->more_epochs == 0) || !fused_heads, e->obs_is_u8, s)) return rc;
  }
  return DX_OK;
}
