// All minibatch updates of one PPO / A2C epoch of the MLP actor-critic enqueued from ONE native
// call -- the loop of derl/alg/common.py:66-78 (Trainer.step: loss -> backward -> clip ->
// optimizer step) over the minibatches of derl/runners/onpolicy.py:44-62 with the per-minibatch
// advantage normalisation of derl/runners/trajectory_transforms.py:84-92.  The launches are
// exactly the ones the per-step entry points issue (dx_adv_stats / normalize, dx_mlp_forward,
// dx_normal_loss / dx_categorical_loss, dx_mlp_backward, dx_grad_sumsq, dx_clip_adam_step), in
// the same order on the same buffers: results are bit-identical, only the host interpreter is
// gone from between them (BASELINE config 3 issues 320 updates of ~95 us of GPU work per
// rollout; ~120 us of Python per update made it host-bound).
#include "common.hpp"

namespace dx {
long long mlp_persist_workspace_bytes(int G);
int mlp_persist_workgroups(const dx_mlp_ctx *c, int mbsize, long long samples);
int launch_mlp_persist_epoch(const dx_mlp_ctx *c, const dx_mlp_epoch *e, int G, hipStream_t stream);
int mlp_persist_check_status(const dx_mlp_epoch *e);
int mlp_last_route();
void mlp_note_route(int route);
}  // namespace dx

extern "C" int dx_mlp_last_route(void) { return dx::mlp_last_route(); }

extern "C" int dx_mlp_persist_plan(const dx_mlp_ctx *c, int mbsize, long long samples, int *workgroups,
                                   long long *workspace_bytes) {
  DX_TRACE("dx_mlp_persist_plan");
  DX_REQUIRE(c != nullptr && workgroups != nullptr && mbsize >= 1 && samples >= 1, "dx_mlp_persist_plan: bad arguments");
  const int G = dx::mlp_persist_workgroups(c, mbsize, samples);
  *workgroups = G;
  if (workspace_bytes) *workspace_bytes = G > 0 ? dx::mlp_persist_workspace_bytes(G) : 0;
  return DX_OK;
}

extern "C" int dx_mlp_ppo_epoch(const dx_mlp_ctx *c, const dx_mlp_epoch *e, void *stream) {
  DX_TRACE("dx_mlp_ppo_epoch");
  DX_REQUIRE(c != nullptr && e != nullptr, "dx_mlp_ppo_epoch: null argument");
  DX_REQUIRE(e->struct_bytes == static_cast<int>(sizeof(dx_mlp_epoch)),
             "dx_mlp_ppo_epoch: struct size mismatch (caller %d, library %d)", e->struct_bytes,
             static_cast<int>(sizeof(dx_mlp_epoch)));
  DX_REQUIRE(e->samples >= 1 && e->mbsize >= 1 && e->mbsize <= c->max_batch,
             "dx_mlp_ppo_epoch: %lld samples in minibatches of %d (max_batch %d)", e->samples, e->mbsize,
             c->max_batch);
  DX_REQUIRE(e->obs && e->actions && e->advantages && e->value_targets && (!e->normalize || (e->adv_normalized && e->stats)) &&
                 e->exp_avg && e->exp_avg_sq && e->sumsq_partials && e->loss_partials && e->loss_out,
             "dx_mlp_ppo_epoch: null buffer");
  DX_REQUIRE(e->mode == 1 || (e->old_log_prob && e->old_values), "dx_mlp_ppo_epoch: PPO needs the rollout's log_prob / values");
  DX_REQUIRE(c->has_logstd ? e->action_is_f32 == 1 : e->action_is_f32 == 0,
             "dx_mlp_ppo_epoch: Gaussian policies take float32 actions, categorical ones int64");
  // a persistent epoch that gave up earlier on this workspace: fail here, whatever route this call
  // would take (the caller's parameters were left untouched by that epoch)
  if (int rc = dx::mlp_persist_check_status(e)) return rc;
  if (e->persistent && e->global_batch <= 0) {  // one persistent launch where the shape is covered
    const int G = dx::mlp_persist_workgroups(c, e->mbsize, e->samples);
    if (G > 0) return dx::launch_mlp_persist_epoch(c, e, G, dx::as_stream(stream));
  }
  const int P = c->policy_out, D = c->obs_dim;
  dx::mlp_note_route(0);
  int k = 0;
  for (long long start = 0; start < e->samples; start += e->mbsize, ++k) {
    const int B = static_cast<int>(e->samples - start < e->mbsize ? e->samples - start : e->mbsize);
    const float *adv = e->advantages + start;
    if (e->normalize) {  // what NormalizeAdvantages launches per minibatch; kept for the caller
      float *norm = e->adv_normalized + start;
      if (int rc = dx_adv_normalize_f32(adv, norm, B, e->norm_eps, e->stats, 0, stream)) return rc;
      adv = norm;
    }
    if (int rc = dx_mlp_pack(c, stream)) return rc;  // no-op for the fused kernels
    if (int rc = dx_mlp_forward(c, e->obs + start * D, B, stream)) return rc;
    const float *olp = e->old_log_prob ? e->old_log_prob + start : nullptr;
    const float *ov = e->old_values ? e->old_values + start : nullptr;
    const long long gb = e->global_batch > 0 ? e->global_batch : B;
    float *loss = e->loss_out + 8LL * k;
    if (c->has_logstd) {
      if (int rc = dx_normal_loss_f32(c->head, c->params + c->off_logstd,
                                      static_cast<const float *>(e->actions) + start * P, olp, adv, ov,
                                      e->value_targets + start, B, P, e->mode, e->cliprange, e->value_loss_coef,
                                      e->entropy_coef, gb, c->dhead, c->grads + c->off_logstd, e->loss_partials,
                                      e->loss_partials_capacity, loss, stream))
        return rc;
    } else {
      if (int rc = dx_categorical_loss_f32(c->head, static_cast<const int64_t *>(e->actions) + start, olp, adv, ov,
                                           e->value_targets + start, B, P, e->mode, e->cliprange,
                                           e->value_loss_coef, e->entropy_coef, gb, c->dhead, e->loss_partials,
                                           e->loss_partials_capacity, loss, stream))
        return rc;
    }
    if (int rc = dx_mlp_backward(c, B, stream)) return rc;
    if (int rc = dx_grad_sumsq_f32(c->grads, c->param_count, e->sumsq_partials, e->npartials, stream)) return rc;
    if (int rc = dx_clip_adam_step_f32(c->params, c->grads, e->exp_avg, e->exp_avg_sq, c->param_count,
                                       e->sumsq_partials, e->npartials, e->max_grad_norm, e->lr, e->beta1, e->beta2,
                                       e->adam_eps, e->first_step + k,
                                       e->grad_norm_out ? e->grad_norm_out + static_cast<long long>(e->grad_norm_stride) * k : nullptr,
                                       stream))
      return rc;
  }
  return DX_OK;
}
