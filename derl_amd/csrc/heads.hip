// Distribution heads and fused loss forward+backward (gfx950).
//
// Categorical head on the padded head output `out[b][0..31]` (columns 0..A-1 logits,
// column A the value; produced by the heads GEMM): one half-wave (32 lanes) per row, lane j
// owns column j, softmax reductions by cross-lane shuffles.
//   rollout : sample by inverse CDF from a uniform, log_prob, value
//             (derl/policies.py:61-80 -- ActorCriticPolicy.act, Categorical)
//   training: PPO clipped-ratio / clipped-value loss with entropy bonus, or A2C loss, and
//             their gradient w.r.t. logits and value in the same pass
//             (derl/alg/ppo.py:45-64,82-98,104; derl/alg/a2c.py:32-33,57,74; closed forms
//             in SURVEY.md Appendix A.1-A.5)
// Diagonal-Gaussian head for the MLP policy: thread per row (derl/policies.py:40-42,66).
#include "heads_dev.hpp"
#include <type_traits>
#include "synth_dev.hpp"

namespace {

constexpr int kHeadLd = 32;  // padded head width

__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct Softmax {
  float logp, p, lse;
};

// lane j < A holds logit j; returns this lane's log-prob / prob (0 for padding lanes)
__device__ __forceinline__ Softmax half_softmax(float logit, bool is_logit) {
  const float mx = half_max(is_logit ? logit : -INFINITY);
  const float e = is_logit ? expf(logit - mx) : 0.f;
  const float s = half_sum(e);
  Softmax r;
  r.lse = mx + logf(s);
  r.logp = is_logit ? logit - r.lse : 0.f;
  r.p = e / s;
  return r;
}

__global__ __launch_bounds__(256) void categorical_act_kernel(
    const float *__restrict__ out, int B, int A, const float *__restrict__ uniforms, uint64_t seed,
    uint64_t counter, int64_t *__restrict__ actions, float *__restrict__ log_prob,
    float *__restrict__ values) {
  const int col = threadIdx.x & 31;
  const int half = threadIdx.x >> 5;  // 8 rows per 256-thread block
  const int b = blockIdx.x * 8 + half;
  const bool row_ok = b < B;
  const float x = row_ok ? out[static_cast<long long>(b) * kHeadLd + col] : 0.f;
  const bool is_logit = col < A;
  const float mx = half_max(is_logit ? x : -INFINITY);
  const float e = is_logit ? expf(x - mx) : 0.f;
  // sequential float32 running sum in column order (matches the CPU statement of the rule)
  float acc = 0.f, cdf = 0.f;
  for (int k = 0; k < A; ++k) {
    acc += __shfl(e, (threadIdx.x & 32) + k);
    if (k == col) cdf = acc;
  }
  const float u = uniforms ? (row_ok ? uniforms[b] : 0.f) : uniform01(seed, counter, b);
  const float thresh = u * acc;
  const unsigned long long below = __ballot(is_logit && cdf <= thresh);
  int a = __popcll((below >> (threadIdx.x & 32)) & 0xffffffffull);
  if (a > A - 1) a = A - 1;
  const float lse = mx + logf(acc);
  const float la = __shfl(x, (threadIdx.x & 32) + a) - lse;
  const float v = __shfl(x, (threadIdx.x & 32) + A);
  if (row_ok && col == 0) {
    actions[b] = a;
    log_prob[b] = la;
    values[b] = v;
  }
}

// Rollout tail in ONE launch, one wave per batch row: sums the split-K partial slabs of the hidden
// layer (adds the bias riding on slab 0; no ReLU after that layer, models.py:112), forms the A + 1
// head dot products (policy logits, value) and samples.  Replaces two GEMM launches (reduce +
// heads) and the act launch of the rollout step.  Cross-lane sums use DPP row operations and
// v_readlane (uniform lane indices), not LDS permutes: the kernel is a dependent chain.
struct HeadsActArgs {
  const float *hid_slabs;
  int nslab;
  long long slab_stride;
  const float *Wh, *bh;
  int B, A;
  const float *uniforms;
  uint64_t seed, counter;
  int64_t *actions;
  float *log_prob, *values;
  int b0;  // index of row 0 in the whole batch (sampling stream position of a batch slice)
};

__device__ __forceinline__ void heads_act_fused_block(const HeadsActArgs &p, int block) {
  const float *__restrict__ hid_slabs = p.hid_slabs;
  const int nslab = p.nslab;
  const long long slab_stride = p.slab_stride;
  const float *__restrict__ Wh = p.Wh;
  const float *__restrict__ bh = p.bh;
  const int B = p.B, A = p.A;
  const float *__restrict__ uniforms = p.uniforms;
  const uint64_t seed = p.seed, counter = p.counter;
  int64_t *__restrict__ actions = p.actions;
  float *__restrict__ log_prob = p.log_prob;
  float *__restrict__ values = p.values;
  const int lane = threadIdx.x & 63;
  const int b = block * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // whole wave exits together
  // the head weights of outputs 0..7, the bias and the uniform do not depend on the slabs:
  // issue them first, one memory round trip in all
  float4 wu0[8], ww0[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    wu0[jj] = *reinterpret_cast<const float4 *>(Wh + jj * 512 + lane * 8);
    ww0[jj] = *reinterpret_cast<const float4 *>(Wh + jj * 512 + lane * 8 + 4);
  }
  const int col = lane & 31;
  const float bias_col = bh[col];
  const float u_given = uniforms ? uniforms[b] : 0.f;
  float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float *row = hid_slabs + static_cast<long long>(b) * 512 + lane * 8;
  // slabs in batches of 7 with all loads issued before the first add (clamped index, masked add);
  // exactly 14 slabs (the weight-stationary linear layer of fc_rollout.hip) as ONE batch: one memory
  // round trip instead of two, summed in slab order like the loop below
  if (nslab == 14) {
    float4 u[14], w[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      u[i] = *reinterpret_cast<const float4 *>(row + i * slab_stride);
      w[i] = *reinterpret_cast<const float4 *>(row + i * slab_stride + 4);
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      h[0] += u[i].x; h[1] += u[i].y; h[2] += u[i].z; h[3] += u[i].w;
      h[4] += w[i].x; h[5] += w[i].y; h[6] += w[i].z; h[7] += w[i].w;
    }
  } else
  for (int z0 = 0; z0 < nslab; z0 += 7) {
    float4 u[7], w[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int z = min(z0 + i, nslab - 1);
      u[i] = *reinterpret_cast<const float4 *>(row + z * slab_stride);
      w[i] = *reinterpret_cast<const float4 *>(row + z * slab_stride + 4);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float k = (z0 + i) < nslab ? 1.f : 0.f;
      h[0] += k * u[i].x; h[1] += k * u[i].y; h[2] += k * u[i].z; h[3] += k * u[i].w;
      h[4] += k * w[i].x; h[5] += k * w[i].y; h[6] += k * w[i].z; h[7] += k * w[i].w;
    }
  }
  // x: lane j (and j + 32) ends up with head output j
  float x = 0.f;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float part = h[0] * wu0[jj].x + h[1] * wu0[jj].y + h[2] * wu0[jj].z + h[3] * wu0[jj].w +
                       h[4] * ww0[jj].x + h[5] * ww0[jj].y + h[6] * ww0[jj].z + h[7] * ww0[jj].w;
    const float tot = wave_sum_all(part);
    x = col == jj ? tot : x;
  }
  for (int j = 8; j <= A; ++j) {  // uniform; more than 7 actions
    const float4 wu = *reinterpret_cast<const float4 *>(Wh + j * 512 + lane * 8);
    const float4 ww = *reinterpret_cast<const float4 *>(Wh + j * 512 + lane * 8 + 4);
    const float part = h[0] * wu.x + h[1] * wu.y + h[2] * wu.z + h[3] * wu.w +
                       h[4] * ww.x + h[5] * ww.y + h[6] * ww.z + h[7] * ww.w;
    const float tot = wave_sum_all(part);
    x = col == j ? tot : x;
  }
  x += bias_col;
  // categorical head (see categorical_act_kernel); lane indices below are uniform -> v_readlane
  float mx = -INFINITY;
  for (int k = 0; k < A; ++k) mx = fmaxf(mx, lane_value(x, k));
  const bool is_logit = col < A;
  const float e = is_logit ? expf(x - mx) : 0.f;
  float acc = 0.f, cdf = 0.f;
  for (int k = 0; k < A; ++k) {  // sequential float32 running sum in column order
    acc += lane_value(e, k);
    if (k == col) cdf = acc;
  }
  const float u = uniforms ? u_given : uniform01(seed, counter, b + p.b0);
  const float thresh = u * acc;
  const unsigned long long below = __ballot(is_logit && cdf <= thresh);
  int a = __popcll(below & 0xffffffffull);
  if (a > A - 1) a = A - 1;
  const float lse = mx + logf(acc);
  const float la = lane_value(x, a) - lse;
  const float val = lane_value(x, A);
  if (lane == 0) {
    actions[b] = a;
    log_prob[b] = la;
    values[b] = val;
  }
}

__global__ __launch_bounds__(256) void heads_act_fused_kernel(const HeadsActArgs p) {
  heads_act_fused_block(p, blockIdx.x);
}

// The rollout step's last launch against the synthetic device env: workgroups [0, heads_blocks)
// finish the policy (slab sum + heads + sampling), the others generate the NEXT observation
// batch, rewards and resets -- the env of the measurement ignores the action, so the two halves
// are independent and the step loses one dependent launch (rollout).
__global__ __launch_bounds__(256) void heads_act_synth_kernel(const HeadsActArgs p, const dx::SynthArgs e,
                                                             int heads_blocks) {
  if (static_cast<int>(blockIdx.x) < heads_blocks) heads_act_fused_block(p, blockIdx.x);
  else dx::synth_atari_block(e, blockIdx.x - heads_blocks, gridDim.x - heads_blocks);
}

struct LossArgs {
  const float *out;        // [B][32] head outputs
  const int64_t *actions;  // [B]
  const float *old_log_prob, *advantages, *old_values, *value_targets;  // [B]
  float *dout;             // [B][32] gradient w.r.t. head outputs (padding columns zero)
  double *partials;        // [gridDim.x][8]
  int B, A;
  int mode;                // 0 = PPO, 1 = A2C
  float cliprange;         // < 0: no clipping (cliprange=None)
  float value_loss_coef, entropy_coef, inv_batch;
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void categorical_loss_kernel(const LossArgs a) {
  __shared__ double red[4][8];
  const int col = threadIdx.x & 31;
  const int half = threadIdx.x >> 5;
  const int b = blockIdx.x * 8 + half;
  const bool row_ok = b < a.B;
  const long long o = static_cast<long long>(b) * kHeadLd + col;
  const float x = row_ok ? a.out[o] : 0.f;
  const bool is_logit = col < a.A;
  const Softmax sm = half_softmax(x, is_logit);
  const float ent = -half_sum(is_logit ? sm.p * sm.logp : 0.f);
  const int act = row_ok ? static_cast<int>(a.actions[b]) : 0;
  const int hbase = threadIdx.x & 32;
  const float lp = __shfl(sm.logp, hbase + act);
  const float v = __shfl(x, hbase + a.A);
  const float adv = row_ok ? a.advantages[b] : 0.f;
  const float vt = row_ok ? a.value_targets[b] : 0.f;
  float pl, vl, dlp, dv;
  if (a.mode == 0) {
    const float old_lp = row_ok ? a.old_log_prob[b] : 0.f;
    const float old_v = row_ok ? a.old_values[b] : 0.f;
    const float ratio = expf(lp - old_lp);
    const float l1 = -ratio * adv;
    pl = l1;
    bool active = true;
    if (a.cliprange >= 0.f) {
      const float lo = 1.f - a.cliprange, hi = 1.f + a.cliprange;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float l2 = -rc * adv;
      pl = fmaxf(l1, l2);
      active = (l1 > l2) || (ratio >= lo && ratio <= hi);
    }
    dlp = active ? -adv * ratio * a.inv_batch : 0.f;
    const float d = v - vt;
    const float e1 = d * d;
    vl = e1;
    bool vactive = true;
    if (a.cliprange >= 0.f) {
      const float dvo = v - old_v;
      const float vc = old_v + fminf(fmaxf(dvo, -a.cliprange), a.cliprange);
      const float e2 = (vc - vt) * (vc - vt);
      vl = fmaxf(e1, e2);
      vactive = (e1 > e2) || (fabsf(dvo) <= a.cliprange);
    }
    dv = vactive ? a.value_loss_coef * 2.f * d * a.inv_batch : 0.f;
  } else {
    pl = -lp * adv;
    dlp = -adv * a.inv_batch;
    const float d = v - vt;
    vl = d * d;
    dv = a.value_loss_coef * 2.f * d * a.inv_batch;
  }
  // dL/dlogit_j = dlp*(1[j=a] - p_j) + (-c_H/B) * (-p_j*(logp_j + H))
  float g = 0.f;
  if (is_logit)
    g = dlp * ((col == act ? 1.f : 0.f) - sm.p) + a.entropy_coef * a.inv_batch * sm.p * (sm.logp + ent);
  else if (col == a.A)
    g = dv;
  if (row_ok) a.dout[o] = g;

  // block partial sums (float64): policy term, entropy, value term, adv, v, vt, (v-vt)^2, v^2
  const bool lead = row_ok && col == 0;
  double s[8] = {lead ? pl : 0.0, lead ? ent : 0.0, lead ? vl : 0.0, lead ? adv : 0.0,
                 lead ? v : 0.0,  lead ? vt : 0.0,  lead ? (double)(v - vt) * (v - vt) : 0.0,
                 lead ? (double)v * v : 0.0};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s[i] = wave_sum_d(s[i]);
    if (lane == 0) red[wave][i] = s[i];
  }
  __syncthreads();
  if (threadIdx.x < 8)
    a.partials[blockIdx.x * 8 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0]=loss, [1]=policy_loss, [2]=entropy, [3]=value_loss, [4]=mean adv, [5]=mean value,
// [6]=mean value target, [7]=r_squared (alg/common.py:9-12 with ppo.py:94's argument order)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const double *partials, int nblocks,
                                                          double count, float value_loss_coef,
                                                          float entropy_coef, float *out) {
  __shared__ double red[4][8];
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += partials[i * 8 + j];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] = wave_sum_d(s[j]);
    if (lane == 0) red[wave][j] = s[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[8];
    for (int j = 0; j < 8; ++j) t[j] = (red[0][j] + red[1][j] + red[2][j] + red[3][j]);
    const float policy = static_cast<float>(t[0] / count);
    const float ent = static_cast<float>(t[1] / count);
    const float value = static_cast<float>(t[2] / count);
    out[0] = (policy - entropy_coef * ent) + value_loss_coef * value;
    out[1] = policy;
    out[2] = ent;
    out[3] = value;
    out[4] = static_cast<float>(t[3] / count);
    out[5] = static_cast<float>(t[4] / count);
    out[6] = static_cast<float>(t[5] / count);
    const double mean_v = t[4] / count;
    const double var_v = count > 1 ? (t[7] - count * mean_v * mean_v) / (count - 1) : 0.0;  // torch .std(): unbiased
    out[7] = static_cast<float>(1.0 - (t[6] / count) / var_v);
  }
}

// ---- diagonal Gaussian head (thread per row; P <= 31 action dimensions) -----------------
// (normal01, normal_act_row, kHalfLog2Pi: heads_dev.hpp)

__global__ __launch_bounds__(256) void normal_act_kernel(
    const float *__restrict__ head, const float *__restrict__ logstd, int B, int P,
    const float *__restrict__ normals, uint64_t seed, uint64_t counter, float *__restrict__ actions,
    float *__restrict__ log_prob, float *__restrict__ values) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float *row = head + static_cast<long long>(b) * kHeadLd;
  log_prob[b] = normal_act_row(row, logstd, P, normals ? normals + static_cast<long long>(b) * P : nullptr, seed, counter, b,
                               actions + static_cast<long long>(b) * P);
  values[b] = row[P];
}

struct NormalLossArgs {
  const float *head, *logstd, *actions;  // (B,32), (P), (B,P)
  const float *old_log_prob, *advantages, *old_values, *value_targets;
  float *dhead;       // (B,32)
  double *partials;   // [gridDim.x][40]: 8 loss sums + 32 dlogstd sums
  int B, P, mode;
  float cliprange, value_loss_coef, entropy_coef, inv_batch;
};

__global__ __launch_bounds__(256) void normal_loss_kernel(const NormalLossArgs a) {
  __shared__ double red[4][40];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool row_ok = b < a.B;
  const long long ro = static_cast<long long>(row_ok ? b : 0) * kHeadLd;
  float lp = 0.f, ent = 0.f;
  for (int d = 0; d < a.P; ++d) {
    const float sigma = expf(a.logstd[d]);
    const float diff = a.actions[static_cast<long long>(row_ok ? b : 0) * a.P + d] - a.head[ro + d];
    lp += -(diff * diff) / (2.f * (sigma * sigma)) - logf(sigma) - kHalfLog2Pi;
    ent += 0.5f + kHalfLog2Pi + logf(sigma);
  }
  const float v = a.head[ro + a.P];
  const float adv = row_ok ? a.advantages[b] : 0.f;
  const float vt = row_ok ? a.value_targets[b] : 0.f;
  float pl, vl, dlp, dv;
  if (a.mode == 0) {
    const float old_lp = row_ok ? a.old_log_prob[b] : 0.f;
    const float old_v = row_ok ? a.old_values[b] : 0.f;
    const float ratio = expf(lp - old_lp);
    const float l1 = -ratio * adv;
    pl = l1;
    bool active = true;
    if (a.cliprange >= 0.f) {
      const float lo = 1.f - a.cliprange, hi = 1.f + a.cliprange;
      const float l2 = -fminf(fmaxf(ratio, lo), hi) * adv;
      pl = fmaxf(l1, l2);
      active = (l1 > l2) || (ratio >= lo && ratio <= hi);
    }
    dlp = active ? -adv * ratio * a.inv_batch : 0.f;
    const float dd = v - vt;
    const float e1 = dd * dd;
    vl = e1;
    bool vactive = true;
    if (a.cliprange >= 0.f) {
      const float dvo = v - old_v;
      const float vc = old_v + fminf(fmaxf(dvo, -a.cliprange), a.cliprange);
      const float e2 = (vc - vt) * (vc - vt);
      vl = fmaxf(e1, e2);
      vactive = (e1 > e2) || (fabsf(dvo) <= a.cliprange);
    }
    dv = vactive ? a.value_loss_coef * 2.f * dd * a.inv_batch : 0.f;
  } else {
    pl = -lp * adv;
    dlp = -adv * a.inv_batch;
    const float dd = v - vt;
    vl = dd * dd;
    dv = a.value_loss_coef * 2.f * dd * a.inv_batch;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // dL/dmean_d = dlp * (a - mu)/sigma^2 ; dL/dlogstd_d = sum_b dlp*((a-mu)^2/sigma^2 - 1) - c_H
  for (int d = 0; d < kHeadLd; ++d) {
    float g = 0.f;
    double dls = 0.0;
    if (d < a.P) {
      const float sigma = expf(a.logstd[d]);
      const float diff = a.actions[static_cast<long long>(row_ok ? b : 0) * a.P + d] - a.head[ro + d];
      const float var = sigma * sigma;
      g = dlp * diff / var;
      if (row_ok) dls = static_cast<double>(dlp) * (diff * diff / var - 1.f) - a.entropy_coef * a.inv_batch;
    } else if (d == a.P) {
      g = dv;
    }
    if (row_ok) a.dhead[ro + d] = g;
    if (d < a.P) {
      dls = wave_sum_d(dls);
      if (lane == 0) red[wave][8 + d] = dls;
    }
  }
  double s[8] = {row_ok ? pl : 0.0, row_ok ? ent : 0.0, row_ok ? vl : 0.0, row_ok ? adv : 0.0,
                 row_ok ? v : 0.0,  row_ok ? vt : 0.0,  row_ok ? (double)(v - vt) * (v - vt) : 0.0,
                 row_ok ? (double)v * v : 0.0};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s[i] = wave_sum_d(s[i]);
    if (lane == 0) red[wave][i] = s[i];
  }
  __syncthreads();
  if (threadIdx.x < 8 + a.P)
    a.partials[blockIdx.x * 40 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Heads + loss of a TRAINING minibatch in ONE launch (round 3): head = hid Wh^T + bh, the
// categorical PPO / A2C loss and its gradient w.r.t. the head outputs, dhid = dhead Wh (the heads'
// dgrad), this workgroup's partial of the heads' weight / bias gradient (slab, summed by the
// finalisation launch as before) and -- in the workgroup that finishes last -- the eight loss
// scalars.  Replaces six launches of the update (heads forward GEMM, loss, loss reduction, heads
// wgrad, heads dgrad, and with `stats` the advantage normalisation): the work is 5 MFLOP on 16.8 MB
// of hid / dhid at minibatch 8192, i.e. one pass over memory.
// One wave per row at a time, lane l owns k = 8 l .. 8 l + 7 of the 512-wide hidden row and (as
// lane & 31) head column `col`; the A + 1 <= 8 weight rows stay in registers; cross-lane sums by
// DPP (wave_sum_all), the per-row loss math is categorical_loss_kernel's, lane = column.
// derl/models.py:198-214 (output layers), derl/alg/ppo.py:24-108 / a2c.py:19-79, and the autograd
// backward of both (alg/common.py:70) for the heads.
// sum / max over each half (32 lanes) of the wave, returned in every lane of the half: four DPP steps
// inside the 16-lane rows and one exchange between the two rows of a half (the __shfl_xor ladder of
// half_sum is five LDS permutes -- ~100 cycles each in a dependent chain)
__device__ __forceinline__ float half_sum_dpp(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane holds its 16-lane row's total
  return v + __shfl_xor(v, 16);
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int moved = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, moved));
}
__device__ __forceinline__ float half_max_dpp(float v) {
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  v = dpp_max<0x141>(v);
  v = dpp_max<0x140>(v);
  return fmaxf(v, __shfl_xor(v, 16));
}
__device__ __forceinline__ Softmax half_softmax_dpp(float logit, bool is_logit) {
  const float mx = half_max_dpp(is_logit ? logit : -INFINITY);
  const float e = is_logit ? expf(logit - mx) : 0.f;
  const float s = half_sum_dpp(e);
  Softmax r;
  r.lse = mx + logf(s);
  r.logp = is_logit ? logit - r.lse : 0.f;
  r.p = e / s;
  return r;
}

// what the fused loss kernels know about the loss (members of their argument structs)
struct LossParams {
  int A, mode;
  bool normalize;
  float cliprange, value_loss_coef, entropy_coef, inv_batch;
};

// One row of the categorical PPO / A2C loss, lane = column of the 32-wide head row `x` (both halves
// of the wave hold the same row): the terms of categorical_loss_kernel and the gradient w.r.t. this
// lane's head output.  `act` and the per-row scalars are uniform over the wave.
struct CatRow { float g, adv, ent, pl, vl, v; };
__device__ __forceinline__ CatRow categorical_loss_row(const LossParams &a, float x, int col, bool is_logit, int act,
                                                       float adv, float vt, float old_lp, float old_v, float meanf,
                                                       float denom) {
  const int A = a.A;
  const Softmax sm = half_softmax_dpp(x, is_logit);
  const float ent = -half_sum_dpp(is_logit ? sm.p * sm.logp : 0.f);
  const float lp = lane_value(sm.logp, act);
  const float v = lane_value(x, A);
  if (a.normalize) adv = (adv - meanf) / denom;
  float pl, vl, dlp, dv;
  if (a.mode == 0) {
    const float ratio = expf(lp - old_lp);
    const float l1 = -ratio * adv;
    pl = l1;
    bool active = true;
    if (a.cliprange >= 0.f) {
      const float lo = 1.f - a.cliprange, hi = 1.f + a.cliprange;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float l2 = -rc * adv;
      pl = fmaxf(l1, l2);
      active = (l1 > l2) || (ratio >= lo && ratio <= hi);
    }
    dlp = active ? -adv * ratio * a.inv_batch : 0.f;
    const float d = v - vt;
    const float e1 = d * d;
    vl = e1;
    bool vactive = true;
    if (a.cliprange >= 0.f) {
      const float dvo = v - old_v;
      const float vc = old_v + fminf(fmaxf(dvo, -a.cliprange), a.cliprange);
      const float e2 = (vc - vt) * (vc - vt);
      vl = fmaxf(e1, e2);
      vactive = (e1 > e2) || (fabsf(dvo) <= a.cliprange);
    }
    dv = vactive ? a.value_loss_coef * 2.f * d * a.inv_batch : 0.f;
  } else {
    pl = -lp * adv;
    dlp = -adv * a.inv_batch;
    const float d = v - vt;
    vl = d * d;
    dv = a.value_loss_coef * 2.f * d * a.inv_batch;
  }
  float gu = 0.f;  // dL/dhead[col]
  if (is_logit)
    gu = dlp * ((col == act ? 1.f : 0.f) - sm.p) + a.entropy_coef * a.inv_batch * sm.p * (sm.logp + ent);
  else if (col == A)
    gu = dv;
  return CatRow{gu, adv, ent, pl, vl, v};
}

// The end of a fused loss launch: every workgroup's float64 partial sums of the eight loss terms are
// written through, and the LAST workgroup (atomic ticket) reduces them in workgroup order to the
// eight loss scalars of loss_reduce_kernel and resets the ticket.  lsum: [waves][8] doubles in LDS,
// already holding each wave's sums; NW = waves per workgroup.
template <int NW>
__device__ __forceinline__ void loss_finish(double *lsum, unsigned *ticket_lds, double *partials, unsigned *counter,
                                            float *loss_out, int B, float value_loss_coef, float entropy_coef) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 8) {  // write-through: the last workgroup reads every workgroup's partials
    double tot = lsum[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += lsum[w * 8 + threadIdx.x];
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(partials + blockIdx.x * 8 + threadIdx.x),
                       __builtin_bit_cast(unsigned long long, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    *ticket_lds = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*ticket_lds != gridDim.x - 1) return;
  // ---- last workgroup: loss_reduce_kernel over the partials (sc1 loads) ----
  double t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += 64 * NW)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      t8[j] += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(partials + i * 8 + j),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    t8[j] = wave_sum_d(t8[j]);
    if (lane == 0) lsum[wave * 8 + j] = t8[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[8];
    for (int j = 0; j < 8; ++j) {
      t[j] = lsum[j];
      for (int w = 1; w < NW; ++w) t[j] += lsum[w * 8 + j];
    }
    const double count = B;
    const float policy = static_cast<float>(t[0] / count), ent = static_cast<float>(t[1] / count);
    const float value = static_cast<float>(t[2] / count);
    float *out = loss_out;
    out[0] = (policy - entropy_coef * ent) + value_loss_coef * value;
    out[1] = policy; out[2] = ent; out[3] = value;
    out[4] = static_cast<float>(t[3] / count);
    out[5] = static_cast<float>(t[4] / count);
    out[6] = static_cast<float>(t[5] / count);
    const double mean_v = t[4] / count;
    const double var_v = count > 1 ? (t[7] - count * mean_v * mean_v) / (count - 1) : 0.0;
    out[7] = static_cast<float>(1.0 - (t[6] / count) / var_v);
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

struct HeadsLossArgs {
  const float *hid;        // [B][512]
  const float *Wh, *bh;    // packed heads [32][512] (rows 0..A-1 policy, row A value), bias [32]
  const int64_t *actions;
  const float *old_log_prob, *advantages, *old_values, *value_targets;
  const double *stats;     // optional {sum, sumsq, n} of the raw advantages: normalise here
  float norm_eps;
  float *adv_norm_out;     // optional: the normalised advantages (with stats)
  float *head, *dhead;     // [B][32]
  float *dhid;             // [B][512]
  float *slab, *bias_slab; // [gridDim.x][32][512], [gridDim.x][32]
  double *partials;        // [gridDim.x][8]
  unsigned *counter;       // zero before the first launch; the last workgroup resets it
  float *loss_out;         // [8]
  int B, A, rows_per_wg, mode;
  float cliprange, value_loss_coef, entropy_coef, inv_batch;
};

constexpr int kHlWaves = 8;  // waves per workgroup of heads_loss_fused_kernel

__global__ __launch_bounds__(64 * kHlWaves) void heads_loss_fused_kernel(const HeadsLossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hl_smem[];
  float *wsum = hl_smem;                                  // [waves][8][512] weight-gradient partials of the waves
  float *bsum = wsum + kHlWaves * 8 * 512;                // [waves][32]
  double *lsum = reinterpret_cast<double *>(bsum + kHlWaves * 32);  // [waves][8]
  __shared__ unsigned ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31;
  const int A = a.A;
  const LossParams lparams{a.A, a.mode, a.stats != nullptr, a.cliprange, a.value_loss_coef, a.entropy_coef, a.inv_batch};
  float4 wu[8], ww[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    wu[j] = *reinterpret_cast<const float4 *>(a.Wh + j * 512 + lane * 8);
    ww[j] = *reinterpret_cast<const float4 *>(a.Wh + j * 512 + lane * 8 + 4);
  }
  const float bias_col = a.bh[col];
  float meanf = 0.f, denom = 1.f;
  if (a.stats) {  // adv_apply_kernel's expression
    const double cnt = a.stats[2], mean = a.stats[0] / cnt;
    double var = a.stats[1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    meanf = static_cast<float>(mean);
    denom = static_cast<float>(sqrt(var)) + a.norm_eps;
  }
  float acc[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
  float bacc = 0.f;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool is_logit = col < A;
  const int row_begin = blockIdx.x * a.rows_per_wg;
  const int row_end = min(a.B, row_begin + a.rows_per_wg);
  constexpr int kAhead = 4;  // rows in flight per wave (their hidden rows AND their per-row scalars)
  for (int r0 = row_begin + wave; r0 < row_end; r0 += kHlWaves * kAhead) {
    float4 hu[kAhead], hw[kAhead];
    float p_adv[kAhead], p_vt[kAhead], p_olp[kAhead], p_ov[kAhead];
    int p_act[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int row = min(r0 + kHlWaves * u, a.B - 1);
      hu[u] = *reinterpret_cast<const float4 *>(a.hid + static_cast<long long>(row) * 512 + lane * 8);
      hw[u] = *reinterpret_cast<const float4 *>(a.hid + static_cast<long long>(row) * 512 + lane * 8 + 4);
      p_act[u] = static_cast<int>(a.actions[row]);
      p_adv[u] = a.advantages[row];
      p_vt[u] = a.value_targets[row];
      p_olp[u] = a.mode == 0 ? a.old_log_prob[row] : 0.f;
      p_ov[u] = a.mode == 0 ? a.old_values[row] : 0.f;
    }
    // The rows of a batch go through every phase TOGETHER (arrays over u, no early exit): the phases
    // are chains of DPP reductions, v_readlane and transcendentals with long dependent latencies, and
    // four independent chains overlap where one row at a time took 1.5-3 us (39 us for 16 rows per wave).
    bool ok[kAhead];
    float x[kAhead], g[kAhead], advn[kAhead], entv[kAhead], plv[kAhead], vlv[kAhead], vv[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      ok[u] = r0 + kHlWaves * u < row_end;  // uniform
      x[u] = 0.f;  // lane j (and j + 32) ends up with head output j
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const float part = hu[u].x * wu[j].x + hu[u].y * wu[j].y + hu[u].z * wu[j].z + hu[u].w * wu[j].w +
                           hw[u].x * ww[j].x + hw[u].y * ww[j].y + hw[u].z * ww[j].z + hw[u].w * ww[j].w;
        const float tot = wave_sum_all(part);
        x[u] = col == j ? tot : x[u];
      }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      x[u] += bias_col;
      if (col > A) x[u] = 0.f;  // padding columns
      const int act = __builtin_amdgcn_readfirstlane(p_act[u]);  // uniform: one row per wave
      const CatRow cr = categorical_loss_row(lparams, x[u], col, is_logit, act, p_adv[u], p_vt[u], p_olp[u], p_ov[u], meanf, denom);
      const float gu = cr.g, adv = cr.adv, ent = cr.ent, pl = cr.pl, vl = cr.vl, v = cr.v;
      g[u] = ok[u] ? gu : 0.f;  // rows past the slice contribute nothing
      advn[u] = adv; entv[u] = ent; plv[u] = pl; vlv[u] = vl; vv[u] = v;
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      if (!ok[u]) continue;  // uniform; only stores and the scalar sums below
      const int b = r0 + kHlWaves * u;
      if (a.stats && a.adv_norm_out && lane == 0) a.adv_norm_out[b] = advn[u];
      if (lane < 32) {
        a.head[static_cast<long long>(b) * kHeadLd + col] = x[u];
        a.dhead[static_cast<long long>(b) * kHeadLd + col] = g[u];
        bacc += g[u];
      }
      // policy term, entropy, value term, adv, v, vt, (v - vt)^2, v^2 (uniform over the wave)
      const float vt = p_vt[u];
      s[0] += plv[u]; s[1] += entv[u]; s[2] += vlv[u]; s[3] += advn[u]; s[4] += vv[u]; s[5] += vt;
      s[6] += static_cast<double>(vv[u] - vt) * (vv[u] - vt); s[7] += static_cast<double>(vv[u]) * vv[u];
    }
    // ---- heads dgrad (this lane's 8 k) and weight-gradient partial ----
    float d8[kAhead][8];
#pragma unroll
    for (int u = 0; u < kAhead; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) d8[u][i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const float gj = lane_value(g[u], j);  // uniform
        const float h[8] = {hu[u].x, hu[u].y, hu[u].z, hu[u].w, hw[u].x, hw[u].y, hw[u].z, hw[u].w};
        d8[u][0] = fmaf(gj, wu[j].x, d8[u][0]); d8[u][1] = fmaf(gj, wu[j].y, d8[u][1]);
        d8[u][2] = fmaf(gj, wu[j].z, d8[u][2]); d8[u][3] = fmaf(gj, wu[j].w, d8[u][3]);
        d8[u][4] = fmaf(gj, ww[j].x, d8[u][4]); d8[u][5] = fmaf(gj, ww[j].y, d8[u][5]);
        d8[u][6] = fmaf(gj, ww[j].z, d8[u][6]); d8[u][7] = fmaf(gj, ww[j].w, d8[u][7]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(gj, h[i], acc[j][i]);
      }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      if (!ok[u]) continue;
      float *drow = a.dhid + static_cast<long long>(r0 + kHlWaves * u) * 512 + lane * 8;
      *reinterpret_cast<float4 *>(drow) = make_float4(d8[u][0], d8[u][1], d8[u][2], d8[u][3]);
      *reinterpret_cast<float4 *>(drow + 4) = make_float4(d8[u][4], d8[u][5], d8[u][6], d8[u][7]);
    }
  }
  // ---- the waves' partials meet in LDS, summed in wave order ----
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float *dst = wsum + (wave * 8 + j) * 512 + lane * 8;
    *reinterpret_cast<float4 *>(dst) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
  }
  if (lane < 32) bsum[wave * 32 + col] = bacc;
  if (lane < 8) {
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) mine = lane == i ? s[i] : mine;
    lsum[wave * 8 + lane] = mine;
  }
  __syncthreads();
  float *slab = a.slab + static_cast<long long>(blockIdx.x) * kHeadLd * 512;
  for (int i = threadIdx.x; i < 8 * 512 / 4; i += 64 * kHlWaves) {  // rows 0..7 (A + 1 <= 8); the rest is never read
    float4 t = reinterpret_cast<const float4 *>(wsum)[i];
#pragma unroll
    for (int w = 1; w < kHlWaves; ++w) {
      const float4 p = reinterpret_cast<const float4 *>(wsum + w * 4096)[i];
      t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
    }
    reinterpret_cast<float4 *>(slab)[i] = t;
  }
  if (threadIdx.x < 32) {
    float t = bsum[threadIdx.x];
#pragma unroll
    for (int w = 1; w < kHlWaves; ++w) t += bsum[w * 32 + threadIdx.x];
    a.bias_slab[static_cast<long long>(blockIdx.x) * kHeadLd + threadIdx.x] = t;
  }
  loss_finish<kHlWaves>(lsum, &ticket, a.partials, a.counter, a.loss_out, a.B, a.value_loss_coef, a.entropy_coef);
}

// ---- the FACTORED tail of the Nature CNN (csrc/tail.hip has the rest) ----
// derl's linear layer is followed by NO activation (derl/models.py:112-115, 198-203): the 3136 -> 512
// layer and the two heads are ONE affine map of the flattened conv output, out = y2 Wc^T + beff with
// Wc = Wh Wfc (A + 1 rows of 3136).  The kernels below read y2 where their predecessors read the
// 512-wide hidden row: the same loss, the same sampling rule, 1/100 of the multiplies.

constexpr int kTailK = 3136;   // floats per flattened conv output (7 x 7 x 64, NHWC order)
constexpr int kTailQ = 12;     // float4 per lane of a row held by one wave (12 x 64 x 4 = 3072) + one float
constexpr int kTlWaves = 8;
constexpr int kTailLdsRows = 12;  // rows of Wc a workgroup keeps in LDS (150 KB); with 13 .. 19 outputs (more than 11 actions)
                                  // the rows beyond come straight from L2 -- every workgroup reads the same 88 KB

struct TailLossArgs {
  const float *y2;         // [B][3136]
  const float *Wc, *beff;  // [Jp][3136] (rows 0..A-1 policy, row A value, the rest zero), [Jp]: Jp = 8 ceil((A + 1) / 8)
  const int64_t *actions;
  const float *old_log_prob, *advantages, *old_values, *value_targets;
  const double *stats;     // optional {sum, sumsq, n} of the raw advantages: normalise here
  float norm_eps;
  float *adv_norm_out;
  float *head, *dhead;     // [B][32]
  double *partials;        // [gridDim.x][8]
  unsigned *counter;
  float *loss_out;         // [8]
  int B, A, rows_per_wg, mode;
  float cliprange, value_loss_coef, entropy_coef, inv_batch;
};

// the A + 1 dot products of NR rows held in registers (y: 12 float4 + 1 float per lane and row) with
// the rows of Wc in LDS: lane j (and j + 32) of x[u] ends up with output j of row u
template <int NR>
__device__ __forceinline__ void tail_dots(const float *wc_lds, const float *wc_global, int NJ, const float4 (&y)[NR][kTailQ],
                                          const float (&yt)[NR], int lane, int col, float (&x)[NR]) {
  for (int j = 0; j < NJ; ++j) {  // uniform
    const float *w = j < kTailLdsRows ? wc_lds + j * kTailK : wc_global + j * kTailK;
    float part[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) part[u] = 0.f;
#pragma unroll
    for (int q = 0; q < kTailQ; ++q) {
      const float4 w4 = reinterpret_cast<const float4 *>(w)[lane + 64 * q];
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        part[u] = fmaf(y[u][q].x, w4.x, part[u]);
        part[u] = fmaf(y[u][q].y, w4.y, part[u]);
        part[u] = fmaf(y[u][q].z, w4.z, part[u]);
        part[u] = fmaf(y[u][q].w, w4.w, part[u]);
      }
    }
    const float wt = w[64 * 4 * kTailQ + lane];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const float tot = wave_sum_all(fmaf(yt[u], wt, part[u]));
      x[u] = col == j ? tot : x[u];
    }
  }
}

// forward of the factored tail + categorical PPO / A2C loss + its gradient w.r.t. the A + 1 outputs +
// the loss scalars, ONE launch: what the linear layer's forward and heads_loss_fused_kernel did
__global__ __launch_bounds__(64 * kTlWaves) void tail_loss_kernel(const TailLossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tl_smem[];  // Wc rows, then [waves][8] doubles
  __shared__ unsigned ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31;
  const int A = a.A, NJ = A + 1, lds_rows = NJ < kTailLdsRows ? NJ : kTailLdsRows;
  double *lsum = reinterpret_cast<double *>(tl_smem + (lds_rows < 8 ? 8 : lds_rows) * kTailK);
  const LossParams lparams{a.A, a.mode, a.stats != nullptr, a.cliprange, a.value_loss_coef, a.entropy_coef, a.inv_batch};
  const int row_begin = blockIdx.x * a.rows_per_wg;
  const int row_end = min(a.B, row_begin + a.rows_per_wg);
  constexpr int kAhead = 2;  // rows per wave and pass: every Wc fragment read from LDS serves both
  // A pass's rows are loaded one pass AHEAD -- the first pass's before Wc is staged, so that the two streams of loads
  // overlap (at minibatch 1,024 a workgroup has ONE pass: its y2 loads used to start behind the staging barrier)
  float4 y[kAhead][kTailQ];
  float yt[kAhead], p_adv[kAhead], p_vt[kAhead], p_olp[kAhead], p_ov[kAhead];
  int p_act[kAhead];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int row = min(r0 + kTlWaves * u, a.B - 1);
      const float *base = a.y2 + static_cast<long long>(row) * kTailK;
#pragma unroll
      for (int q = 0; q < kTailQ; ++q) y[u][q] = reinterpret_cast<const float4 *>(base)[lane + 64 * q];
      yt[u] = base[64 * 4 * kTailQ + lane];
      p_act[u] = static_cast<int>(a.actions[row]);
      p_adv[u] = a.advantages[row];
      p_vt[u] = a.value_targets[row];
      p_olp[u] = a.mode == 0 ? a.old_log_prob[row] : 0.f;
      p_ov[u] = a.mode == 0 ? a.old_values[row] : 0.f;
    }
  };
  if (row_begin + wave < row_end) fetch(row_begin + wave);
  for (int i = threadIdx.x; i < lds_rows * (kTailK / 4); i += 64 * kTlWaves)
    reinterpret_cast<float4 *>(tl_smem)[i] = reinterpret_cast<const float4 *>(a.Wc)[i];
  const float bias_col = col <= A ? a.beff[col] : 0.f;
  float meanf = 0.f, denom = 1.f;
  if (a.stats) {  // adv_apply_kernel's expression
    const double cnt = a.stats[2], mean = a.stats[0] / cnt;
    double var = a.stats[1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    meanf = static_cast<float>(mean);
    denom = static_cast<float>(sqrt(var)) + a.norm_eps;
  }
  __syncthreads();
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool is_logit = col < A;
  for (int r0 = row_begin + wave; r0 < row_end; r0 += kTlWaves * kAhead) {
    float x[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) x[u] = 0.f;
    tail_dots<kAhead>(tl_smem, a.Wc, NJ, y, yt, lane, col, x);
    CatRow cr[kAhead];
    float vts[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      x[u] += bias_col;
      if (col > A) x[u] = 0.f;  // padding columns
      const int act = __builtin_amdgcn_readfirstlane(p_act[u]);
      cr[u] = categorical_loss_row(lparams, x[u], col, is_logit, act, p_adv[u], p_vt[u], p_olp[u], p_ov[u], meanf, denom);
      vts[u] = p_vt[u];
    }
    if (r0 + kTlWaves * kAhead < row_end) fetch(r0 + kTlWaves * kAhead);  // the next pass's rows, under this pass's stores
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const bool ok = r0 + kTlWaves * u < row_end;  // uniform
      if (!ok) continue;  // rows past the slice: nothing stored, nothing summed
      const int b = r0 + kTlWaves * u;
      if (a.stats && a.adv_norm_out && lane == 0) a.adv_norm_out[b] = cr[u].adv;
      if (lane < 32) {
        a.head[static_cast<long long>(b) * kHeadLd + col] = x[u];
        a.dhead[static_cast<long long>(b) * kHeadLd + col] = cr[u].g;
      }
      const float vt = vts[u];
      s[0] += cr[u].pl; s[1] += cr[u].ent; s[2] += cr[u].vl; s[3] += cr[u].adv; s[4] += cr[u].v; s[5] += vt;
      s[6] += static_cast<double>(cr[u].v - vt) * (cr[u].v - vt); s[7] += static_cast<double>(cr[u].v) * cr[u].v;
    }
  }
  if (lane < 8) {
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) mine = lane == i ? s[i] : mine;
    lsum[wave * 8 + lane] = mine;
  }
  __syncthreads();
  loss_finish<kTlWaves>(lsum, &ticket, a.partials, a.counter, a.loss_out, a.B, a.value_loss_coef, a.entropy_coef);
}

// tail_loss_kernel AND tail.hip's tail_bwd_kernel in ONE pass over y2 (up to 7 actions), round 5: a row's outputs, its
// loss and dL/dout, then -- from the same registers -- dy2 = relu'(y2) * (dout Wc) and the workgroup's partial
// G = dout^T y2, s = sum dout.  (Two launches read y2 twice -- 206 MB at minibatch 8192 -- and at minibatch 1,024 each
// was a latency chain of its own: 17 + 9 us.)  Threads and registers are tail_bwd_kernel's: 448 threads, thread t owns
// columns t + 448 i (i < 7) with their Wc and G entries; the outputs of four rows at a time meet through LDS (seven
// wave sums per output, added in wave order), waves 0-3 take one row each through categorical_loss_row, and every
// thread reads the four rows' dL/dout back from LDS.
constexpr int kFbThreads = 448, kFbCols = kTailK / kFbThreads, kFbRows = 4, kFbWaves = kFbThreads / 64;
static_assert(kFbThreads * kFbCols == kTailK, "seven columns per thread");

struct TailBwdOut {
  float *dy2;    // [B][3136]
  float *gslab;  // [gridDim.x][Jp][3136] partial G
  float *sslab;  // [gridDim.x][Jp] partial s
  int Jp;
};

using f32x2 = __attribute__((ext_vector_type(2))) float;
// a (kept by the wave's lanes 0 .. 31) and b (kept by lanes 32 .. 63), each summed over the lane pairs (l, l + 32).
// Inline asm on purpose: with a wave-uniform operand hipcc 7.2 folds __builtin_amdgcn_permlane32_swap's two results into
// one register (v_permlane32_swap v3, v1; v_add v1, v3, v3 -- measured wrong sums, gpurun_out/tmp/meet_probe.hip).
__device__ __forceinline__ float swap_sum32(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = [a.lo | b.lo], b = [a.hi | b.hi]
  return a + b;
}
// a (kept by the even 16-lane rows) and b (kept by the odd rows), each summed over the row pairs (0, 1) and (2, 3)
__device__ __forceinline__ float swap_sum16(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // a = rows [a0 b0 a2 b2], b = rows [a1 b1 a3 b3]
  return a + b;
}
// the sum over each 16-lane row, in every lane of the row
__device__ __forceinline__ float row_sum_all(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  return dpp_add<0x140, 0xf>(v);  // row_mirror
}

template <int NJ>
__global__ __launch_bounds__(kFbThreads) void tail_loss_bwd_kernel(const TailLossArgs a, const TailBwdOut o) {
  __shared__ float red[kFbWaves][kFbRows * 8];
  __shared__ float dsh[kFbRows][8];
  __shared__ double lsum[kFbWaves * 8];
  __shared__ unsigned ticket;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 31;
  const int A = a.A;
  const LossParams lparams{a.A, a.mode, a.stats != nullptr, a.cliprange, a.value_loss_coef, a.entropy_coef, a.inv_batch};
  // Up to six outputs: in PAIRS (2 jp, 2 jp + 1) -- one v_pk_fma_f32 multiplies a y2 element by both rows' weights (and, in
  // the backward half, adds both rows' dL/dout x y2 to G): the fp32 vector peak needs the packed form; every element's own
  // fma chain is the one the unpacked code forms.  An odd NJ carries a zero row.  Seven and eight outputs stay unpacked
  // (the aligned register pairs spill there).
  constexpr bool PACK = NJ <= 6;
  constexpr int NP = (NJ + 1) / 2, NR = PACK ? NP : NJ;  // register rows of wc / g
  using Row = std::conditional_t<PACK, f32x2, float>;
  Row wc[NR][kFbCols], g[NR][kFbCols];
  float sacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sacc[j] = 0.f;
#pragma unroll
  for (int jr = 0; jr < NR; ++jr)
#pragma unroll
    for (int i = 0; i < kFbCols; ++i) {
      if constexpr (PACK) {
        wc[jr][i].x = a.Wc[(2 * jr) * kTailK + t + kFbThreads * i];
        wc[jr][i].y = 2 * jr + 1 < NJ ? a.Wc[(2 * jr + 1) * kTailK + t + kFbThreads * i] : 0.f;
        g[jr][i] = f32x2{0.f, 0.f};
      } else {
        wc[jr][i] = a.Wc[jr * kTailK + t + kFbThreads * i];
        g[jr][i] = 0.f;
      }
    }
  auto w_of = [&](int j, int i) -> float {
    if constexpr (PACK) return j & 1 ? wc[j >> 1][i].y : wc[j >> 1][i].x;
    else return wc[j][i];
  };
  const float bias_col = col <= A ? a.beff[col] : 0.f;
  float meanf = 0.f, denom = 1.f;
  if (a.stats) {  // adv_apply_kernel's expression
    const double cnt = a.stats[2], mean = a.stats[0] / cnt;
    double var = a.stats[1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    meanf = static_cast<float>(mean);
    denom = static_cast<float>(sqrt(var)) + a.norm_eps;
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool is_logit = col < A;
  const int r0 = blockIdx.x * a.rows_per_wg, r1 = min(a.B, r0 + a.rows_per_wg);
  // a pass's four rows (and this wave's row scalars, waves 0-3) are loaded one pass AHEAD, behind the previous pass's dots
  float y[kFbRows][kFbCols], yn[kFbRows][kFbCols];
  int p_act = 0, n_act = 0;
  float p_adv = 0.f, p_vt = 0.f, p_olp = 0.f, p_ov = 0.f, n_adv = 0.f, n_vt = 0.f, n_olp = 0.f, n_ov = 0.f;
  auto fetch = [&](int r) {
#pragma unroll
    for (int u = 0; u < kFbRows; ++u) {
      const long long row = min(r + u, a.B - 1);
#pragma unroll
      for (int i = 0; i < kFbCols; ++i) yn[u][i] = a.y2[row * kTailK + t + kFbThreads * i];
    }
    const int myrow = min(r + (wave < kFbRows ? wave : 0), a.B - 1);
    n_act = static_cast<int>(a.actions[myrow]);
    n_adv = a.advantages[myrow]; n_vt = a.value_targets[myrow];
    n_olp = a.mode == 0 ? a.old_log_prob[myrow] : 0.f; n_ov = a.mode == 0 ? a.old_values[myrow] : 0.f;
  };
  if (r0 < r1) fetch(r0);
  for (int r = r0; r < r1; r += kFbRows) {
#pragma unroll
    for (int u = 0; u < kFbRows; ++u)
#pragma unroll
      for (int i = 0; i < kFbCols; ++i) y[u][i] = yn[u][i];
    p_act = n_act; p_adv = n_adv; p_vt = n_vt; p_olp = n_olp; p_ov = n_ov;
    if (r + kFbRows < r1) fetch(r + kFbRows);
    // ---- the four rows' outputs: per-thread partial dots, wave sums, seven waves through LDS ----
    // The 4 x NJ partial dots of a lane, summed over the wave TRANSPOSED: rows (u, u + 2) meet across the wave's halves
    // (v_permlane32_swap: one swap + one add per pair), rows (0, 1) / (2, 3) across the 16-lane rows (v_permlane16_swap), then
    // four DPP steps inside a row -- row u of the wave ends with output j's total in every lane: 10 NJ instructions where 4 NJ
    // whole-wave sums took 28 NJ.
    float mine = 0.f;
    auto meet = [&](int j, float p0, float p1, float p2, float p3) {  // output j's four row sums -> row u of the wave
      const float tot = row_sum_all(swap_sum16(swap_sum32(p0, p2), swap_sum32(p1, p3)));
      mine = (lane & 15) == j ? tot : mine;
    };
    if constexpr (PACK) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        f32x2 part[kFbRows];
#pragma unroll
        for (int u = 0; u < kFbRows; ++u) {
          f32x2 acc2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < kFbCols; ++i) acc2 = __builtin_elementwise_fma(f32x2{y[u][i], y[u][i]}, wc[jp][i], acc2);
          part[u] = acc2;
        }
        meet(2 * jp, part[0].x, part[1].x, part[2].x, part[3].x);
        if (2 * jp + 1 < NJ) meet(2 * jp + 1, part[0].y, part[1].y, part[2].y, part[3].y);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float part[kFbRows];
#pragma unroll
        for (int u = 0; u < kFbRows; ++u) {
          part[u] = 0.f;
#pragma unroll
          for (int i = 0; i < kFbCols; ++i) part[u] = fmaf(y[u][i], wc[j][i], part[u]);
        }
        meet(j, part[0], part[1], part[2], part[3]);
      }
    }
    if ((lane & 15) < 8) red[wave][(lane >> 4) * 8 + (lane & 15)] = mine;
    __syncthreads();
    if (wave < kFbRows) {  // one row per wave, lane = column (both halves hold the same row)
      const int u = wave;
      float x = 0.f;
      if (col < NJ) {
#pragma unroll
        for (int w = 0; w < kFbWaves; ++w) x += red[w][u * 8 + col];
        x += bias_col;
      }
      const int act = __builtin_amdgcn_readfirstlane(p_act);
      const CatRow cr = categorical_loss_row(lparams, x, col, is_logit, act, p_adv, p_vt, p_olp, p_ov, meanf, denom);
      const bool ok = r + u < r1;  // uniform
      if (lane < 8) dsh[u][lane] = ok ? cr.g : 0.f;
      if (ok) {
        const int b = r + u;
        if (a.stats && a.adv_norm_out && lane == 0) a.adv_norm_out[b] = cr.adv;
        if (lane < 32) {
          a.head[static_cast<long long>(b) * kHeadLd + col] = x;
          a.dhead[static_cast<long long>(b) * kHeadLd + col] = cr.g;
        }
        s[0] += cr.pl; s[1] += cr.ent; s[2] += cr.vl; s[3] += cr.adv; s[4] += cr.v; s[5] += p_vt;
        s[6] += static_cast<double>(cr.v - p_vt) * (cr.v - p_vt); s[7] += static_cast<double>(cr.v) * cr.v;
      }
    }
    __syncthreads();
    // ---- dy2 and the partial G / s from the same rows (tail_bwd_kernel's arithmetic) ----
#pragma unroll
    for (int u = 0; u < kFbRows; ++u) {
      if (r + u >= r1) break;  // uniform
      float d[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) d[j] = dsh[u][j];
      float *out = o.dy2 + static_cast<long long>(r + u) * kTailK + t;
#pragma unroll
      for (int i = 0; i < kFbCols; ++i) {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) dsum = fmaf(d[j], w_of(j, i), dsum);
        if constexpr (PACK) {
#pragma unroll
          for (int jp = 0; jp < NP; ++jp)
            g[jp][i] = __builtin_elementwise_fma(f32x2{d[2 * jp], 2 * jp + 1 < NJ ? d[2 * jp + 1] : 0.f}, f32x2{y[u][i], y[u][i]}, g[jp][i]);
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) g[j][i] = fmaf(d[j], y[u][i], g[j][i]);
        }
        out[kFbThreads * i] = y[u][i] > 0.f ? dsum : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) sacc[j] += d[j];
    }
    // (the next pass's first barrier orders its writes of red / dsh behind these reads: every thread passes it
    // only after reading dsh here)
  }
  float *slab = o.gslab + static_cast<long long>(blockIdx.x) * o.Jp * kTailK;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < kFbCols; ++i) {
        if constexpr (PACK) slab[j * kTailK + t + kFbThreads * i] = j & 1 ? g[j >> 1][i].y : g[j >> 1][i].x;
        else slab[j * kTailK + t + kFbThreads * i] = g[j][i];
      }
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) o.sslab[blockIdx.x * o.Jp + j] = sacc[j];
  }
  if (lane < 8) {
    double mine_d = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) mine_d = lane == i ? s[i] : mine_d;
    lsum[wave * 8 + lane] = wave < kFbRows ? mine_d : 0.0;
  }
  __syncthreads();
  loss_finish<kFbWaves>(lsum, &ticket, a.partials, a.counter, a.loss_out, a.B, a.value_loss_coef, a.entropy_coef);
}

// rollout: out = y2 Wc^T + beff and the categorical sampling of heads_act_fused_block, one wave per row
struct TailActArgs {
  const float *y2;
  const float *Wc, *beff;
  int B, A;
  const float *uniforms;
  uint64_t seed, counter;
  int64_t *actions;
  float *log_prob, *values;
  int b0;  // index of row 0 in the whole batch (sampling stream position of a batch slice)
};

__device__ __forceinline__ void tail_act_block(const TailActArgs &p, int block, float *wc_lds) {
  const int lane = threadIdx.x & 63, col = lane & 31;
  const int A = p.A, NJ = A + 1;
  const int b = block * 4 + (threadIdx.x >> 6);
  const int row = min(b, p.B - 1);
  float4 y[1][kTailQ];
  float yt[1];
  const float *base = p.y2 + static_cast<long long>(row) * kTailK;
#pragma unroll
  for (int q = 0; q < kTailQ; ++q) y[0][q] = reinterpret_cast<const float4 *>(base)[lane + 64 * q];
  yt[0] = base[64 * 4 * kTailQ + lane];
  const float bias_col = col <= A ? p.beff[col] : 0.f;
  const float u_given = p.uniforms ? p.uniforms[row] : 0.f;
  const int lds_rows = NJ < kTailLdsRows ? NJ : kTailLdsRows;
  for (int i = threadIdx.x; i < lds_rows * (kTailK / 4); i += 256)
    reinterpret_cast<float4 *>(wc_lds)[i] = reinterpret_cast<const float4 *>(p.Wc)[i];
  __syncthreads();
  if (b >= p.B) return;  // whole wave exits together (after the barrier)
  float xs[1] = {0.f};
  tail_dots<1>(wc_lds, p.Wc, NJ, y, yt, lane, col, xs);
  const float x = xs[0] + bias_col;
  // categorical head (see heads_act_fused_block); lane indices below are uniform -> v_readlane
  float mx = -INFINITY;
  for (int k = 0; k < A; ++k) mx = fmaxf(mx, lane_value(x, k));
  const bool is_logit = col < A;
  const float e = is_logit ? expf(x - mx) : 0.f;
  float acc = 0.f, cdf = 0.f;
  for (int k = 0; k < A; ++k) {  // sequential float32 running sum in column order
    acc += lane_value(e, k);
    if (k == col) cdf = acc;
  }
  const float u = p.uniforms ? u_given : uniform01(p.seed, p.counter, b + p.b0);
  const float thresh = u * acc;
  const unsigned long long below = __ballot(is_logit && cdf <= thresh);
  int act = __popcll(below & 0xffffffffull);
  if (act > A - 1) act = A - 1;
  const float lse = mx + logf(acc);
  const float la = lane_value(x, act) - lse;
  const float val = lane_value(x, A);
  if (lane == 0) {
    p.actions[b] = act;
    p.log_prob[b] = la;
    p.values[b] = val;
  }
}

__global__ __launch_bounds__(256) void tail_act_kernel(const TailActArgs p) {
  extern __shared__ __attribute__((aligned(16))) float ta_smem[];
  tail_act_block(p, blockIdx.x, ta_smem);
}

// with the synthetic device env's next observation batch / rewards / resets in the other workgroups
// (heads_act_synth_kernel's arrangement)
__global__ __launch_bounds__(256) void tail_act_synth_kernel(const TailActArgs p, const dx::SynthArgs e, int act_blocks) {
  extern __shared__ __attribute__((aligned(16))) float ta_smem[];
  if (static_cast<int>(blockIdx.x) < act_blocks) tail_act_block(p, blockIdx.x, ta_smem);
  else dx::synth_atari_block(e, blockIdx.x - act_blocks, gridDim.x - act_blocks);
}

// the categorical loss_reduce_kernel with a row stride of 40 doubles
__global__ __launch_bounds__(256) void loss_reduce40_kernel(const double *partials, int nblocks,
                                                            double count, float value_loss_coef,
                                                            float entropy_coef, float *out, int P,
                                                            float *dlogstd) {
  __shared__ double red[4][8];
  if (dlogstd != nullptr && static_cast<int>(threadIdx.x) < P) {  // dL/dlogstd: the block partials summed in block order
    double t = 0.0;
    for (int i = 0; i < nblocks; ++i) t += partials[i * 40 + 8 + threadIdx.x];
    dlogstd[threadIdx.x] = static_cast<float>(t);
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += partials[i * 40 + j];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] = wave_sum_d(s[j]);
    if (lane == 0) red[wave][j] = s[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[8];
    for (int j = 0; j < 8; ++j) t[j] = (red[0][j] + red[1][j] + red[2][j] + red[3][j]);
    const float policy = static_cast<float>(t[0] / count);
    const float ent = static_cast<float>(t[1] / count);
    const float value = static_cast<float>(t[2] / count);
    out[0] = (policy - entropy_coef * ent) + value_loss_coef * value;
    out[1] = policy; out[2] = ent; out[3] = value;
    out[4] = static_cast<float>(t[3] / count);
    out[5] = static_cast<float>(t[4] / count);
    out[6] = static_cast<float>(t[5] / count);
    const double mean_v = t[4] / count;
    const double var_v = count > 1 ? (t[7] - count * mean_v * mean_v) / (count - 1) : 0.0;
    out[7] = static_cast<float>(1.0 - (t[6] / count) / var_v);
  }
}

}  // namespace

extern "C" int dx_normal_act_f32(const float *head_out, const float *logstd, int B, int P,
                                 const float *normals, uint64_t seed, uint64_t counter, float *actions,
                                 float *log_prob, float *values, void *stream) {
  DX_TRACE("dx_normal_act_f32");
  DX_REQUIRE(B >= 0 && P >= 1 && P <= 31, "dx_normal_act_f32: need 1 <= P <= 31 (P=%d)", P);
  if (B == 0) return DX_OK;
  DX_REQUIRE(head_out && logstd && actions && log_prob && values, "dx_normal_act_f32: null pointer");
  hipLaunchKernelGGL(normal_act_kernel, dim3(dx::cdiv(B, 256)), dim3(256), 0, dx::as_stream(stream),
                     head_out, logstd, B, P, normals, seed, counter, actions, log_prob, values);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_normal_loss_f32(const float *head_out, const float *logstd, const float *actions,
                                  const float *old_log_prob, const float *advantages,
                                  const float *old_values, const float *value_targets, int B, int P,
                                  int mode, float cliprange, float value_loss_coef, float entropy_coef,
                                  long long global_batch, float *dhead_out, float *dlogstd_out,
                                  double *partials, int partials_capacity, float *loss_out,
                                  void *stream) {
  DX_TRACE("dx_normal_loss_f32");
  DX_REQUIRE(B >= 1 && P >= 1 && P <= 31, "dx_normal_loss_f32: need B >= 1 and 1 <= P <= 31");
  DX_REQUIRE(mode == 0 || mode == 1, "dx_normal_loss_f32: mode must be 0 (PPO) or 1 (A2C)");
  DX_REQUIRE(head_out && logstd && actions && advantages && value_targets && dhead_out && dlogstd_out &&
                 partials && loss_out,
             "dx_normal_loss_f32: null pointer");
  DX_REQUIRE(mode == 1 || (old_log_prob && old_values), "dx_normal_loss_f32: PPO needs old_log_prob/old_values");
  const int blocks = dx::cdiv(B, 256);
  DX_REQUIRE(partials_capacity >= blocks * 40, "dx_normal_loss_f32: partials needs %d doubles", blocks * 40);
  if (global_batch <= 0) global_batch = B;
  NormalLossArgs a{head_out, logstd, actions, old_log_prob, advantages, old_values, value_targets, dhead_out,
                   partials, B, P, mode, cliprange, value_loss_coef, entropy_coef,
                   1.0f / static_cast<float>(global_batch)};
  hipStream_t s = dx::as_stream(stream);
  hipLaunchKernelGGL(normal_loss_kernel, dim3(blocks), dim3(256), 0, s, a);
  DX_LAUNCH_CHECK();
  // the loss terms and dL/dlogstd from the block partials in ONE single-block launch
  hipLaunchKernelGGL(loss_reduce40_kernel, dim3(1), dim3(256), 0, s, partials, blocks, static_cast<double>(B),
                     value_loss_coef, entropy_coef, loss_out, P, dlogstd_out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

namespace {
// (re-open the anonymous namespace closed above for the definitions that follow)
}  // namespace

extern "C" int dx_categorical_act_f32(const float *head_out, int B, int A, const float *uniforms,
                                      uint64_t seed, uint64_t counter, int64_t *actions,
                                      float *log_prob, float *values, void *stream) {
  DX_TRACE("dx_categorical_act_f32");
  DX_REQUIRE(B >= 0 && A >= 1 && A <= 31, "dx_categorical_act_f32: need 1 <= A <= 31 (A=%d)", A);
  if (B == 0) return DX_OK;
  DX_REQUIRE(head_out && actions && log_prob && values, "dx_categorical_act_f32: null pointer");
  hipLaunchKernelGGL(categorical_act_kernel, dim3(dx::cdiv(B, 8)), dim3(256), 0, dx::as_stream(stream),
                     head_out, B, A, uniforms, seed, counter, actions, log_prob, values);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_categorical_loss_f32(const float *head_out, const int64_t *actions,
                                       const float *old_log_prob, const float *advantages,
                                       const float *old_values, const float *value_targets, int B,
                                       int A, int mode, float cliprange, float value_loss_coef,
                                       float entropy_coef, long long global_batch, float *dhead_out,
                                       double *partials, int partials_capacity, float *loss_out,
                                       void *stream) {
  DX_TRACE("dx_categorical_loss_f32");
  DX_REQUIRE(B >= 1 && A >= 1 && A <= 31, "dx_categorical_loss_f32: need B >= 1 and 1 <= A <= 31");
  DX_REQUIRE(mode == 0 || mode == 1, "dx_categorical_loss_f32: mode must be 0 (PPO) or 1 (A2C)");
  DX_REQUIRE(head_out && actions && advantages && value_targets && dhead_out && partials && loss_out,
             "dx_categorical_loss_f32: null pointer");
  DX_REQUIRE(mode == 1 || (old_log_prob && old_values), "dx_categorical_loss_f32: PPO needs old_log_prob/old_values");
  const int blocks = dx::cdiv(B, 8);
  DX_REQUIRE(partials_capacity >= blocks * 8, "dx_categorical_loss_f32: partials needs %d doubles", blocks * 8);
  if (global_batch <= 0) global_batch = B;
  LossArgs a{head_out, actions, old_log_prob, advantages, old_values, value_targets, dhead_out,
             partials, B, A, mode, cliprange, value_loss_coef, entropy_coef,
             1.0f / static_cast<float>(global_batch)};
  hipStream_t s = dx::as_stream(stream);
  hipLaunchKernelGGL(categorical_loss_kernel, dim3(blocks), dim3(256), 0, s, a);
  DX_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, partials, blocks,
                     static_cast<double>(B), value_loss_coef, entropy_coef, loss_out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

namespace dx {
// DX_ENOSUP: more than 7 actions (the A + 1 weight rows live in registers): the caller keeps the
// separate heads / loss launches
int launch_heads_loss_fused(const float *hid, const float *Wh, const float *bh, const int64_t *actions,
                            const float *old_log_prob, const float *advantages, const float *old_values,
                            const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out,
                            float *head, float *dhead, float *dhid, float *slab, float *bias_slab, int nslab,
                            int rows_per_slab, int B, int A, int mode, float cliprange, float value_loss_coef,
                            float entropy_coef, long long global_batch, double *partials, int partials_capacity,
                            unsigned *counter, float *loss_out, hipStream_t stream) {
  if (A + 1 > 8) return DX_ENOSUP;
  DX_REQUIRE(B >= 1 && A >= 1 && nslab >= 1 && rows_per_slab >= 1 && static_cast<long long>(nslab) * rows_per_slab >= B,
             "heads_loss: bad shape B=%d A=%d slabs=%d x %d rows", B, A, nslab, rows_per_slab);
  DX_REQUIRE(mode == 0 || mode == 1, "heads_loss: mode must be 0 (PPO) or 1 (A2C)");
  DX_REQUIRE(hid && Wh && bh && actions && advantages && value_targets && head && dhead && dhid && slab && bias_slab &&
                 partials && counter && loss_out, "heads_loss: null pointer");
  DX_REQUIRE(mode == 1 || (old_log_prob && old_values), "heads_loss: PPO needs old_log_prob / old_values");
  DX_REQUIRE(partials_capacity >= 8 * nslab, "heads_loss: partials needs %d doubles", 8 * nslab);
  if (global_batch <= 0) global_batch = B;
  const HeadsLossArgs a{hid, Wh, bh, actions, old_log_prob, advantages, old_values, value_targets, stats, norm_eps,
                        adv_norm_out, head, dhead, dhid, slab, bias_slab, partials, counter, loss_out, B, A,
                        rows_per_slab, mode, cliprange, value_loss_coef, entropy_coef,
                        1.0f / static_cast<float>(global_batch)};
  constexpr int lds = (kHlWaves * 8 * 512 + kHlWaves * 32) * 4 + kHlWaves * 8 * 8;
  DX_LDS_OPT_IN(heads_loss_fused_kernel, lds);
  hipLaunchKernelGGL(heads_loss_fused_kernel, dim3(nslab), dim3(64 * kHlWaves), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_heads_act_fused(const float *hid_slabs, int nslab, long long slab_stride, const float *Wh,
                           const float *bh, int B, int A, const float *uniforms, uint64_t seed,
                           uint64_t counter, int64_t *actions, float *log_prob, float *values,
                           hipStream_t stream) {
  DX_REQUIRE(B >= 1 && A >= 1 && A <= 31 && nslab >= 1, "heads_act: bad shape B=%d A=%d nslab=%d", B, A, nslab);
  DX_REQUIRE(hid_slabs && Wh && bh && actions && log_prob && values, "heads_act: null pointer");
  const HeadsActArgs p{hid_slabs, nslab, slab_stride, Wh, bh, B, A, uniforms, seed, counter, actions, log_prob, values, 0};
  hipLaunchKernelGGL(heads_act_fused_kernel, dim3(cdiv(B, 4)), dim3(256), 0, stream, p);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_heads_act_synth(const float *hid_slabs, int nslab, long long slab_stride, const float *Wh,
                           const float *bh, int B, int A, uint64_t seed, uint64_t counter, int64_t *actions,
                           float *log_prob, float *values, void *frames, long long frame_bytes, float *rewards,
                           uint8_t *resets, uint64_t env_seed, uint64_t env_counter, float p_reward,
                           float p_reset, int env0, long long vec0, hipStream_t stream) {
  DX_REQUIRE(B >= 1 && A >= 1 && A <= 31 && nslab >= 1, "heads_act_synth: bad shape B=%d A=%d nslab=%d", B, A, nslab);
  DX_REQUIRE(hid_slabs && Wh && bh && actions && log_prob && values, "heads_act_synth: null pointer");
  DX_REQUIRE(frames && frame_bytes > 0 && frame_bytes % 16 == 0 && aligned(frames, 16),
             "heads_act_synth: frames must be 16-byte aligned, size a multiple of 16");
  const HeadsActArgs p{hid_slabs, nslab, slab_stride, Wh, bh, B, A, nullptr, seed, counter, actions, log_prob, values, env0};
  const SynthArgs e{static_cast<uint4 *>(frames), frame_bytes / 16, rewards, resets, B, env_seed, env_counter,
                    p_reward, p_reset, vec0, env0};
  const int hb = cdiv(B, 4);
  hipLaunchKernelGGL(heads_act_synth_kernel, dim3(hb + synth_blocks(e.nvec, B)), dim3(256), 0, stream, p, e, hb);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// forward + loss of the factored tail AND its backward pass over y2 in one launch (tail_loss_bwd_kernel): up to 7
// actions; `gslab` / `sslab` / `nwg` / `rows_per_wg` are launch_tail_bwd's (tail.hip owns the slab layout)
template <int NJ>
static int launch_tail_loss_bwd_as(const TailLossArgs &a, const TailBwdOut &o, int nwg, hipStream_t stream) {
  hipLaunchKernelGGL(tail_loss_bwd_kernel<NJ>, dim3(nwg), dim3(kFbThreads), 0, stream, a, o);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
int launch_tail_loss_bwd(const float *y2, const float *Wc, const float *beff, const int64_t *actions,
                         const float *old_log_prob, const float *advantages, const float *old_values,
                         const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out, float *head,
                         float *dhead, int B, int A, int mode, float cliprange, float value_loss_coef, float entropy_coef,
                         long long global_batch, double *partials, int partials_capacity, unsigned *counter,
                         float *loss_out, float *dy2, float *gslab, float *sslab, int Jp, int nwg, int rows_per_wg,
                         hipStream_t stream) {
  if (A + 1 > 8) return DX_ENOSUP;
  DX_REQUIRE(B >= 1 && A >= 1, "tail_loss_bwd: bad shape B=%d A=%d", B, A);
  DX_REQUIRE(mode == 0 || mode == 1, "tail_loss_bwd: mode must be 0 (PPO) or 1 (A2C)");
  DX_REQUIRE(y2 && Wc && beff && actions && advantages && value_targets && head && dhead && partials && counter && loss_out &&
                 dy2 && gslab && sslab,
             "tail_loss_bwd: null pointer");
  DX_REQUIRE(mode == 1 || (old_log_prob && old_values), "tail_loss_bwd: PPO needs old_log_prob / old_values");
  DX_REQUIRE(nwg >= 1 && rows_per_wg >= 1 && static_cast<long long>(nwg) * rows_per_wg >= B, "tail_loss_bwd: %d workgroups x %d rows < %d",
             nwg, rows_per_wg, B);
  DX_REQUIRE(partials_capacity >= 8 * nwg, "tail_loss_bwd: partials needs %d doubles", 8 * nwg);
  if (global_batch <= 0) global_batch = B;
  const TailLossArgs a{y2, Wc, beff, actions, old_log_prob, advantages, old_values, value_targets, stats, norm_eps,
                       adv_norm_out, head, dhead, partials, counter, loss_out, B, A, rows_per_wg, mode, cliprange,
                       value_loss_coef, entropy_coef, 1.0f / static_cast<float>(global_batch)};
  const TailBwdOut o{dy2, gslab, sslab, Jp};
  switch (A + 1) {
    case 2: return launch_tail_loss_bwd_as<2>(a, o, nwg, stream);
    case 3: return launch_tail_loss_bwd_as<3>(a, o, nwg, stream);
    case 4: return launch_tail_loss_bwd_as<4>(a, o, nwg, stream);
    case 5: return launch_tail_loss_bwd_as<5>(a, o, nwg, stream);
    case 6: return launch_tail_loss_bwd_as<6>(a, o, nwg, stream);
    case 7: return launch_tail_loss_bwd_as<7>(a, o, nwg, stream);
    default: return launch_tail_loss_bwd_as<8>(a, o, nwg, stream);
  }
}

// forward + loss of the factored tail (tail_loss_kernel); DX_ENOSUP for more than 18 actions
int launch_tail_loss(const float *y2, const float *Wc, const float *beff, const int64_t *actions,
                     const float *old_log_prob, const float *advantages, const float *old_values,
                     const float *value_targets, const double *stats, float norm_eps, float *adv_norm_out, float *head,
                     float *dhead, int B, int A, int mode, float cliprange, float value_loss_coef, float entropy_coef,
                     long long global_batch, double *partials, int partials_capacity, unsigned *counter,
                     float *loss_out, hipStream_t stream) {
  if (A + 1 > 19) return DX_ENOSUP;
  DX_REQUIRE(B >= 1 && A >= 1, "tail_loss: bad shape B=%d A=%d", B, A);
  DX_REQUIRE(mode == 0 || mode == 1, "tail_loss: mode must be 0 (PPO) or 1 (A2C)");
  DX_REQUIRE(y2 && Wc && beff && actions && advantages && value_targets && head && dhead && partials && counter && loss_out,
             "tail_loss: null pointer");
  DX_REQUIRE(mode == 1 || (old_log_prob && old_values), "tail_loss: PPO needs old_log_prob / old_values");
  // 16 rows per workgroup at least (one pass of its 8 waves x 2 rows), 256 workgroups at most
  int nwg = cdiv(B, 16) < 256 ? cdiv(B, 16) : 256;
  const int rows = cdiv(B, nwg);
  nwg = cdiv(B, rows);
  DX_REQUIRE(partials_capacity >= 8 * nwg, "tail_loss: partials needs %d doubles", 8 * nwg);
  if (global_batch <= 0) global_batch = B;
  const TailLossArgs a{y2, Wc, beff, actions, old_log_prob, advantages, old_values, value_targets, stats, norm_eps,
                       adv_norm_out, head, dhead, partials, counter, loss_out, B, A, rows, mode, cliprange,
                       value_loss_coef, entropy_coef, 1.0f / static_cast<float>(global_batch)};
  const int lds_rows = A + 1 < 8 ? 8 : (A + 1 < kTailLdsRows ? A + 1 : kTailLdsRows);
  const int lds = lds_rows * kTailK * 4 + kTlWaves * 8 * 8;
  DX_LDS_OPT_IN(tail_loss_kernel, kTailLdsRows * kTailK * 4 + kTlWaves * 8 * 8);
  hipLaunchKernelGGL(tail_loss_kernel, dim3(nwg), dim3(64 * kTlWaves), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

static int configure_tail_act() {
  DX_LDS_OPT_IN(tail_act_kernel, kTailLdsRows * kTailK * 4);
  DX_LDS_OPT_IN(tail_act_synth_kernel, kTailLdsRows * kTailK * 4);
  return DX_OK;
}

// the rollout's factored tail: y2 (B, 3136) -> actions / log_prob / values
int launch_tail_act(const float *y2, const float *Wc, const float *beff, int B, int A, const float *uniforms, uint64_t seed,
                    uint64_t counter, int64_t *actions, float *log_prob, float *values, hipStream_t stream) {
  DX_REQUIRE(B >= 1 && A >= 1 && A + 1 <= 19, "tail_act: bad shape B=%d A=%d", B, A);
  DX_REQUIRE(y2 && Wc && beff && actions && log_prob && values, "tail_act: null pointer");
  if (int rc = configure_tail_act()) return rc;
  const TailActArgs p{y2, Wc, beff, B, A, uniforms, seed, counter, actions, log_prob, values, 0};
  hipLaunchKernelGGL(tail_act_kernel, dim3(cdiv(B, 4)), dim3(256), (A + 1 < kTailLdsRows ? A + 1 : kTailLdsRows) * kTailK * 4, stream, p);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_tail_act_synth(const float *y2, const float *Wc, const float *beff, int B, int A, uint64_t seed, uint64_t counter,
                          int64_t *actions, float *log_prob, float *values, void *frames, long long frame_bytes,
                          float *rewards, uint8_t *resets, uint64_t env_seed, uint64_t env_counter, float p_reward,
                          float p_reset, int env0, long long vec0, hipStream_t stream) {
  DX_REQUIRE(B >= 1 && A >= 1 && A + 1 <= 19, "tail_act_synth: bad shape B=%d A=%d", B, A);
  DX_REQUIRE(y2 && Wc && beff && actions && log_prob && values, "tail_act_synth: null pointer");
  DX_REQUIRE(frames && frame_bytes > 0 && frame_bytes % 16 == 0 && aligned(frames, 16),
             "tail_act_synth: frames must be 16-byte aligned, size a multiple of 16");
  if (int rc = configure_tail_act()) return rc;
  const TailActArgs p{y2, Wc, beff, B, A, nullptr, seed, counter, actions, log_prob, values, env0};
  const SynthArgs e{static_cast<uint4 *>(frames), frame_bytes / 16, rewards, resets, B, env_seed, env_counter,
                    p_reward, p_reset, vec0, env0};
  const int ab = cdiv(B, 4);
  hipLaunchKernelGGL(tail_act_synth_kernel, dim3(ab + synth_blocks(e.nvec, B)), dim3(256), (A + 1 < kTailLdsRows ? A + 1 : kTailLdsRows) * kTailK * 4, stream, p, e, ab);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
}  // namespace dx
