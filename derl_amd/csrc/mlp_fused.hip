// Two-net tanh MLP actor-critic in ONE launch per direction (observations up to 64 wide).
//
// The GEMMs of derl/models.py:224-271 are tiny here (64-wide hidden layers, 21.6 KFLOP per
// sample): on the implicit-GEMM kernels one update is ~36 launches of 5-30 us each (BASELINE
// config 3 does 320 updates per rollout).  These kernels keep a tile
// of rows and all three weight matrices of one net in LDS and run the whole chain on the vector
// ALUs (plain fp32 fma chains, the arithmetic of the reference's fp32 Linear layers):
//   forward : x -> tanh -> tanh -> head columns, keeps xpad / h1 / h2 for the backward
//   backward: dhead -> per-workgroup partial dW / db of all three layers (slabs, summed by
//             permute_reduce as before) with the two tanh' dgrads in between.
// blockIdx.y = net (0 policy, 1 value); workgroups loop over row tiles.
#include "mlp_fused.hpp"
#include "heads_dev.hpp"
#include "synth_dev.hpp"

namespace dx {
namespace {

constexpr int kH = 64, kHeadLd = 32, kThreads = 256;
constexpr int kLdT = kH + 4;  // row stride of a 64-wide weight matrix read with lane = row: 16-byte aligned rows (one
                              // ds_read_b128 per four weights) whose 16-lane read groups still land on distinct banks (272 B = 17 quads)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// acc[rr] += sum_k w(k) * in[(row0 + rr) * ld + k], k in [0, K): W rows in LDS, stride = 4 (mod 8) floats
template <int ROWS, int K>
__device__ inline void dot_rows(float (&acc)[ROWS], const float *wrow, const float *in, int ld) {
#pragma unroll 4
  for (int k = 0; k < K; k += 4) {
    const f32x4 w = lds4(wrow + k);
    const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const f32x4 x = lds4(in + rr * ld + k);
      acc[rr] = fmaf(w0, x[0], acc[rr]);
      acc[rr] = fmaf(w1, x[1], acc[rr]);
      acc[rr] = fmaf(w2, x[2], acc[rr]);
      acc[rr] = fmaf(w3, x[3], acc[rr]);
    }
  }
}

template <int RPW, int DP>
__global__ __launch_bounds__(kThreads) void mlp_forward_fused_kernel(MlpFusedArgs a) {
  constexpr int R = 4 * RPW, LD0 = DP + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *W0 = lds;                  // [64][DP+4]
  float *W1 = W0 + kH * LD0;        // [64][68]
  float *W2 = W1 + kH * kLdT;       // [32][68]
  float *bs = W2 + kHeadLd * kLdT;  // b0[64] b1[64] b2[32]
  float *xs = bs + 160;             // [R][DP]
  float *hs = xs + R * DP;          // [R][64]
  float *gs = hs + R * kH;          // [R][64]
  const int net = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int D = a.D, outs = net == 0 ? a.P : 1, col = net == 0 ? 0 : a.P;
  const float *p = a.params;
  {
    const float *w0 = p + a.off_w[3 * net], *w1 = p + a.off_w[3 * net + 1], *w2 = p + a.off_w[3 * net + 2];
    for (int i = t; i < kH * DP; i += kThreads) {
      const int j = i / DP, k = i - j * DP;
      W0[j * LD0 + k] = k < D ? w0[j * D + k] : 0.f;
    }
    for (int i = t; i < kH * kH; i += kThreads) W1[(i >> 6) * kLdT + (i & 63)] = w1[i];
    for (int i = t; i < kHeadLd * kH; i += kThreads)
      W2[(i >> 6) * kLdT + (i & 63)] = (i >> 6) < outs ? w2[i] : 0.f;
    if (t < kH) {
      bs[t] = p[a.off_b[3 * net] + t];
      bs[kH + t] = p[a.off_b[3 * net + 1] + t];
      if (t < kHeadLd) bs[2 * kH + t] = t < outs ? p[a.off_b[3 * net + 2] + t] : 0.f;
    }
  }
  float *h1 = a.h1[net], *h2 = a.h2[net];
  const int ntiles = (a.B + R - 1) / R;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long row0 = static_cast<long long>(tile) * R;
    for (int i = t; i < R * DP; i += kThreads) {
      const int r = i / DP, k = i - r * DP;
      const long long row = row0 + r;
      const float v = (row < a.B && k < D) ? a.obs[row * D + k] : 0.f;
      xs[i] = v;
      if (net == 0 && row < a.B) a.xpad[row * DP + k] = v;
    }
    __syncthreads();
    const int r0 = wave * RPW;
    float acc[RPW];
    // layer 0
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) acc[rr] = bs[lane];
    dot_rows<RPW, DP>(acc, W0 + lane * LD0, xs + r0 * DP, DP);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const float h = tanhf(acc[rr]);
      hs[(r0 + rr) * kH + lane] = h;
      if (row0 + r0 + rr < a.B) h1[(row0 + r0 + rr) * kH + lane] = h;
    }
    __syncthreads();
    // layer 1
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) acc[rr] = bs[kH + lane];
    dot_rows<RPW, kH>(acc, W1 + lane * kLdT, hs + r0 * kH, kH);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const float h = tanhf(acc[rr]);
      gs[(r0 + rr) * kH + lane] = h;
      if (row0 + r0 + rr < a.B) h2[(row0 + r0 + rr) * kH + lane] = h;
    }
    __syncthreads();
    // layer 2: lane = (half of the wave's rows, output column)
    {
      constexpr int HR = RPW / 2;
      const int o = lane & 31, rh = r0 + (lane >> 5) * HR;
      float out[HR];
#pragma unroll
      for (int rr = 0; rr < HR; ++rr) out[rr] = bs[2 * kH + o];
      dot_rows<HR, kH>(out, W2 + o * kLdT, gs + rh * kH, kH);
      if (o < outs) {
#pragma unroll
        for (int rr = 0; rr < HR; ++rr)
          if (row0 + rh + rr < a.B) a.head[(row0 + rh + rr) * kHeadLd + col + o] = out[rr];
      }
    }
    // the next tile's staging only writes xs (last read before the first barrier above)
  }
}

// The whole rollout horizon of the Gaussian MLP policy against the MuJoCo-shaped synthetic env in ONE launch
// (derl/runners/env_runner.py:43-65 for the measurement env): the env's next observation is a hash of (seed,
// counter, env, component) and does not depend on the action, and a policy step reads only its own env's
// observation, so a workgroup keeps 8 envs for all T steps with BOTH nets' weights resident in LDS -- per step
// the forward of mlp_forward_fused_kernel (the same fma chains per row: identical bits), the sample of
// dx_normal_act_f32 (heads_dev.hpp: normal_act_row) and dx_synth_mujoco_step's values.  The per-step loop is
// ~11 launches per step (two native, the rest the env's); config 3's 64 steps took 2.3 ms of its 15.4 ms iteration.
template <int DP>
__global__ __launch_bounds__(2 * kThreads) void mlp_rollout_synth_kernel(const MlpRolloutArgs a) {
  constexpr int RPW = 2, R = 4 * RPW, LD0 = DP + 4;
  constexpr int kNet = kH * LD0 + kH * kLdT + kHeadLd * kLdT + 160;  // floats of one net's weights and biases
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds + 2 * kNet;     // [R][DP]
  float *hsb = xs + R * DP;       // [2 nets][R][64]
  float *gsb = hsb + 2 * R * kH;  // [2 nets][R][64]
  float *hd = gsb + 2 * R * kH;   // [R][32]: the policy's means, then the value
  float *lpt = hd + R * kHeadLd;  // [R][32]: the log-probability's terms per action dimension
  // threads 0-255: the policy net, 256-511: the value net -- the two chains of a step run side by side
  const int t = threadIdx.x, net = t >> 8, tt = t & 255, lane = tt & 63, wave = tt >> 6;
  const int D = a.f.D, P = a.f.P, N = a.f.B;
  const float *p = a.f.params;
  float *W0 = lds + net * kNet, *W1 = W0 + kH * LD0, *W2 = W1 + kH * kLdT, *bs = W2 + kHeadLd * kLdT;
  float *hs = hsb + net * R * kH, *gs = gsb + net * R * kH;
  const int outs = net == 0 ? P : 1, col = net == 0 ? 0 : P;
  {
    const float *w0 = p + a.f.off_w[3 * net], *w1 = p + a.f.off_w[3 * net + 1], *w2 = p + a.f.off_w[3 * net + 2];
    for (int i = tt; i < kH * DP; i += kThreads) {
      const int j = i / DP, k = i - j * DP;
      W0[j * LD0 + k] = k < D ? w0[j * D + k] : 0.f;
    }
    for (int i = tt; i < kH * kH; i += kThreads) W1[(i >> 6) * kLdT + (i & 63)] = w1[i];
    for (int i = tt; i < kHeadLd * kH; i += kThreads) W2[(i >> 6) * kLdT + (i & 63)] = (i >> 6) < outs ? w2[i] : 0.f;
    if (tt < kH) {
      bs[tt] = p[a.f.off_b[3 * net] + tt];
      bs[kH + tt] = p[a.f.off_b[3 * net + 1] + tt];
      if (tt < kHeadLd) bs[2 * kH + tt] = tt < outs ? p[a.f.off_b[3 * net + 2] + tt] : 0.f;
    }
  }
  const long long row0 = static_cast<long long>(blockIdx.x) * R;
  for (int i = t; i < R * DP; i += 2 * kThreads) {
    const int r = i / DP, k = i - r * DP;
    xs[i] = (row0 + r < N && k < D) ? a.obs[(row0 + r) * D + k] : 0.f;
  }
  __syncthreads();
  const int r0 = wave * RPW;
  for (int step = 0; step < a.T; ++step) {
    {
      float acc[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) acc[rr] = bs[lane];
      dot_rows<RPW, DP>(acc, W0 + lane * LD0, xs + r0 * DP, DP);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) hs[(r0 + rr) * kH + lane] = tanhf(acc[rr]);
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) acc[rr] = bs[kH + lane];
      dot_rows<RPW, kH>(acc, W1 + lane * kLdT, hs + r0 * kH, kH);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) gs[(r0 + rr) * kH + lane] = tanhf(acc[rr]);
      __syncthreads();
      constexpr int HR = RPW / 2;
      const int o = lane & 31, rh = r0 + (lane >> 5) * HR;
      float out[HR];
#pragma unroll
      for (int rr = 0; rr < HR; ++rr) out[rr] = bs[2 * kH + o];
      dot_rows<HR, kH>(out, W2 + o * kLdT, gs + rh * kH, kH);
      if (o < outs) {
#pragma unroll
        for (int rr = 0; rr < HR; ++rr) hd[(rh + rr) * kHeadLd + col + o] = out[rr];
      }
    }
    __syncthreads();
    // ---- sample: thread = (row, action dimension); the env's step for these rows by the other half ----
    const long long srow = static_cast<long long>(step) * N;
    if (t < R * 32) {
      const int r = t >> 5, d = t & 31;
      const long long b = row0 + r;
      if (d < P && b < N)
        lpt[r * 32 + d] = normal_act_dim(hd[r * kHeadLd + d], p[a.off_logstd + d], normal01(a.policy_seed, a.policy_counter + step, static_cast<uint64_t>(b) * 32 + d),
                                         a.actions + (srow + b) * P + d);
    } else {
      const uint64_t key = synth_mujoco_key(a.env_seed, a.env_counter + step);
      float *next = a.obs + static_cast<long long>(step + 1) * N * D;
      const int i = t - R * 32;  // R * 32 <= 256 threads left: one (row, component) each for DP <= 32, two for 64
      for (int j = i; j < R * DP; j += 2 * kThreads - R * 32) {
        const int r = j / DP, k = j - r * DP;
        const long long b = row0 + r;
        float v = 0.f;
        if (b < N && k < D) {
          v = synth_mujoco_obs(key, b, k);
          next[b * D + k] = v;
        }
        xs[j] = v;
        if (k == 0 && b < N) {
          a.rewards[srow + b] = synth_mujoco_reward(key, b);
          a.resets[srow + b] = synth_mujoco_reset(key, b, a.p_reset) ? 1 : 0;
        }
      }
    }
    __syncthreads();
    if (t < R && row0 + t < N) {  // the terms in dimension order, as dx_normal_act_f32 adds them
      float lp = 0.f;
      for (int d = 0; d < P; ++d) lp += lpt[t * 32 + d];
      a.log_prob[srow + row0 + t] = lp;
      a.values[srow + row0 + t] = hd[t * kHeadLd + P];
    }
    // (hd / lpt are rewritten only behind the next step's barriers)
  }
}

// acc[rr] += sum_i in[(r0+rr)*64 + i] * W[i*64 + lane], i in [0, K): W rows [i][j] (lane = j)
template <int ROWS>
__device__ inline void dot_cols(float (&acc)[ROWS], const float *W, const float *in, int K, int lane) {
  for (int i = 0; i < K; i += 4) {
    const float w0 = W[i * kH + lane], w1 = W[(i + 1) * kH + lane], w2 = W[(i + 2) * kH + lane],
                w3 = W[(i + 3) * kH + lane];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const f32x4 g = lds4(in + rr * kH + i);
      acc[rr] = fmaf(g[0], w0, acc[rr]);
      acc[rr] = fmaf(g[1], w1, acc[rr]);
      acc[rr] = fmaf(g[2], w2, acc[rr]);
      acc[rr] = fmaf(g[3], w3, acc[rr]);
    }
  }
}

template <int RPW, int DP>
__global__ __launch_bounds__(kThreads) void mlp_backward_fused_kernel(MlpFusedArgs a) {
  constexpr int R = 4 * RPW, KG = DP / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *W1 = lds;              // [64][64]  W1[i][j]
  float *W2 = W1 + kH * kH;     // [32][64]  rows = head columns of this net (others zero)
  float *xs = W2 + kHeadLd * kH;  // [R][DP]
  float *h1s = xs + R * DP;     // [R][64]
  float *h2s = h1s + R * kH;    // [R][64]; reused for dL/d(pre-tanh 1)
  float *g2s = h2s + R * kH;    // [R][64]  dL/d(pre-tanh 2)
  float *ds = g2s + R * kH;     // [R][32]  dhead, all 32 columns
  float *g1s = h2s;
  const int net = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int outs = net == 0 ? a.P : 1, col = net == 0 ? 0 : a.P;
  {
    const float *w1 = a.params + a.off_w[3 * net + 1], *w2 = a.params + a.off_w[3 * net + 2];
    for (int i = t; i < kH * kH; i += kThreads) W1[i] = w1[i];
    for (int i = t; i < kHeadLd * kH; i += kThreads) {
      const int c = (i >> 6) - col;  // head column -> output row of this net
      W2[i] = (c >= 0 && c < outs) ? w2[c * kH + (i & 63)] : 0.f;
    }
  }
  // accumulators: dW1[i][jg..jg+15], dW0[i][kg..kg+KG-1], dW2[c][j8..j8+7] + the bias sums
  const int wi = t >> 2, jg = (t & 3) * 16, kg = (t & 3) * KG, wc = t >> 3, j8 = (t & 7) * 8;
  float aw1[16], aw0[KG], aw2[8], ab1 = 0.f, ab0 = 0.f, ab2 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) aw1[q] = 0.f;
#pragma unroll
  for (int q = 0; q < KG; ++q) aw0[q] = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) aw2[q] = 0.f;
  const float *h1 = a.h1[net], *h2 = a.h2[net];
  const int ntiles = (a.B + R - 1) / R;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long row0 = static_cast<long long>(tile) * R;
    const long long rows = a.B - row0 < R ? a.B - row0 : R;
    __syncthreads();  // the previous tile's readers are done (also orders the weight staging)
    for (int i = t; i < R * DP / 4; i += kThreads) {
      const bool ok = i < rows * (DP / 4);
      reinterpret_cast<f32x4 *>(xs)[i] = ok ? reinterpret_cast<const f32x4 *>(a.xpad + row0 * DP)[i] : f32x4{0, 0, 0, 0};
    }
    for (int i = t; i < R * kH / 4; i += kThreads) {
      const bool ok = i < rows * (kH / 4);
      reinterpret_cast<f32x4 *>(h1s)[i] = ok ? reinterpret_cast<const f32x4 *>(h1 + row0 * kH)[i] : f32x4{0, 0, 0, 0};
      reinterpret_cast<f32x4 *>(h2s)[i] = ok ? reinterpret_cast<const f32x4 *>(h2 + row0 * kH)[i] : f32x4{0, 0, 0, 0};
    }
    for (int i = t; i < R * kHeadLd / 4; i += kThreads) {
      const bool ok = i < rows * (kHeadLd / 4);
      reinterpret_cast<f32x4 *>(ds)[i] = ok ? reinterpret_cast<const f32x4 *>(a.dhead + row0 * kHeadLd)[i] : f32x4{0, 0, 0, 0};
    }
    __syncthreads();
    // wgrad of layer 2 (all 32 head columns; the reduction picks this net's rows)
    for (int r = 0; r < R; ++r) {
      const float d = ds[r * kHeadLd + wc];
      const f32x4 u = lds4(h2s + r * kH + j8), v = lds4(h2s + r * kH + j8 + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        aw2[q] = fmaf(d, u[q], aw2[q]);
        aw2[4 + q] = fmaf(d, v[q], aw2[4 + q]);
      }
      ab2 += d;
    }
    // dgrad through layer 2 and tanh'
    const int r0 = wave * RPW;
    {
      float acc[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) acc[rr] = 0.f;
      for (int c = col; c < col + outs; ++c) {
        const float w = W2[c * kH + lane];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) acc[rr] = fmaf(ds[(r0 + rr) * kHeadLd + c], w, acc[rr]);
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const float y = h2s[(r0 + rr) * kH + lane];
        g2s[(r0 + rr) * kH + lane] = acc[rr] * (1.f - y * y);
      }
    }
    __syncthreads();
    // wgrad of layer 1
    for (int r = 0; r < R; ++r) {
      const float d = g2s[r * kH + wi];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 u = lds4(h1s + r * kH + jg + 4 * q4);
#pragma unroll
        for (int q = 0; q < 4; ++q) aw1[4 * q4 + q] = fmaf(d, u[q], aw1[4 * q4 + q]);
      }
      ab1 += d;
    }
    // dgrad through layer 1 and tanh' (h2s is dead: every wave passed the barrier above)
    {
      float acc[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) acc[rr] = 0.f;
      dot_cols<RPW>(acc, W1, g2s + r0 * kH, kH, lane);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const float y = h1s[(r0 + rr) * kH + lane];
        g1s[(r0 + rr) * kH + lane] = acc[rr] * (1.f - y * y);
      }
    }
    __syncthreads();
    // wgrad of layer 0
    for (int r = 0; r < R; ++r) {
      const float d = g1s[r * kH + wi];
#pragma unroll
      for (int q4 = 0; q4 < KG / 4; ++q4) {
        const f32x4 u = lds4(xs + r * DP + kg + 4 * q4);
#pragma unroll
        for (int q = 0; q < 4; ++q) aw0[4 * q4 + q] = fmaf(d, u[q], aw0[4 * q4 + q]);
      }
      ab0 += d;
    }
  }
  // partial sums of this workgroup -> its slab
  const long long z = blockIdx.x, cap = a.ms_cap;
  float *base = a.slabs + net * a.slab_per_net;
  float *s0w = base, *s0b = s0w + cap * kH * DP, *s1w = s0b + cap * kH, *s1b = s1w + cap * kH * kH,
        *s2w = s1b + cap * kH, *s2b = s2w + cap * kHeadLd * kH;
#pragma unroll
  for (int q4 = 0; q4 < KG / 4; ++q4)
    *reinterpret_cast<f32x4 *>(s0w + z * kH * DP + wi * DP + kg + 4 * q4) =
        f32x4{aw0[4 * q4], aw0[4 * q4 + 1], aw0[4 * q4 + 2], aw0[4 * q4 + 3]};
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4)
    *reinterpret_cast<f32x4 *>(s1w + z * kH * kH + wi * kH + jg + 4 * q4) =
        f32x4{aw1[4 * q4], aw1[4 * q4 + 1], aw1[4 * q4 + 2], aw1[4 * q4 + 3]};
#pragma unroll
  for (int q4 = 0; q4 < 2; ++q4)
    *reinterpret_cast<f32x4 *>(s2w + z * kHeadLd * kH + wc * kH + j8 + 4 * q4) =
        f32x4{aw2[4 * q4], aw2[4 * q4 + 1], aw2[4 * q4 + 2], aw2[4 * q4 + 3]};
  if ((t & 3) == 0) {
    s0b[z * kH + wi] = ab0;
    s1b[z * kH + wi] = ab1;
  }
  if ((t & 7) == 0) s2b[z * kHeadLd + wc] = ab2;
}

template <int RPW, int DP>
constexpr size_t forward_lds() {
  return sizeof(float) * (kH * (DP + 4) + kH * kLdT + kHeadLd * kLdT + 160 + 4 * RPW * DP + 2 * 4 * RPW * kH);
}

template <int RPW, int DP>
constexpr size_t backward_lds() {
  return sizeof(float) * (kH * kH + kHeadLd * kH + 4 * RPW * DP + 3 * 4 * RPW * kH + 4 * RPW * kHeadLd);
}

template <int RPW, int DP>
int forward_as(const MlpFusedArgs &a, hipStream_t s) {
  static_assert(forward_lds<RPW, DP>() <= 64 * 1024, "forward tile does not fit the default LDS limit");
  const int ntiles = cdiv(a.B, 4 * RPW);
  const size_t bytes = forward_lds<RPW, DP>();
  const dim3 grid(ntiles < 512 ? ntiles : 512, 2);
  hipLaunchKernelGGL((mlp_forward_fused_kernel<RPW, DP>), grid, dim3(kThreads), bytes, s, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

template <int RPW, int DP>
int backward_as(const MlpFusedArgs &a, hipStream_t s) {
  static_assert(backward_lds<RPW, DP>() <= 64 * 1024, "backward tile does not fit the default LDS limit");
  const size_t bytes = backward_lds<RPW, DP>();
  hipLaunchKernelGGL((mlp_backward_fused_kernel<RPW, DP>), dim3(a.nslab, 2), dim3(kThreads), bytes, s, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

bool mlp_fused_supported(int obs_pad) { return obs_pad == 32 || obs_pad == 64; }

// rows per workgroup tile: 32 for big batches, 16 / 8 so that small ones still spread over CUs
int mlp_fused_tile_rows(int B, int obs_pad) {
  if (obs_pad == 64) return B >= 2048 ? 16 : 8;
  return B >= 4096 ? 32 : B >= 1024 ? 16 : 8;
}

template <int DP>
int rollout_as(const MlpRolloutArgs &a, hipStream_t s) {
  constexpr int R = 8;
  constexpr size_t bytes = sizeof(float) * (2 * (kH * (DP + 4) + kH * kLdT + kHeadLd * kLdT + 160) + R * DP + 4 * R * kH + 2 * R * kHeadLd);
  static_assert(bytes <= 160 * 1024, "LDS");
  DX_LDS_OPT_IN(mlp_rollout_synth_kernel<DP>, static_cast<int>(bytes));
  hipLaunchKernelGGL((mlp_rollout_synth_kernel<DP>), dim3(cdiv(a.f.B, R)), dim3(2 * kThreads), bytes, s, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_mlp_rollout_synth(const MlpRolloutArgs &a, hipStream_t s) {
  DX_REQUIRE(a.obs && a.actions && a.log_prob && a.values && a.rewards && a.resets && a.T >= 1 && a.f.B >= 1 && a.off_logstd >= 0,
             "mlp_rollout_synth: bad arguments");
  if (a.f.Dp == 32) return rollout_as<32>(a, s);
  if (a.f.Dp == 64) return rollout_as<64>(a, s);
  return fail(DX_ENOSUP, "mlp_rollout_synth: observations wider than 64");
}

int launch_mlp_forward_fused(const MlpFusedArgs &a, hipStream_t s) {
  const int R = mlp_fused_tile_rows(a.B, a.Dp);
  if (a.Dp == 32) {
    if (R == 32) return forward_as<8, 32>(a, s);
    if (R == 16) return forward_as<4, 32>(a, s);
    return forward_as<2, 32>(a, s);
  }
  if (a.Dp == 64) {
    if (R == 16) return forward_as<4, 64>(a, s);
    return forward_as<2, 64>(a, s);
  }
  return fail(DX_ENOSUP, "mlp fused forward: obs_pad %d", a.Dp);
}

int launch_mlp_backward_fused(const MlpFusedArgs &a, hipStream_t s) {
  const int R = mlp_fused_tile_rows(a.B, a.Dp);
  if (a.nslab < 1 || a.nslab > a.ms_cap) return fail(DX_EINVAL, "mlp fused backward: nslab %d of %d", a.nslab, a.ms_cap);
  if (a.Dp == 32) {
    if (R == 32) return backward_as<8, 32>(a, s);
    if (R == 16) return backward_as<4, 32>(a, s);
    return backward_as<2, 32>(a, s);
  }
  if (a.Dp == 64) {
    if (R == 16) return backward_as<4, 64>(a, s);
    return backward_as<2, 64>(a, s);
  }
  return fail(DX_ENOSUP, "mlp fused backward: obs_pad %d", a.Dp);
}

}  // namespace dx
