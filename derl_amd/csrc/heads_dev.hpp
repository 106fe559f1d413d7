// Device helpers shared by the head kernels (heads.hip) and the rollout's one-kernel step
// (convstack.hip): the counter-based uniform of the sampling rule and wave-wide sums without LDS.
#pragma once
#include "common.hpp"

namespace {

// counter-based uniform in [0,1): 2 rounds of a 64-bit mix (splitmix64 finaliser) of
// (seed, counter, row); 24 random bits -> exactly representable float
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t counter, uint64_t row) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter * 0x100000001B3ull + row + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);  // rows masked off receive 0
}

// sum over the 64 lanes, returned in every lane (gfx9 wave64 DPP reduction + readlane 63)
__device__ __forceinline__ float wave_sum_all(float v) {
  v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);   // row_mirror: every lane holds its row's (16 lanes) total
  v = dpp_add<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- diagonal Gaussian head: one row's sample (derl/policies.py:40-42,66), shared by dx_normal_act_f32 (heads.hip)
// and the one-launch MLP rollout (mlp_fused.hip) so that both write the same bits ----
constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // log(sqrt(2*pi))
__device__ __forceinline__ float normal01(uint64_t seed, uint64_t counter, uint64_t idx) {
  // Box-Muller from two counter-based uniforms; u1 in (0,1]
  const float u1 = 1.0f - uniform01(seed, counter * 2, idx);
  const float u2 = uniform01(seed ^ 0xA5A5A5A5A5A5A5A5ull, counter * 2 + 1, idx);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
// one action dimension: the sample and its term of the log-probability
__device__ __forceinline__ float normal_act_dim(float mu, float logstd_d, float eps, float *act_out) {
  const float sigma = expf(logstd_d);
  const float act = mu + sigma * eps;
  *act_out = act;
  const float diff = act - mu;
  return -(diff * diff) / (2.f * (sigma * sigma)) - logf(sigma) - kHalfLog2Pi;
}
// row = the policy's P means followed by the value; actions (P) and the log-probability of row b (terms added in
// dimension order)
__device__ __forceinline__ float normal_act_row(const float *row, const float *logstd, int P, const float *normals_row, uint64_t seed,
                                                uint64_t counter, long long b, float *actions_row) {
  float lp = 0.f;
  for (int d = 0; d < P; ++d) {
    const float eps = normals_row ? normals_row[d] : normal01(seed, counter, static_cast<uint64_t>(b) * 32 + d);
    lp += normal_act_dim(row[d], logstd[d], eps, actions_row + d);
  }
  return lp;
}

__device__ __forceinline__ float lane_value(float v, int lane_uniform) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}

}  // namespace
