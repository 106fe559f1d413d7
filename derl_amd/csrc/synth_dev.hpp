// Device body of the counter-based synthetic Atari env step, shared by its own launch
// (synth.hip) and the rollout's fused heads + env launch (heads.hip).  `block` / `nblocks`: the
// caller's share of a grid -- every value is a hash of (seed, counter, position), so the split
// over workgroups does not change the result.
#pragma once
#include "common.hpp"

namespace dx {
namespace {

struct SynthArgs {
  uint4 *frames;      // nvec 16-byte vectors: the next observation batch
  long long nvec;
  float *rewards;     // (nenvs) or nullptr
  uint8_t *resets;    // (nenvs) or nullptr
  int nenvs;
  uint64_t seed, counter;
  float p_reward, p_reset;
  long long vec0;     // position of frames[0] / rewards[0] in the whole batch: a launch that covers
  int env0;           // a slice of the envs draws the values the whole-batch launch would
};

__device__ __forceinline__ uint64_t synth_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ void synth_atari_block(const SynthArgs &a, int block, int nblocks) {
  const uint64_t key = synth_mix64(a.seed * 0x9E3779B97F4A7C15ull + a.counter);
  const long long stride = static_cast<long long>(nblocks) * 256;
  const long long gid = static_cast<long long>(block) * 256 + threadIdx.x;
  for (long long i = gid; i < a.nvec; i += stride) {
    const uint64_t p = static_cast<uint64_t>(i + a.vec0);
    const uint64_t x = synth_mix64(key + 2 * p * 0x9E3779B97F4A7C15ull);
    const uint64_t y = synth_mix64(key + (2 * p + 1) * 0x9E3779B97F4A7C15ull);
    a.frames[i] = make_uint4(static_cast<uint32_t>(x), static_cast<uint32_t>(x >> 32),
                             static_cast<uint32_t>(y), static_cast<uint32_t>(y >> 32));
  }
  for (long long e = gid; e < a.nenvs; e += stride) {
    const uint64_t r = synth_mix64(~key + static_cast<uint64_t>(e + a.env0) * 0xD1B54A32D192ED03ull);
    const float u0 = static_cast<float>(r & 0xffffff) * (1.0f / 16777216.0f);
    const float u1 = static_cast<float>((r >> 24) & 0xffffff) * (1.0f / 16777216.0f);
    if (a.rewards) a.rewards[e] = u0 < a.p_reward ? ((r >> 63) ? -1.f : 1.f) : 0.f;
    if (a.resets) a.resets[e] = u1 < a.p_reset ? 1 : 0;
  }
}

// ---- the MuJoCo-shaped measurement env (derl_amd/env/synthetic.py: SyntheticMuJoCoEnv): float32 observations
// N(0, 1) clipped to +-10 (the range derl/env/mujoco_wrappers.py:64-124's Normalize produces), rewards N(0, 1), resets
// Bernoulli(p) -- every value a hash of (seed, counter, env, component), so any split over launches and workgroups
// draws the same numbers (synth.hip: dx_synth_mujoco_step; mlp_rollout.hip: the whole horizon in one launch) ----
__device__ __forceinline__ float synth_normal(uint64_t key, uint64_t idx) {  // Box-Muller on two 24-bit uniforms
  const uint64_t r = synth_mix64(key + idx * 0x9E3779B97F4A7C15ull);
  const float u1 = (static_cast<float>(r & 0xffffff) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = static_cast<float>((r >> 24) & 0xffffff) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
__device__ __forceinline__ uint64_t synth_mujoco_key(uint64_t seed, uint64_t counter) {
  return synth_mix64(seed * 0x9E3779B97F4A7C15ull + counter);
}
__device__ __forceinline__ float synth_mujoco_obs(uint64_t key, long long env, int k) {  // component k < 64 of env's observation
  const float z = synth_normal(key, static_cast<uint64_t>(env) * 64 + k);
  return fminf(fmaxf(z, -10.f), 10.f);
}
__device__ __forceinline__ float synth_mujoco_reward(uint64_t key, long long env) {
  return synth_normal(~key, static_cast<uint64_t>(env));
}
__device__ __forceinline__ bool synth_mujoco_reset(uint64_t key, long long env, float p_reset) {
  const uint64_t r = synth_mix64((key ^ 0xD1B54A32D192ED03ull) + static_cast<uint64_t>(env) * 0x9E3779B97F4A7C15ull);
  return static_cast<float>(r >> 40) * (1.0f / 16777216.0f) < p_reset;
}

// grid of the stand-alone launch (and the env share of the fused one)
inline int synth_blocks(long long nvec, int nenvs) {
  long long blocks = (nvec + 256 * 4 - 1) / (256 * 4);
  if (blocks > 4096) blocks = 4096;
  return static_cast<int>(blocks < 1 ? 1 : blocks);
}

}  // namespace
}  // namespace dx
