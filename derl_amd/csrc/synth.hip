// Counter-based synthetic environment step on the device (measurement input):
// uint8 frames i.i.d. uniform 0..255, rewards sign*[u < p_reward] in {-1,0,1}, resets
// Bernoulli(p_reset).  Stateless: every value is a hash of (seed, counter, position), so a
// step is one HBM-write-bound launch and reruns reproduce.
#include "synth_dev.hpp"

namespace {

__global__ __launch_bounds__(256) void synth_atari_kernel(const dx::SynthArgs a) {
  dx::synth_atari_block(a, blockIdx.x, gridDim.x);
}

// one thread per (env, component): obs_dim components, then the env's reward and reset
__global__ __launch_bounds__(256) void synth_mujoco_kernel(float *obs, float *rewards, uint8_t *resets, int nenvs, int obs_dim,
                                                           uint64_t seed, uint64_t counter, float p_reset) {
  const uint64_t key = dx::synth_mujoco_key(seed, counter);
  const long long total = static_cast<long long>(nenvs) * obs_dim;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const long long e = i / obs_dim;
    const int k = static_cast<int>(i - e * obs_dim);
    obs[i] = dx::synth_mujoco_obs(key, e, k);
    if (k == 0) {
      if (rewards) rewards[e] = dx::synth_mujoco_reward(key, e);
      if (resets) resets[e] = dx::synth_mujoco_reset(key, e, p_reset) ? 1 : 0;
    }
  }
}

}  // namespace

extern "C" int dx_synth_mujoco_step(float *obs, float *rewards, uint8_t *resets, int nenvs, int obs_dim, uint64_t seed,
                                    uint64_t counter, float p_reset, void *stream) {
  DX_TRACE("dx_synth_mujoco_step");
  DX_REQUIRE(obs && nenvs >= 1 && obs_dim >= 1 && obs_dim <= 64, "dx_synth_mujoco_step: need observations of 1 .. 64 components");
  const long long total = static_cast<long long>(nenvs) * obs_dim;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(synth_mujoco_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     dx::as_stream(stream), obs, rewards, resets, nenvs, obs_dim, seed, counter, p_reset);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

extern "C" int dx_synth_atari_step(void *frames, long long frame_bytes_total, float *rewards,
                                   uint8_t *resets, int nenvs, uint64_t seed, uint64_t counter,
                                   float p_reward, float p_reset, void *stream) {
  DX_TRACE("dx_synth_atari_step");
  DX_REQUIRE(frames && frame_bytes_total > 0 && frame_bytes_total % 16 == 0 && dx::aligned(frames, 16),
             "dx_synth_atari_step: frames must be 16-byte aligned, size a multiple of 16");
  DX_REQUIRE(nenvs >= 0, "dx_synth_atari_step: nenvs < 0");
  const long long nvec = frame_bytes_total / 16;
  const dx::SynthArgs a{static_cast<uint4 *>(frames), nvec, rewards, resets, nenvs, seed, counter, p_reward, p_reset, 0, 0};
  hipLaunchKernelGGL(synth_atari_kernel, dim3(dx::synth_blocks(nvec, nenvs)), dim3(256), 0,
                     dx::as_stream(stream), a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
